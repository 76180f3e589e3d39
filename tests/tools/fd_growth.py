#!/usr/bin/env python3
"""Do repeated batches leak process resources?  Open file descriptors and resident memory after every 100 create / run / close
cycles of a one-sample batch:  fd_growth.py hip|hostsim [cycles]"""
import os, sys, tempfile, resource
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
from ambigram_amd import api, synth
which = sys.argv[1]; cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 600
lib = api.load() if which == "hip" else api.load(os.path.join(HERE, "hostsim", "libambigram_hostsim.so"))
if which == "hip":
    lib.ambi_set_device(0)
d = tempfile.mkdtemp()
s = synth.make_sample(64, 128, "wide", 9, seed=5)
lh, sols = s.write(d, "a")
for c in range(cycles + 1):
    if c % 100 == 0:
        print("cycle %4d: %4d fds, max rss %d MB" % (c, len(os.listdir("/proc/self/fd")), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024), flush=True)
    r = api.reconstruct_sample(lib, lh, sols)
