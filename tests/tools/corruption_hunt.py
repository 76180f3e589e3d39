#!/usr/bin/env python3
"""Host-memory corruption detector: samples are generated twice (same seed) and their texts compared after the engine has
run on the previous sample -- a stray write from the engine (or a late one from the GPU) into Python's heap shows up as a
difference.  corruption_hunt.py hip|hostsim count"""
import os, sys, tempfile
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
from ambigram_amd import api, synth
which, count = sys.argv[1], int(sys.argv[2])
lib = api.load() if which == "hip" else api.load(os.path.join(HERE, "hostsim", "libambigram_hostsim.so"))
if which == "hip":
    lib.ambi_set_device(0)
work = tempfile.mkdtemp(prefix="ambi_hunt_")
first = 100000
def gen(i):
    tier = ("chain", "wide", "mixed", "skew")[i % 4]
    K = (7, 9, 11, 13, 15, 17, 19)[i % 7] if tier != "skew" else (21, 23, 27, 33, 41)[i % 5]
    nseg = (40, 64, 96, 128, 256)[i % 5]
    return synth.make_sample(nseg, 2 * nseg, tier, K, seed=first + 50000 + i, imperfect=(i // 2) % 2, n_del=i % 7)
bad = 0
for i in range(count):
    s = gen(i)
    lh, sols = s.write(work, "k%d" % i)
    t = gen(i)
    if s.lh_text != t.lh_text or s.sol_texts != t.sol_texts or open(lh).read() != t.lh_text or open(sols[0]).read() != t.sol_texts[0]:
        bad += 1
        print("CORRUPTED text at sample", i, flush=True)
        if bad >= 3:
            break
    for rev in (False, True):
        api.reconstruct_sample(lib, lh, sols, reversed_=rev)
    for f in [lh] + list(sols):
        os.remove(f)
print("done: %d samples, %d corrupted" % (i + 1, bad), flush=True)
