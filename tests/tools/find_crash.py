#!/usr/bin/env python3
"""Replays the synthetic-sample phase of tests/soak.py one sample at a time and logs the sample index BEFORE every step, so
that a hard crash (an exception escaping from native code) names its input:  python3 tests/tools/find_crash.py hip|hostsim first count [log]"""
import os, sys, tempfile
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import parity
from ambigram_amd import api, synth
from oracle import oracle_py
which, first, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
log = open(sys.argv[4], "w") if len(sys.argv) > 4 else sys.stdout
oracle_py.build(ref=False)
lib = api.load() if which == "hip" else api.load(os.path.join(HERE, "hostsim", "libambigram_hostsim.so"))
if which == "hip":
    lib.ambi_set_device(0)
work = tempfile.mkdtemp(prefix="ambi_crash_")
for i in range(count // 5):
    tier = ("chain", "wide", "mixed", "skew")[i % 4]
    K = (7, 9, 11, 13, 15, 17, 19)[i % 7] if tier != "skew" else (21, 23, 27, 33, 41)[i % 5]
    nseg = (40, 64, 96, 128, 256)[i % 5]
    s = synth.make_sample(nseg, 2 * nseg, tier, K, seed=first + 50000 + i, imperfect=(i // 2) % 2, n_del=i % 7)
    lh, sols = s.write(work, "k%d" % i)
    for rev in (False, True):
        print("i %d rev %d oracle" % (i, rev), file=log, flush=True)
        o = oracle_py.run_bfb(lh, sols, reversed_=rev)
        if not o["ok"]:
            import shutil
            keep = os.path.join(os.path.dirname(os.path.abspath(log.name)) if log is not sys.stdout else ".", "crash_files")
            os.makedirs(keep, exist_ok=True)
            for f in [lh] + list(sols):
                shutil.copy(f, keep)
            try:
                nfd = len(os.listdir("/proc/self/fd"))
            except OSError as e:
                nfd = repr(e)
            st = open("/proc/self/status").read()
            print("fds %s; %s; maps %d; python reads lh: %d bytes, sol: %d bytes" % (nfd, [l for l in st.splitlines() if l.startswith(("Threads", "VmRSS", "VmSize"))],
                  sum(1 for _ in open("/proc/self/maps")), len(open(lh).read()), len(open(sols[0]).read())), file=log, flush=True)
            print("ORACLE FAILED at i %d rev %d: %s; files copied to %s; retry: %r" % (i, rev, o["err"], keep, oracle_py.run_bfb(lh, sols, reversed_=rev)["ok"]), file=log, flush=True)
            sys.exit(3)
        print("i %d rev %d engine" % (i, rev), file=log, flush=True)
        d = parity.compare(lib, oracle_py, lh, sols, reversed_=rev)
    for f in [lh] + list(sols):
        os.remove(f)
print("done", file=log, flush=True)
