#!/usr/bin/env python3
"""One sample of the soak's synthetic phase through the oracle, with the digests of its files:  one_sample.py first i"""
import sys, os, tempfile, hashlib
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
from ambigram_amd import synth
from oracle import oracle_py
oracle_py.build(ref=False)
first, i = int(sys.argv[1]), int(sys.argv[2])
tier = ("chain", "wide", "mixed", "skew")[i % 4]
K = (7, 9, 11, 13, 15, 17, 19)[i % 7] if tier != "skew" else (21, 23, 27, 33, 41)[i % 5]
nseg = (40, 64, 96, 128, 256)[i % 5]
s = synth.make_sample(nseg, 2 * nseg, tier, K, seed=first + 50000 + i, imperfect=(i // 2) % 2, n_del=i % 7)
d = tempfile.mkdtemp()
lh, sols = s.write(d, "k%d" % i)
print(tier, K, nseg, hashlib.md5(open(lh, "rb").read()).hexdigest(), [hashlib.md5(open(x, "rb").read()).hexdigest() for x in sols], flush=True)
o = oracle_py.run_bfb(lh, sols)
print("oracle ok", o["ok"], o["chr"][0]["first_valid"], flush=True)
