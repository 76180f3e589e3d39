#!/usr/bin/env python3
"""Long randomized parity run, not collected by pytest (minutes):  python3 tests/soak.py [hip|hostsim] [first_seed] [count]
Random decompositions (both orientations, scan budget 2: units without a valid order in the first orientation go through
the parallel search of both passes; NOTE no known input has a first valid order other than order 0 of a pass -- `deep` in
the statistics stays 0 -- the search for a deeper one is tested with injected verdicts, engine_checks.check_injected_validity)
and synthetic samples of every tier with imperfect fold-backs and path-editing SVs, engine against the oracle."""
import os, sys, tempfile, time
os.environ.setdefault("AMBI_EXPERIMENTS", "1")   # the engine honours its AMBI_* switches only with this (ambi_common.hpp: ambi_env)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import engine_checks as ec
import parity
from ambigram_amd import api, synth
from oracle import oracle_py

which = sys.argv[1] if len(sys.argv) > 1 else "hip"
first = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
count = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
oracle_py.build(ref=False)
lib = api.load() if which == "hip" else api.load(os.path.join(HERE, "hostsim", "libambigram_hostsim.so"))
if which == "hip":
    lib.ambi_set_device(0)
work = tempfile.mkdtemp(prefix="ambi_soak_")
t0 = time.time()
st = ec.check_random_decompositions(lib, oracle_py, work, range(first, first + count), budget=2)
print("random decompositions %d..%d: %s  (%.0f s)" % (first, first + count, st, time.time() - t0), flush=True)
t0 = time.time()
bad = 0
n = 0
agreed_none = agreed_ub = 0
for i in range(count // 5):
    tier = ("chain", "wide", "mixed", "skew")[i % 4]
    K = (7, 9, 11, 13, 15, 17, 19)[i % 7] if tier != "skew" else (21, 23, 27, 33, 41)[i % 5]
    nseg = (40, 64, 96, 128, 256)[i % 5]
    s = synth.make_sample(nseg, 2 * nseg, tier, K, seed=first + 50000 + i, imperfect=(i // 2) % 2, n_del=i % 7)
    lh, sols = s.write(work, "k%d" % i)
    bad_before = bad
    for rev in (False, True):
        d = parity.compare(lib, oracle_py, lh, sols, reversed_=rev)
        n += 1
        if d:
            # two outcomes are agreement, not difference: the oracle finds no valid order either (e.g. a cyclic DAG: no
            # topological order at all), or the oracle flags that the reference reads out of bounds and the engine refuses
            oc = oracle_py.run_bfb(lh, sols, reversed_=rev)["chr"][0]
            if d == ["engine failed: no valid BFB order"] and oc["first_valid"] < 0 and not oc["ub_valid"]:
                agreed_none += 1
            elif d == ["engine failed: reference behaviour undefined on this input (out-of-bounds read)"] and oc["ub_valid"]:
                agreed_ub += 1
            else:
                bad += 1
                print("MISMATCH", lh, rev, d[:3], flush=True)
    if bad == bad_before:      # a long run must not fill the temporary directory (the files of a mismatch stay)
        for f in [lh] + list(sols):
            try:
                os.remove(f)
            except OSError:
                pass
print("synthetic samples: %d compared, %d mismatches; agreed: %d without a valid order, %d refused where the reference reads out of bounds (%.0f s)"
      % (n, bad, agreed_none, agreed_ub, time.time() - t0))
t0 = time.time()
st = ec.check_all_mode(lib, oracle_py, work, seeds=range(first, first + count // 20))
print("--all mode: %s (%.0f s)" % (st, time.time() - t0), flush=True)
ec.check_search_budget(lib, oracle_py, work)
ec.check_large_lattice(lib, oracle_py, work, K=56, k2=6)
print("search budget, large lattice: ok")
sys.exit(1 if bad else 0)
