#!/usr/bin/env python3
"""The ILP fill kernel alone (one 256-segment model, 56.5 M non-zeros), for rocprofv3 passes:  python3 profiles/tools/ilp_fill_probe.py [reps]"""
import os, sys, tempfile
os.environ.setdefault("AMBI_EXPERIMENTS", "1")   # the engine honours its AMBI_* switches only with this (ambi_common.hpp: ambi_env)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ambigram_amd import api, synth
lib = api.load(os.environ.get("AMBI_BENCH_LIB") or None); lib.ambi_set_device(0)
tmp = tempfile.mkdtemp()
s = synth.make_sample(256, 512, "wide", 19, seed=2000)
lh, sols = s.write(tmp, "s0")
g = api.Graph(lib, lh)
b = api.Batch(lib); b.add_chromosome_sol(g, 0, sols[0]); b.upload(); b.run(0); b.download()
prep = b.unit_prepare(0, 256); r = b.unit_result(0)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    m = api.IlpModel(lib, g, 0, prep["seg_cn"], prep["junc_cn"], r["bias"], float(sum(prep["seg_cn"][1:])), device=True)
    print("fill %.4f ms = %.0f GB/s of 12-byte entries (%d non-zeros)" % (m.kernel_ms, 12 * m.nnz / (m.kernel_ms * 1e-3) / 1e9, m.nnz))
    m.close()
