#!/usr/bin/env python3
"""Where the end-to-end latency of ONE sample goes (packed unit on the host -> final path on the host): per-call host times of
upload / run / fetch_paths / unit_path, and (AMBI_DEBUG_LATENCY=1) when the express and plan words arrived."""
import os, sys, time, tempfile
os.environ.setdefault("AMBI_EXPERIMENTS", "1")   # the engine honours its AMBI_* switches only with this (ambi_common.hpp: ambi_env)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from ambigram_amd import api, synth
lib = api.load(); lib.ambi_set_device(0)
tmp = tempfile.mkdtemp()
s = synth.make_sample(256, 512, "wide", 19, seed=2000)
lh, sols = s.write(tmp, "s0")
g = api.Graph(lib, lh)
one = api.Batch(lib); one.add_chromosome_sol(g, 0, sols[0])
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for _ in range(10):
    one.upload(); one.run(0); one.fetch_paths(); one.wait()
acc = [0.0] * 5
for _ in range(reps):
    t0 = time.perf_counter(); one.upload()
    t1 = time.perf_counter(); one.run(0)
    t2 = time.perf_counter(); one.fetch_paths()
    t3 = time.perf_counter(); p = one.unit_path(0, 1)
    t4 = time.perf_counter(); one.wait()
    t5 = time.perf_counter()
    for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
        acc[i] += d
print("us per call: upload %.1f  run %.1f  fetch_paths %.1f  unit_path %.1f  (wait for the table afterwards %.1f)  e2e %.1f" %
      tuple([a / reps * 1e6 for a in acc] + [sum(acc[:4]) / reps * 1e6]))
