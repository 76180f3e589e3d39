#!/usr/bin/env python3
"""Single-sample latencies of bench.py's `single_sample` leg alone (no 4096-batch): inputs resident (run -> wait_results), with the
order table (run -> wait), end to end (upload -> run -> fetch_paths -> unit_path)."""
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
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
one.upload()
for _ in range(10):
    one.run(0); one.wait()
a = 0.0
for _ in range(reps):
    t = time.perf_counter(); one.run(0); one.wait_results(); a += time.perf_counter() - t; one.wait()
t = time.perf_counter()
for _ in range(reps):
    one.run(0); one.wait()
b = (time.perf_counter() - t)
c = 0.0
for _ in range(reps):
    t = time.perf_counter(); one.upload(); one.run(0); one.fetch_paths(); p = one.unit_path(0, 1); c += time.perf_counter() - t; one.wait()
print("us: resident run->results %.1f   run->table %.1f   e2e upload->path %.1f" % (a / reps * 1e6, b / reps * 1e6, c / reps * 1e6))
if os.environ.get("AMBI_STAGE_PROFILE"):
    one.run(0); one.wait(); one.download()      # (the stage profile is printed by download())
