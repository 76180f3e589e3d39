#!/usr/bin/env python3
"""A/B of two builds of the engine library on one box:  python3 profiles/tools/lib_ab.py libA.so libB.so [libC.so ...] [reps]
Each library runs the bench batch (4096 units, K = 19) in its own child process, interleaved; prints ms per step and the
enumerate kernel's event time."""
import os, subprocess, sys
os.environ.setdefault("AMBI_EXPERIMENTS", "1")   # the engine honours its AMBI_* switches only with this (ambi_common.hpp: ambi_env)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, tempfile, time
sys.path.insert(0, %r)
import torch
from ambigram_amd import api, synth
api._preload_hip_runtime()
lib = api.load(sys.argv[1]); lib.ambi_set_device(0); torch.cuda.set_device(0)
tmp = tempfile.mkdtemp(); B = 4096
b = api.Batch(lib); keep = []
for i in range(B):
    s = synth.make_sample(256, 512, "wide", 19, seed=2000 + i)
    lh, sols = s.write(tmp, "s%%d" %% i)
    g = api.Graph(lib, lh); keep.append(g); b.add_chromosome_sol(g, 0, sols[0])
b.upload(); st = torch.cuda.current_stream().cuda_stream
b.run(0, st); b.wait()
for _ in range(3): b.run(0, st)
b.wait(); b.set_timing_only(["ambi_enumerate_kernel"])
t = time.perf_counter()
for _ in range(20): b.run(0, st)
b.wait(); dt = (time.perf_counter() - t) / 20
print("%%-40s %%.4f ms per step, enumerate %%.4f ms" %% (os.path.basename(sys.argv[1]), dt * 1e3, b.kernel_times()["ambi_enumerate_kernel"]))
''' % ROOT
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
reps = int(sys.argv[-1]) if not sys.argv[-1].endswith(".so") else 2
for r in range(reps):
    for l in libs:
        out = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(l)], capture_output=True, text=True)
        print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
