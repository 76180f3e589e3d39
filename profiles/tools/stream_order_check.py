"""Does the order in which the host application creates streams change the step?  4096-sample bench batch on the default stream or on a stream
of the caller's own, with N torch streams created BEFORE the engine's side streams:  python3 queue_probe_check.py default|own N"""
import os, sys, time, tempfile
os.environ.setdefault("AMBI_EXPERIMENTS", "1")   # the engine honours its AMBI_* switches only with this (ambi_common.hpp: ambi_env)
sys.path.insert(0, "/root/repo")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from ambigram_amd import api, synth
mode = sys.argv[1]
lib = api.load(); lib.ambi_set_device(0)
extra = [torch.cuda.Stream() for _ in range(int(sys.argv[2]))]      # streams the host application made first
st = extra[0].cuda_stream if (extra and mode == "own") else torch.cuda.current_stream().cuda_stream
tmp = tempfile.mkdtemp(); B = 4096
b = api.Batch(lib); keep = []
for i in range(B):
    s = synth.make_sample(256, 512, "wide", 19, seed=2000 + i, n_del=2 if i % 8 == 7 else 0, n_dup=1 if i % 8 == 7 else 0)
    lh, sols = s.write(tmp, "s%d" % i)
    g = api.Graph(lib, lh); keep.append(g); b.add_chromosome_sol(g, 0, sols[0])
b.upload()
b.run(0, st); b.wait()
for _ in range(3): b.run(0, st)
b.wait()
t = time.perf_counter()
for _ in range(20): b.run(0, st)
b.wait(); dt = (time.perf_counter() - t) / 20
print("caller stream %-8s  %s streams made first: %.4f ms per step" % (mode, sys.argv[2], dt * 1e3))
