import os, sys, time, tempfile
os.environ.setdefault("AMBI_EXPERIMENTS", "1")   # the engine honours its AMBI_* switches only with this (ambi_common.hpp: ambi_env)
sys.path.insert(0, "/root/repo")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from ambigram_amd import api, synth
lib = api.load(); lib.ambi_set_device(0)
tmp = tempfile.mkdtemp()
s = synth.make_sample(256, 512, "wide", 19, seed=2000)
lh, sols = s.write(tmp, "s0")
g = api.Graph(lib, lh)
one = api.Batch(lib); one.add_chromosome_sol(g, 0, sols[0])
for name, st in (("null stream", 0), ("torch current stream", torch.cuda.current_stream().cuda_stream), ("torch side stream", torch.cuda.Stream().cuda_stream)):
    one.upload()
    for _ in range(10):
        one.run(0, st); one.wait()
    c = 0.0
    reps = 300
    for _ in range(reps):
        t = time.perf_counter(); one.upload(); one.run(0, st); one.fetch_paths(); p = one.unit_path(0, 1); c += time.perf_counter() - t; one.wait()
    print("%-24s e2e %.1f us" % (name, c / reps * 1e6))
