import os, sys, tempfile, time
os.environ.setdefault("AMBI_EXPERIMENTS", "1")   # the engine honours its AMBI_* switches only with this (ambi_common.hpp: ambi_env)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from ambigram_amd import api, synth
lib = api.load(); lib.ambi_set_device(0); torch.cuda.set_device(0)
tmp = tempfile.mkdtemp()
s = synth.make_sample(256, 512, "wide", 19, seed=2000)
lh, sols = s.write(tmp, "s0")
g = api.Graph(lib, lh); one = api.Batch(lib); one.add_chromosome_sol(g, 0, sols[0]); one.upload()
st = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    one.run(0, st); one.wait()
span = 0
for _ in range(300):
    t = time.perf_counter(); one.run(0, st); one.wait_results(); span += time.perf_counter() - t; one.wait()
print("run -> results: %.1f us" % (span / 300 * 1e6))
one.download()
