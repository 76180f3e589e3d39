#!/usr/bin/env python3
"""Experiment: two resident batches on two HIP streams, steps alternating between them (double buffering), against one
batch on one stream.  python3 profiles/tools/two_batches.py [batch] [steps]"""
import os, sys, tempfile, time
os.environ.setdefault("AMBI_EXPERIMENTS", "1")   # the engine honours its AMBI_* switches only with this (ambi_common.hpp: ambi_env)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ambigram_amd import api, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = api.load(); lib.ambi_set_device(0); torch.cuda.set_device(0)
tmp = tempfile.mkdtemp(prefix="ambi_two_")
batches, keep = [], []
for k in range(2):
    b = api.Batch(lib)
    for i in range(B):
        s = synth.make_sample(256, 512, "wide", 19, seed=2000 + k * B + i)
        lh, sols = s.write(tmp, "s%d_%d" % (k, i))
        g = api.Graph(lib, lh); keep.append(g)
        b.add_chromosome_sol(g, 0, sols[0])
    b.upload()
    batches.append(b)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for b, st in zip(batches, streams):
    b.run(0, st.cuda_stream); b.wait()
    for _ in range(2):
        b.run(0, st.cuda_stream)
    b.wait()
torch.cuda.synchronize()
# one batch, one stream
t = time.perf_counter()
for _ in range(steps):
    batches[0].run(0, streams[0].cuda_stream)
batches[0].wait(); torch.cuda.synchronize()
one = (time.perf_counter() - t) / steps
# two batches, two streams, alternating
t = time.perf_counter()
for k in range(steps):
    batches[k % 2].run(0, streams[k % 2].cuda_stream)
batches[0].wait(); batches[1].wait(); torch.cuda.synchronize()
two = (time.perf_counter() - t) / steps
print("one batch / one stream: %.4f ms per step (%.0f /s); two batches / two streams: %.4f ms per step (%.0f /s)" % (one * 1e3, B / one, two * 1e3, B / two))
