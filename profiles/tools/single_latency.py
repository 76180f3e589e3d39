#!/usr/bin/env python3
"""Latency of ONE 256-segment sample resident in HBM: launch of the chain -> results complete (host-synchronised).

    python3 profiles/tools/single_latency.py [reps] [K]
Prints the mean wall time per run and the HIP-event time of every kernel of the chain.
"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ambigram_amd import api, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
K = int(sys.argv[2]) if len(sys.argv) > 2 else 19
lib = api.load()
lib.ambi_set_device(0)
torch.cuda.set_device(0)
tmp = tempfile.mkdtemp(prefix="ambi_single_")
s = synth.make_sample(256, 512, "wide", K, seed=2000)
lh, sols = s.write(tmp, "s0")
g = api.Graph(lib, lh)
one = api.Batch(lib)
one.add_chromosome_sol(g, 0, sols[0])
one.upload()
stream = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    one.run(0, stream); one.wait()
t = time.perf_counter()
for _ in range(reps):
    one.run(0, stream); one.wait()
wall = (time.perf_counter() - t) / reps
one.set_timing(True)
for _ in range(32):
    one.run(0, stream); one.wait()
kt = one.kernel_times()
one.set_timing(False)
one.download()
r = one.unit_result(0)
print("single sample: %.1f us per run (R = %d orders, status %d)" % (wall * 1e6, r["num_orders"], r["status"]))
print({k: round(v * 1e3, 1) for k, v in kt.items()}, "us per kernel")
