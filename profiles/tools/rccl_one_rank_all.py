#!/usr/bin/env python3
"""One-rank RCCL exercise of the sharded --all merge (a 1-GPU box cannot run two ranks): the engine's own bitmap pool is
wrapped as a torch tensor through the CUDA array interface and goes through dist.all_reduce(MAX) on the nccl backend.
    python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 profiles/tools/rccl_one_rank_all.py
"""
import os, sys, tempfile
os.environ.setdefault("AMBI_EXPERIMENTS", "1")   # the engine honours its AMBI_* switches only with this (ambi_common.hpp: ambi_env)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from ambigram_amd import api, synth
from ambigram_amd.dist import all_mode_merge

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
lib = api.load(); lib.ambi_set_device(0)
tmp = tempfile.mkdtemp()
graphs, b = [], api.Batch(lib)
for i in range(8):
    s = synth.make_sample(64, 128, "wide", 13, seed=6200 + i)
    lh, sols = s.write(tmp, "w%d" % i)
    g = api.Graph(lib, lh); graphs.append(g)
    b.add_chromosome_sol(g, 0, sols[0])
b.all_set_shard(0, 1)
b.upload()
stream = torch.cuda.current_stream().cuda_stream
b.run(api.FLAG_ALL, stream); b.wait()
before = [b.all_orders(u, 0).tolist() for u in range(8)]
all_mode_merge(b, "cuda", stream, force=True)
after = [b.all_orders(u, 0).tolist() for u in range(8)]
assert before == after and all(len(x) == 924 for x in after), [len(x) for x in after]
print("one-rank RCCL merge of the --all pool: ok (%d bytes)" % b.all_device()[1])
dist.destroy_process_group()
