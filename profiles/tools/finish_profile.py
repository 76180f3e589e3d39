#!/usr/bin/env python3
"""Shader-clock marks inside the full finish stage for units that carry path-editing SVs (run with AMBI_STAGE_PROFILE=1):
python3 profiles/tools/finish_profile.py [units]  -- marks 17..21 are cycles from the stage's first mark (16)."""
import os, sys, tempfile
os.environ.setdefault("AMBI_EXPERIMENTS", "1")   # the engine honours its AMBI_* switches only with this (ambi_common.hpp: ambi_env)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ambigram_amd import api, synth
libp = os.environ.get("AMBI_PROFILE_LIB")
if libp: api._preload_hip_runtime()
lib = api.load(libp); lib.ambi_set_device(0); torch.cuda.set_device(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lean = len(sys.argv) > 2 and sys.argv[2] == "lean"     # units without SVs: the lean finish stage
tmp = tempfile.mkdtemp(); b = api.Batch(lib); keep = []
for i in range(n):
    s = synth.make_sample(256, 512, "wide", 19, seed=2000 + 8 * i + 7, n_del=0 if lean else 2, n_dup=0 if lean else 1)
    lh, sols = s.write(tmp, "s%d" % i)
    g = api.Graph(lib, lh); keep.append(g); b.add_chromosome_sol(g, 0, sols[0])
b.upload(); st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    b.run(0, st); b.wait()
b.download()
r = [b.unit_result(u) for u in range(n)]
print("units", n, "status ok", sum(x["status"] == 0 for x in r), "mean path", sum(x["path_len"] for x in r) / n, "edited", sum(x["path_indel_len"] != x["path_len"] for x in r))
print(b.kernel_times() if hasattr(b, "kernel_times") else "")
