#!/usr/bin/env python3
"""A few steps of the bench batch in one mode, for a rocprofv3 --kernel-trace timeline (profiles/tools/timeline.py):
    python3 profiles/tools/mode_steps.py [lazy|plain] [steps] [lib.so|-] [sv_every] [default|own|own_first|rccl|rccl_own]
the last argument: which stream the batch runs on -- torch's current (legacy default) stream, a torch stream of the caller's own created
after / before the engine has created its side streams, and the same with a one-rank RCCL group initialised first."""
import os, sys, tempfile, time
os.environ.setdefault("AMBI_EXPERIMENTS", "1")   # the engine honours its AMBI_* switches only with this (ambi_common.hpp: ambi_env)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from ambigram_amd import api, synth
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
libp = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else None
sv_every = int(sys.argv[4]) if len(sys.argv) > 4 else 8
smode = sys.argv[5] if len(sys.argv) > 5 else "default"
if smode.startswith("rccl"):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    t_ = torch.ones(4, device="cuda"); dist.all_reduce(t_); torch.cuda.synchronize()
own = None
if smode == "own_first": torch.cuda.set_device(0); own = torch.cuda.Stream()
if libp: api._preload_hip_runtime()
lib = api.load(libp); lib.ambi_set_device(0); torch.cuda.set_device(0)
tmp = tempfile.mkdtemp(); B = int(os.environ.get("AMBI_STEPS_BATCH", "4096"))
b = api.Batch(lib); keep = []
for i in range(B):
    edits = sv_every > 0 and i % sv_every == sv_every - 1
    s = synth.make_sample(256, 512, "wide", 19, seed=2000 + i, n_del=2 if edits else 0, n_dup=1 if edits else 0)
    lh, sols = s.write(tmp, "s%d" % i)
    g = api.Graph(lib, lh); keep.append(g); b.add_chromosome_sol(g, 0, sols[0])
b.upload()
if smode in ("own", "rccl_own"): b.run(0, torch.cuda.current_stream().cuda_stream); b.wait(); own = torch.cuda.Stream()
st = own.cuda_stream if own is not None else torch.cuda.current_stream().cuda_stream
flags = api.FLAG_LAZY_ORDERS if mode == "lazy" else 0
b.run(0, st); b.wait()
for _ in range(3): b.run(flags, st)
b.wait(); torch.cuda.synchronize()
t = time.perf_counter()
if mode == "hostpaths":      # every step ends with its final paths on their way to the host (bench.py's paths_on_host leg)
    for i in range(steps):
        b.run(0, st); b.runs_to_host(1, i % 2, st)
        if i >= 1: b.runs_wait((i - 1) % 2)
    b.runs_wait((steps - 1) % 2)
else:
    for _ in range(steps): b.run(flags, st)
b.wait(); torch.cuda.synchronize()
print("%s on %s stream: %.4f ms per step" % (mode, smode, (time.perf_counter() - t) / steps * 1e3))
