#!/usr/bin/env python3
"""Which HIP streams of a process dispatch side by side?  ambi_debug_stream_probe over pairs of: the legacy default stream (0) and
N streams created here.   python3 profiles/tools/stream_pairs.py [n_streams] [torch|hip]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from ambigram_amd import api
lib = api.load(); lib.ambi_set_device(0); torch.cuda.set_device(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
kind = sys.argv[2] if len(sys.argv) > 2 else "torch"
streams = [0]
keep = []
if kind == "torch":
    for _ in range(n): s = torch.cuda.Stream(); keep.append(s); streams.append(s.cuda_stream)
else:
    hip = C.CDLL("libamdhip64.so")
    for _ in range(n):
        h = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(h), 1) == 0; streams.append(h.value)
us = C.c_float()
print("rows: stream with the backlog; columns: stream of the tiny kernel; microseconds until it has run")
for a in range(len(streams)):
    row = []
    for b in range(len(streams)):
        if a == b: row.append("   -  "); continue
        lib.ambi_debug_stream_probe(C.c_void_p(streams[a]), C.c_void_p(streams[b]), C.byref(us))
        row.append("%6.1f" % us.value)
    print("%2d: %s" % (a, " ".join(row)))
