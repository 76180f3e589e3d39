#!/usr/bin/env python3
"""Kernel timeline of the last steps of a rocprofv3 --kernel-trace run (rocpd database): start / end of every dispatch
relative to the first kernel of the step, and the gaps between them.
    python3 profiles/tools/timeline.py <results.db> [steps]
"""
import sqlite3, sys
db = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# a step starts with ambi_prepare_kernel
idx = [i for i, r in enumerate(rows) if "ambi_prepare_kernel" in r[0]]
for s in idx[-steps - 1:-1]:
    e = idx[idx.index(s) + 1]
    t0 = rows[s][1]
    print("step:")
    for name, a, b in rows[s:e]:
        short = name.split("(")[0].replace("ambi::", "").replace("void ", "")[:44]
        print("  %-44s start %8.1f us  end %8.1f us  dur %7.1f us" % (short, (a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3))
    print("  next step starts at %.1f us" % ((rows[e][1] - t0) / 1e3))
