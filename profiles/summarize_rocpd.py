#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs into the small text files kept under profiles/.

    python profiles/summarize_rocpd.py stats  <results.db> <out.csv>          # --kernel-trace --stats pass
    python profiles/summarize_rocpd.py pmc    <results.db> <out.csv>          # one --pmc pass: per-kernel mean of every counter
    python profiles/summarize_rocpd.py traffic <wr.db> <rd.db> <out.json> --batch B --workload W --kernel K

ROCm 7.2's rocprofv3 writes a rocpd SQLite database by default; its views `top_kernels`, `kernels` and
`counters_collection` carry what the old *_kernel_stats.csv / *_counter_collection.csv files did.
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-byte read requests at 64 bytes, so it is
doubled here (MI355X_MICROARCH.md, "HBM"); WRITE_SIZE is exact for 16-byte-per-lane streaming stores.
"""
import argparse
import csv
import json
import sqlite3
import sys


def stats(db, out):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
        "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.3f" % (100.0 * r[2] / total), r[4], r[5]])
    return rows


def pmc(db, out):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
        "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Counter", "Dispatches", "MeanValuePerDispatch", "MeanDurationNs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], "%.3f" % r[3], "%.1f" % r[4]])
    return rows


def mean_counter(db, kernel_substr, counter, skip=4):
    """mean per dispatch over the timed launches (the first `skip` dispatches are the sizing + warm-up passes)"""
    c = sqlite3.connect(db)
    vals = [r[0] for r in c.execute(
        "select value from counters_collection where counter_name=? and kernel_name like ? order by dispatch_id",
        (counter, "%" + kernel_substr + "%"))]
    vals = vals[skip:] if len(vals) > skip else vals
    return sum(vals) / len(vals) if vals else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["stats", "pmc", "traffic"])
    ap.add_argument("paths", nargs="+")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--workload", default="256/512/wide/K19")
    ap.add_argument("--kernel", default="ambi_enumerate_blocks_kernel")
    ap.add_argument("--bench-name", default="ambi_enumerate_kernel", help="name bench.py uses for this kernel's timer")
    a = ap.parse_args()
    if a.mode == "stats":
        for r in stats(a.paths[0], a.paths[1])[:12]:
            print("%-70s calls=%d avg=%.1f us" % (r[0][:70], r[1], r[3] / 1e3))
    elif a.mode == "pmc":
        for r in pmc(a.paths[0], a.paths[1]):
            print("%-60s %-22s n=%d mean=%.3f" % (r[0][:60], r[1], r[2], r[3]))
    else:
        wr_kib = mean_counter(a.paths[0], a.kernel, "WRITE_SIZE")
        rd_kib = mean_counter(a.paths[1], a.kernel, "FETCH_SIZE")
        wr = wr_kib * 1024.0
        rd = rd_kib * 1024.0 * 2.0      # gfx950 correction: FETCH_SIZE tallies 128-byte requests at 64 bytes
        # every kernel of a step: sum over ALL dispatches of the run / number of steps (= dispatches of the prepare kernel)
        def total(db, counter):
            c = sqlite3.connect(db)
            tot = c.execute("select sum(value) from counters_collection where counter_name=? and kernel_name like '%ambi_%'", (counter,)).fetchone()[0] or 0.0
            steps = c.execute("select count(*) from counters_collection where counter_name=? and kernel_name like '%ambi_prepare_kernel%'", (counter,)).fetchone()[0] or 1
            per_kernel = {r[0].split("(")[0].replace("void ", "").replace("ambi::", ""): r[1] * 1024.0 / steps for r in c.execute(
                "select kernel_name, sum(value) from counters_collection where counter_name=? and kernel_name like '%ambi_%' group by kernel_name", (counter,))}
            return tot * 1024.0 / steps, per_kernel
        wr_all, wr_k = total(a.paths[0], "WRITE_SIZE")
        rd_all, rd_k = total(a.paths[1], "FETCH_SIZE")
        step_kernels = ("ambi_prepare_kernel", "ambi_plan_kernel", "ambi_blocks_build_kernel", "ambi_enumerate_blocks_kernel", "ambi_enumerate_kernel",
                        "ambi_first_kernel", "ambi_finish_lean_kernel", "ambi_finish_kernel", "ambi_finish_ext_kernel", "ambi_finish_edit_kernel", "ambi_express_kernel", "ambi_lattice_kernel")
        per_kernel = {k: {"write": wr_k.get(k, 0.0), "fetch_corrected": 2.0 * rd_k.get(k, 0.0)} for k in sorted(set(wr_k) | set(rd_k))
                      if k.split("<")[0] in step_kernels}
        step_bytes = sum(v["write"] + v["fetch_corrected"] for v in per_kernel.values())
        j = {"kernel": a.bench_name, "device_kernel": a.kernel, "batch": a.batch, "workload": a.workload,
             "hbm_bytes_per_step_all_kernels": step_bytes, "per_kernel_bytes_per_step": per_kernel,
             "write_bytes_per_launch": wr, "fetch_bytes_per_launch_corrected": rd,
             "fetch_size_raw_kib": rd_kib, "write_size_raw_kib": wr_kib,
             "hbm_bytes_per_launch": wr + rd,
             "method": "rocprofv3 --kernel-trace --pmc WRITE_SIZE and --pmc FETCH_SIZE in separate passes; "
                       "KiB -> bytes; FETCH_SIZE doubled (gfx950 128-byte requests counted at 64 bytes)"}
        json.dump(j, open(a.paths[2], "w"), indent=1)
        print(json.dumps(j))


if __name__ == "__main__":
    sys.exit(main())
