#!/usr/bin/env python3
"""bench.py -- BFB reconstructions/sec on MI355X (BASELINE.json metric).

A "step" is one pass of the whole post-.sol reconstruction pipeline (SURVEY.md 8a #7,#8,#11-#16,#20: getJuncCN,
bias, getIndelBias, targetCN, constructDAG, allTopologicalOrders, getBFB+imperfectFBI, indelBFB, output junctions)
over one batch of synthetic units that is already resident in HBM; the results (paths, breakpoints, output junctions)
stay in the HBM of the GPU that produced them.  Samples are independent, so N GPUs run N batches (weak scaling) with
no collective inside a step; for N > 1 the timed region ends with the single exchange the north star names ("a single
RCCL gather over xGMI at the end"): the final paths are packed on the device in run-length form, sent to rank 0 with
one RCCL gather and expanded there (`--gather 0`: leave them in each GPU's HBM, `--gather 2`: gather at the end of every
step).

Workload (config.workload): BASELINE.json configs[2], the configuration the metric is quoted on: synthetic
256-segment / 512-junction .lh samples, wide DAG tier K=19 (R = C(18,9) = 48 620 topological orders per sample),
default CLI mode (first valid order).  Every rank holds its own batch of `--batch` samples (weak scaling); the ILP
solve is replaced by the planted .sol exactly as SURVEY.md 8c/8d prescribes.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The engine runs its kernels on four HIP streams (caller's, scan, lean finish, direct launch for the units whose SVs may edit the path).  The HIP runtime puts the
# streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order; once RCCL has created its own
# streams (init_process_group), two of the engine's land on ONE queue and the finish kernels run BEHIND the enumerate
# kernel instead of beside it: 1.55 instead of 1.10 ms per step, measured.  Eight queues keep them apart (and are no worse
# without RCCL: 1.10 vs 1.11-1.14).  Must be in the environment before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md; ~6.3 TB/s achievable)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="samples (units) per GPU")
    ap.add_argument("--segs", type=int, default=256)
    ap.add_argument("--juncs", type=int, default=512)
    ap.add_argument("--tier", default="wide", choices=["chain", "wide", "mixed"])
    ap.add_argument("--K", type=int, default=19)
    ap.add_argument("--sv-every", type=int, default=8, help="every N-th sample carries deletions / duplications that make indelBFB EDIT the path "
                    "(SURVEY.md 8d padding; those units go through indelBFB with edits -- the edit stage on the runs of the path, the full stage for what it hands on -- inside the timed region); 0: none")
    ap.add_argument("--mode", default="default", choices=["default", "all"], help="all: the timed step runs --all (every order of every sample "
                    "evaluated by the fused unrank + evaluate kernel); the headline value stays reconstructions/s")
    ap.add_argument("--all-steps", type=int, default=2, help="steps of the --all leg reported beside the headline (0 disables it)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget (0 disables it)")
    ap.add_argument("--single-reps", type=int, default=200, help="repetitions of the single-sample latency leg (0 disables it)")
    ap.add_argument("--pipelined", type=int, default=1, help="1: also report the throughput with two resident batches on two streams (N = 1 only)")
    ap.add_argument("--host-paths", type=int, default=1, help="1: also report the step that ENDS WITH THE PATHS ON THE HOST (packed runs + one D2H per step, double-buffered; SURVEY.md 8d's region)")
    ap.add_argument("--streams-leg", type=int, default=1, help="1: also report the step on a stream of the caller's own and, in a child process, behind a one-rank RCCL group")
    ap.add_argument("--sharded-leg", type=int, default=1, help="1: also report ambi_batch_run_sharded (the C-level multi-device driver) with 1 / 2 / 4 shares of the batch on the ONE GPU")
    ap.add_argument("--lazy", type=int, default=1, help="1: also report the step without the order tables (AMBI_FLAG_LAZY_ORDERS); 0: skip that leg")
    ap.add_argument("--target-lanes", type=int, default=0)
    ap.add_argument("--slices", type=int, default=0, help="unit ranges run on separate HIP streams (0: engine default)")
    ap.add_argument("--gather", type=int, default=-1, help="1: the timed region ends with the packing of the paths + ONE RCCL gather to "
                    "rank 0 (north star: 'a single RCCL gather over xGMI at the end'); 2: at the end of EVERY step; 0: results stay in "
                    "each GPU's HBM; -1 (default): 1 when N > 1, 0 at N = 1 where there is nothing to exchange")
    return ap.parse_args()


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from ambigram_amd import api, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    torch.cuda.set_device(local_rank)
    # AMBI_FORCE_DIST=1 (diagnostics): initialise RCCL and issue the end-of-batch collectives even with one rank, so that a
    # 1-GPU box exercises the very calls the N > 1 job makes (launch under torch.distributed.run --nproc-per-node 1)
    force_dist = os.environ.get("AMBI_FORCE_DIST") == "1" and "RANK" in os.environ
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.slices > 0:
        os.environ["AMBI_SLICES"] = str(args.slices); os.environ["AMBI_EXPERIMENTS"] = "1"   # (an experiment switch: honoured only with the second one set)
    # the HIP engine; raises if it has not been built (no CPU fallback).  AMBI_BENCH_LIB: another BUILD of the same engine
    # library (profiles/tools/ab.sh compares two builds on one box with it)
    lib = api.load(os.environ.get("AMBI_BENCH_LIB") or None)
    if os.environ.get("AMBI_BENCH_LIB"): api._preload_hip_runtime()
    lib.ambi_set_device(local_rank)

    # ---- workload: B samples per rank, seeds as SURVEY.md 8d (seed = 1000*config + sample index) ----
    B = args.batch
    tmp = tempfile.mkdtemp(prefix="ambi_bench_")
    graphs, files = [], []
    batch = api.Batch(lib)
    batch.configure(target_lanes=args.target_lanes)
    for i in range(B):
        edits = args.sv_every > 0 and i % args.sv_every == args.sv_every - 1
        s = synth.make_sample(args.segs, args.juncs, args.tier, args.K, seed=1000 * 2 + rank * B + i, n_del=2 if edits else 0, n_dup=1 if edits else 0)
        lh, sols = s.write(tmp, "s%d" % i)
        g = api.Graph(lib, lh)
        graphs.append(g)
        batch.add_chromosome_sol(g, 0, sols[0])
        files.append((lh, sols))
    batch.upload()                         # inputs now resident in HBM
    stream = torch.cuda.current_stream().cuda_stream

    # one untimed pass to size the order arena and the gather buffers
    batch.run(0, stream); batch.wait(); batch.download()
    res = [batch.unit_result(u) for u in range(B)]
    bad = [r for r in res if r["status"] != 0]
    if bad:
        raise SystemExit("bench: %d units did not reconstruct: %r" % (len(bad), bad[0]))
    from ambigram_amd.dist import RunExchange
    total_cells = sum(r["path_indel_len"] for r in res)
    n_runs, n_cells = RunExchange.probe(batch, B, "cuda", 1, stream)
    assert n_cells == total_cells
    px = RunExchange(lib, B, n_runs, n_cells, "cuda", world=world, rank=rank, force_collectives=force_dist)

    gather_mode = (1 if (world > 1 or force_dist) else 0) if args.gather < 0 else args.gather

    def gather():
        # the single end-of-batch exchange of the north star: the final paths in run-length form (a few dozen runs of
        # consecutive segments per sample instead of thousands of cells), one all-gather of the per-sample counts and one
        # gather of the runs to rank 0 (RCCL over xGMI), where every rank's runs are expanded into cells again.  The
        # reconstruction itself has no exchange step (samples are independent), so at N = 1 there is nothing to send and
        # the step is the pipeline alone.
        px.pack(batch, 1, stream)
        px.exchange(stream)
        px.expand(stream)

    run_flags = api.FLAG_ALL if args.mode == "all" else 0

    def step():
        batch.run(run_flags, stream)
        if run_flags:
            batch.wait()                   # --all: the evaluation of every order is issued when the run is waited for
        if gather_mode == 2:
            gather()

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()

    # Warm-up with HIP events around EVERY kernel (on the streams the kernels run on): per-kernel times of the step and
    # which kernel dominates.  In the timed region only that kernel keeps its events: every event pair is a marker in the
    # stream and the twelve of a fully timed step cost ~4 % (profiles/r01_slices.md).
    batch.set_timing(True)
    for _ in range(max(args.warmup, 1)):
        step()
    if gather_mode != 0:
        gather()                           # untimed: first use of the exchange (collective setup, lazily loaded kernels)
    batch.wait()
    torch.cuda.synchronize()
    warm_times = {k: v for k, v in batch.kernel_times().items() if v >= 0 and k != "ambi_all_kernel"}
    warm_spans = {k: v for k, v in batch.kernel_spans().items() if v[0] >= 0 and v[1] >= 0 and k in warm_times}
    # The roofline kernel is chosen by measured time, not by name: the largest HIP-event duration among the kernels whose duration is
    # set by their own work.  Two spans are NOT such durations and are listed (`time_ranking`, `critical_path`) but not candidates:
    # "ambi_finish_kernel" (lean finish + list kernel on their stream) and "ambi_finish_ext_kernel" (direct full finish) run BESIDE the
    # order-table kernel as paced side work -- the lean kernel's grid is sized so that its few workgroups work through the batch for
    # as long as the table is being written (HipBackend::finish_grid_for), the direct launch's workgroups wait for free slots on the
    # CUs -- so their spans track the table kernel's by construction (0.25 / 0.10 ms when they run alone, profiles/r04_*).
    paced = ("ambi_finish_kernel", "ambi_finish_ext_kernel")
    own_work = {k: v for k, v in warm_times.items() if k not in paced}
    dom = max(own_work, key=lambda k: own_work[k]) if own_work else "ambi_enumerate_kernel"
    time_ranking = [{"kernel": k, "ms": round(v, 4), "paced_side_work": k in paced} for k, v in sorted(warm_times.items(), key=lambda kv: -kv[1])]
    batch.set_timing_only([dom])
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if gather_mode == 1:
        gather()                           # the single exchange at the end of the job, inside the timed region
    batch.wait()
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    if world > 1 or force_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ktimes = dict(warm_times)
    ktimes[dom] = batch.kernel_times().get(dom, warm_times.get(dom, float("nan")))   # the dominant kernel: measured over the timed steps
    batch.set_timing(False)
    # the step's critical path from the HIP-event timeline of the warm-up steps (all kernels timed): the front (everything before
    # the first workgroup of the order-table kernel), the table kernel, and the tail (what still runs after the table is complete:
    # the finish kernels on their side streams), all in ms from the start of the step's first kernel
    critical = None
    if "ambi_enumerate_kernel" in warm_spans and "ambi_prepare_kernel" in warm_spans:
        t_a, t_b = warm_spans["ambi_enumerate_kernel"]
        last = max(warm_spans, key=lambda k: warm_spans[k][1])
        critical = {"front_ms": t_a, "table_ms": t_b - t_a, "tail_ms": max(0.0, warm_spans[last][1] - t_b), "last_kernel": last,
                    "end_of_last_kernel_ms": warm_spans[last][1],
                    "spans_ms": {k: [round(v[0], 4), round(v[1], 4)] for k, v in sorted(warm_spans.items(), key=lambda kv: kv[1][0])},
                    "note": "HIP events on the streams the kernels run on, warm-up steps (every kernel carries events there: ~4 % slower than the timed steps); "
                            "ambi_finish_kernel = lean finish + list kernel on their stream, ambi_finish_ext_kernel = the direct launch on its own stream: ambi_finish_edit_kernel (indelBFB on the runs of the path) + the full-stage launch for what it hands on"}

    # sanity (outside the timed region): the exchanged payload, expanded again, equals the downloaded paths
    gather()
    torch.cuda.synchronize()
    batch.download()
    assert int(px.totals[1].item()) == total_cells and int(px.totals[0].item()) == n_runs
    l_host = px.lengths.cpu().tolist()
    assert l_host == [batch.unit_result(u)["path_indel_len"] for u in range(B)]
    if rank == 0:
        import numpy as np
        mine = px.cells_all[0][:total_cells].cpu().numpy()
        want = np.concatenate([batch.unit_path(u, 1) for u in range(B)]) if B else np.zeros(0, np.int32)
        assert mine.shape == want.shape and bool((mine == want).all()), "expanded runs differ from the downloaded paths"

    if rank != 0:
        if world > 1 or force_dist:
            dist.destroy_process_group()
        return

    units_total = B * world
    value = units_total * args.steps / dt
    ms_per_step = dt / args.steps * 1e3

    # ---- algorithmic bytes: B = 8n + 24m + 16K + 2RK + 4EL + 4P per unit, split per kernel ----
    n, m = args.segs, None
    per_kernel = {"ambi_prepare_kernel": 0, "ambi_plan_kernel": 0, "ambi_blocks_build_kernel": 0, "ambi_enumerate_kernel": 0, "ambi_first_kernel": 0, "ambi_finish_kernel": 0}
    formula = 0
    stored_table_bytes = 0      # what the order tables occupy in the engine's own row layout (5 bits per node up to 32 nodes, csrc/ambi_orders.hpp)
    def row_bytes(K):
        return 4 * ((5 * K + 31) // 32) if K <= 32 else (4 * ((6 * K + 31) // 32) if K <= 63 else 128)
    for u, r in enumerate(res):
        mj = graphs[u].n_junc
        K, R, E, L, P, P2 = r["n_nodes"], r["num_orders"], r["evaluated"], r["bkp_len"], r["path_len"], r["path_indel_len"]
        per_kernel["ambi_prepare_kernel"] += 8 * n + 24 * mj + 16 * K
        per_kernel["ambi_plan_kernel"] += 64
        per_kernel["ambi_enumerate_kernel"] += R * K                 # every order written once (one byte per node)
        stored_table_bytes += R * row_bytes(K)
        per_kernel["ambi_first_kernel"] += E * K + 2 * E * L         # orders read until the first valid one, bkp written
        # bkp read, path written; the path after indelBFB is written only when indelBFB changed it
        per_kernel["ambi_finish_kernel"] += 2 * L + 4 * P + (4 * P2 if r["path_indel_stored"] else 0)
        formula += 8 * n + 24 * mj + 16 * K + 2 * R * K + 4 * E * L + 4 * P
    # every kernel is launched once per slice and step (slices run on separate streams and overlap); kernel_times() is
    # the average duration of ONE launch, so the bytes are taken per launch as well
    slices = max(1, batch.slices())
    dom_ms = ktimes.get(dom, float("nan"))
    dom_bytes = per_kernel.get(dom, 0) / slices
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms and dom_ms > 0 else None
    # HBM traffic by the PMC counters (rocprofv3 --pmc passes of this very command, corrected as MI355X_MICROARCH.md
    # prescribes; profiles/traffic_r04.json, written by profiles/tools/collect.sh + summarize): per launch of the dominant
    # kernel, and summed over every kernel of a step
    traffic, step_traffic = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_r04.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("kernel") == dom and tj.get("batch") == B and tj.get("slices", 1) == slices and tj.get("workload") == "%d/%d/%s/K%d/sv%d" % (args.segs, args.juncs, args.tier, args.K, args.sv_every):
                traffic = tj.get("hbm_bytes_per_launch")
                step_traffic = tj.get("hbm_bytes_per_step_all_kernels")
        except Exception:
            traffic = None
    stored = (stored_table_bytes / slices) if dom == "ambi_enumerate_kernel" else None
    stored_rate = stored / (dom_ms * 1e-3) / 1e9 if stored and dom_ms and dom_ms > 0 else None
    roofline = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic,
                "algorithmic_bytes_per_launch": dom_bytes, "kernel_ms": dom_ms, "launches_per_step": slices,
                # the same launch counted in the bytes the table really has: the engine stores a row at 5 bits per node (12 bytes at
                # K = 19 where SURVEY 8d's algorithmic figure, used for `achieved`, counts 19), so `traffic` is BELOW the algorithmic bytes
                "stored_bytes_per_launch": stored, "stored_GBps": stored_rate, "stored_frac": (stored_rate / HBM_PEAK_GBPS) if stored_rate else None,
                "row_layout": "5 bits per node up to 32 nodes, 6 up to 63 (lossless; ambi_batch_unit_orders unpacks), one byte per node above",
                "algorithmic_frac": (achieved / HBM_PEAK_GBPS) if achieved else None,
                "dominant_by": "largest HIP-event duration among the kernels whose duration is set by their own work (warm-up steps; re-measured over the timed steps); the paced finish spans are ranked in time_ranking and sit in critical_path",
                "time_ranking": time_ranking,
                "critical_path": critical,
                "all_kernels_ms": ktimes, "all_kernels_note": "%s: HIP events over the timed steps; the other kernels: over the warm-up steps" % dom,
                "step_hbm_bytes_pmc": step_traffic,
                "step_hbm_GBps_pmc": (step_traffic / (dt / args.steps) / 1e9) if step_traffic else None}

    # ---- CPU baseline: the oracle (a single-thread port of the reference path) on this box's host cores ----
    cpu = None
    if args.cpu_seconds > 0 and world == 1:      # rank 0 at N = 1 only
        from oracle import oracle_py
        oracle_py.build(ref=False)
        spent, recon, whole, cnt = 0.0, 0.0, 0.0, 0
        t_start = time.perf_counter()
        for (lh, sols) in files:
            r = oracle_py.run_bfb(lh, sols)
            assert r["ok"]
            # the oracle doubles as a last parity check on the benchmarked inputs
            assert r["chr"][0]["path_indel"] == batch.unit_path(cnt, 1).tolist(), "parity lost on bench sample %d" % cnt
            recon += r["recon_seconds"]; whole += r["seconds"]; cnt += 1
            spent = time.perf_counter() - t_start
            if spent > args.cpu_seconds:
                break
        model = ""
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        # the same port at -O0, the optimisation level of the reference's shipped build (CMakeLists.txt:7-8), on a few of the
        # same samples
        o0 = None
        try:
            n0, recon0 = 0, 0.0
            t_start = time.perf_counter()
            for (lh, sols) in files[:max(1, min(cnt, 64))]:
                r0 = oracle_py.run_bfb(lh, sols, O0=True)
                recon0 += r0["recon_seconds"]; n0 += 1
                if time.perf_counter() - t_start > max(3.0, args.cpu_seconds / 3):
                    break
            o0 = {"value": n0 / recon0 if recon0 > 0 else None, "samples": n0, "slowdown_vs_O3": (recon0 / n0) / (recon / cnt) if recon > 0 and n0 else None}
        except Exception as e:      # the -O0 build is optional
            o0 = {"error": str(e)}
        cpu = {"value": cnt / recon if recon > 0 else None, "unit": "reconstructions/s", "cores": 1, "kind": "port",
               "sample": "%d of the %d benchmarked samples, stages #7,#8,#11-#16,#20 only (%.2f s); whole oracle run incl. .lh "
                         "parse and the variableIdx map: %.1f /s" % (cnt, B, recon, cnt / whole if whole > 0 else 0),
               "cpu_model": model, "host_cores_available": os.cpu_count(), "compiled": "-O3", "port_at_O0": o0}

    # ---- the step that ends where SURVEY.md 8d's timed region ends: the final paths ON THE HOST (the reference prints every
    # path, LGM.cpp:3684-3689).  Every step: run -> ambi_batch_runs_to_host (pack kernels + ONE device-to-host copy of {counts, runs}
    # into pinned memory on a copy stream of the engine's own) -- double-buffered, so the copy of step i travels while step i+1
    # computes; the host waits for step i-1's slot while step i runs.  A sampled subset is expanded on the host and compared with
    # the downloaded paths.
    host_paths = None
    if world == 1 and args.host_paths:
        def host_loop(k):
            for i in range(k):
                batch.run(0, stream); batch.runs_to_host(1, i % 2, stream)
                if i >= 1:
                    batch.runs_wait((i - 1) % 2)
            return batch.runs_wait((k - 1) % 2)
        # untimed: the same loop once (the first copies of a process pay for the pinned blocks, the copy stream and the runtime's copy
        # path: ~8 ms in all, spread over the first dozen calls)
        host_loop(max(args.steps, 8)); batch.wait(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        view = host_loop(args.steps)
        batch.wait(); torch.cuda.synchronize()
        dth = time.perf_counter() - t1
        assert view["n_cells"] == total_cells and view["n_runs"] == n_runs
        batch.download()
        step_ = max(1, B // 64)
        for u in range(0, B, step_):
            assert batch.runs_unit_path((args.steps - 1) % 2, u).tolist() == batch.unit_path(u, 1).tolist(), "runs on the host differ from the downloaded path of unit %d" % u
        host_paths = {"ms_per_step": dth / args.steps * 1e3, "value": B * args.steps / dth, "unit": "reconstructions/s",
                      "bytes_per_step": int(view["bytes"]), "copied_bytes_per_step": int(view["copied_bytes"]), "runs_per_step": int(view["n_runs"]), "cells_per_step": int(view["n_cells"]),
                      "vs_step_in_hbm": (dth / args.steps) / (dt / args.steps),
                      "units_expanded_on_host_and_compared": len(range(0, B, step_)),
                      "how": "every step: ambi_batch_run, then ambi_batch_runs_to_host: ONE D2H copy of {lengths, run counts, runs} -- which the finish kernels write beside the path cells, into one of two blocks "
                             "alternating from run to run -- into pinned memory on the engine's copy stream; two pinned slots alternate; the host waits for step i-1's copy while step i runs; the last step's copy is waited for inside the timed region"}

    # ---- the same step on OTHER STREAMS (the engine picks side streams that dispatch beside whatever stream the caller passes:
    # DESIGN.md section 7): a stream of the caller's own from torch's pool; and, in a child process, the legacy default stream behind
    # a one-rank RCCL group (init_process_group + one all_reduce first), which is what every rank of an N > 1 job looks like.
    streams_leg = None
    if world == 1 and args.streams_leg:
        own = torch.cuda.Stream()
        for _ in range(max(args.warmup, 1)):
            batch.run(0, own.cuda_stream)
        batch.wait(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            batch.run(0, own.cuda_stream)
        batch.wait(); torch.cuda.synchronize()
        own_ms = (time.perf_counter() - t1) / args.steps * 1e3
        batch.download()
        assert [batch.unit_result(u)["path_indel_len"] for u in range(B)] == [r["path_indel_len"] for r in res]
        rccl_ms, rccl_note = None, None
        try:
            import subprocess
            env = dict(os.environ); env.pop("AMBI_BENCH_LIB", None); env["AMBI_STEPS_BATCH"] = str(B)
            pr = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "tools", "mode_steps.py"), "plain", str(args.steps), os.environ.get("AMBI_BENCH_LIB") or "-", str(args.sv_every), "rccl"],
                                capture_output=True, text=True, timeout=300, env=env)
            for line in pr.stdout.splitlines():
                if "ms per step" in line:
                    rccl_ms = float(line.split(":")[-1].split()[0])
            if rccl_ms is None:
                rccl_note = (pr.stderr or pr.stdout)[-300:]
        except Exception as e:
            rccl_note = str(e)
        streams_leg = {"own_stream_ms_per_step": own_ms, "rccl_one_rank_ms_per_step": rccl_ms, "default_stream_ms_per_step": ms_per_step,
                       "own_stream_vs_default": own_ms / ms_per_step, "rccl_one_rank_vs_default": (rccl_ms / ms_per_step) if rccl_ms else None,
                       "note": rccl_note or "rccl_one_rank: a child process (its own synthetic batch of the same seeds + 1000, boxes' run-to-run spread applies)"}
        batch.run(0, stream); batch.wait(); batch.download()

    # ---- ambi_batch_run_sharded, the multi-device driver below Python (north star: "samples shard across the 8 GPUs of one node"), with
    # 1 / 2 / 4 SHARES of the same 4096 samples on the ONE GPU of this box: every share has its own resident inputs, backend, stream and
    # host thread; what comes back per call is every unit's header and final path in run-length form (pinned memory), nothing is merged.
    # The host-side cost per call and the loss against one share are the figures that carry over to N devices (where the shares do not
    # compete for one GPU); no scaling curve is claimed from this.
    sharded = None
    if world == 1 and args.sharded_leg:
        sharded = {}
        for k in (1, 2, 4):
            sb = api.Batch(lib)
            sb.configure(target_lanes=args.target_lanes)
            for g, (lh, sols) in zip(graphs, files):
                sb.add_chromosome_sol(g, 0, sols[0])
            devs = [local_rank] * k
            for _ in range(max(args.warmup, 2)):
                sb.run_sharded(0, devs)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                sb.run_sharded(0, devs)
            dts = time.perf_counter() - t1
            for u in range(0, B, max(1, B // 32)):
                assert sb.unit_path(u, 1).tolist() == batch.unit_path(u, 1).tolist(), "sharded run: path of unit %d differs" % u
            sharded["%d_shares" % k] = {"ms_per_call": dts / args.steps * 1e3, "value": B * args.steps / dts}
            sb.close()
        sharded["unit"] = "reconstructions/s"
        sharded["four_shares_vs_one"] = sharded["4_shares"]["value"] / sharded["1_shares"]["value"]
        sharded["one_share_vs_step_in_hbm"] = sharded["1_shares"]["ms_per_call"] / ms_per_step
        sharded["how"] = ("ambi_batch_run_sharded(flags, [0] * k): per call every share runs its resident units on a stream of its own and copies headers + run-length "
                          "final paths to pinned memory (ONE D2H per share); the call returns when all shares have; k shares on ONE GPU compete for it")

    # ---- the step WITHOUT the order tables (AMBI_FLAG_LAZY_ORDERS: tables written on demand only).  The reference materialises
    # every topological order (LGM.cpp:3380-3409) and so does the headline step; in default mode nothing reads that table
    # (the scan reads the first orders the lattice stage unranks), so this is what the reconstructions alone cost.
    lazy = None
    if world == 1 and args.lazy:
        for _ in range(max(args.warmup, 1)):
            batch.run(api.FLAG_LAZY_ORDERS, stream)
        batch.wait(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            batch.run(api.FLAG_LAZY_ORDERS, stream)
        batch.wait(); torch.cuda.synchronize()
        dtl = time.perf_counter() - t1
        batch.download()
        assert [batch.unit_result(u)["path_indel_len"] for u in range(B)] == [r["path_indel_len"] for r in res]
        lazy = {"ms_per_step": dtl / args.steps * 1e3, "value": B * args.steps / dtl, "unit": "reconstructions/s",
                "what": "same batch, same results, AMBI_FLAG_LAZY_ORDERS: prepare (lattice, first 64 orders) -> scan -> finish; no plan / enumerate kernels, "
                        "the order tables are written only when asked for (ambi_batch_unit_orders) or when a scan runs out of its 64-order budget"}
        batch.run(0, stream); batch.wait(); batch.download()      # the tables again for the legs below

    # ---- two resident batches on two streams (double buffering): the latency-bound prepare / plan kernels of one batch
    # run under the HBM-bound enumerate kernel of the other.  Reported beside the headline value, which stays the plain
    # one-batch-after-the-other measurement; same units, every step a complete pass over one batch.
    pipelined = None
    if args.pipelined and world == 1:
        other = api.Batch(lib)
        other.configure(target_lanes=args.target_lanes)
        for g, (lh, sols) in zip(graphs, files):
            other.add_chromosome_sol(g, 0, sols[0])
        other.upload()
        s2 = torch.cuda.Stream()
        pair = [(batch, stream), (other, s2.cuda_stream)]
        other.run(0, s2.cuda_stream); other.wait()
        for k in range(2 * max(args.warmup, 1)):
            pair[k % 2][0].run(0, pair[k % 2][1])
        batch.wait(); other.wait(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(args.steps):
            pair[k % 2][0].run(0, pair[k % 2][1])
        batch.wait(); other.wait(); torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        other.download(); batch.download()
        assert all(other.unit_result(u)["status"] == 0 for u in range(B))
        assert other.unit_path(B - 1, 1).tolist() == batch.unit_path(B - 1, 1).tolist()
        pipelined = {"value": B * args.steps / dt2, "unit": "reconstructions/s", "ms_per_step": dt2 / args.steps * 1e3, "steps": args.steps,
                     "how": "two resident batches of %d samples, steps alternate between them on two HIP streams" % B}
        other.close()

    # ---- single-sample latency (north star: "single-sample latency speed-up vs reference single-thread CPU") ----
    # one 256-segment sample resident in HBM: launch of the whole chain -> results complete (host-synchronised), and
    # the same with the upload of the unit and the download of its path included; the CPU figure is the oracle on the
    # SAME sample (stages #7,#8,#11-#16,#20), so the ratio compares like with like.
    single = None
    if args.single_reps > 0 and world == 1:
        one = api.Batch(lib)
        one.add_chromosome_sol(graphs[0], 0, files[0][1][0])
        # The one-sample batch runs on a stream of the caller's own, not on the legacy default stream: its implicit ordering
        # against every other blocking stream costs ~10 us per run (profiles/tools/stream_latency.py: 102 us end to end on the
        # default stream, 88 on a stream of one's own; AMBI_BENCH_STREAM=default measures on the default stream).  The 4096-sample
        # legs stay on the default stream: a further stream created BEFORE the engine's own changes which of them share a hardware
        # queue (1.36 instead of 0.87 ms per step, profiles/r03_notes.md).
        batch_stream = stream
        if os.environ.get("AMBI_BENCH_STREAM") != "default":
            own_stream = torch.cuda.Stream()
            stream = own_stream.cuda_stream
        one.upload()
        one.run(0, stream); one.wait()
        for _ in range(10):
            one.run(0, stream); one.wait()
        # (a) launch -> RESULTS complete (path, breakpoints, output junctions in HBM): for a batch this small the engine
        #     reconstructs the sample in one kernel (order 0 straight from the DAG) and builds the order table behind it;
        #     the wait for the table is outside the timed span.  (b) launch -> everything complete, table included.
        span = 0.0
        for _ in range(args.single_reps):
            t1 = time.perf_counter()
            one.run(0, stream); one.wait_results()
            span += time.perf_counter() - t1
            one.wait()
        gpu_ms = span / args.single_reps * 1e3
        t1 = time.perf_counter()
        for _ in range(args.single_reps):
            one.run(0, stream); one.wait()
        gpu_all_ms = (time.perf_counter() - t1) / args.single_reps * 1e3
        # (c) END TO END, SURVEY.md 8d's region: the unit packed on the host -> its final path on the host.  upload writes the
        #     input image into pinned staging, run queues its ONE copy to HBM ahead of the kernels, the reconstruction kernel
        #     mirrors header + final paths + output junctions into a pinned mailbox, the host spins on a pinned word:
        #     no copy command behind the kernel.  (d) the same through wait + download of the whole result blob, order
        #     table complete.  (e) (c) with a NEW batch object per sample (create, .sol parse, pack, ..., destroy).
        reps2 = max(1, args.single_reps // 2)
        one.wait()
        for _ in range(5):
            one.upload(); one.run(0, stream); one.fetch_paths(); one.wait()
        span = 0.0
        for _ in range(reps2):
            t1 = time.perf_counter()
            one.upload(); one.run(0, stream); one.fetch_paths(); p_e2e = one.unit_path(0, 1)
            span += time.perf_counter() - t1
            one.wait()
        e2e_ms = span / reps2 * 1e3
        assert p_e2e.tolist() == batch.unit_path(0, 1).tolist()
        reps3 = max(1, args.single_reps // 4)
        t1 = time.perf_counter()
        for _ in range(reps3):
            one.upload(); one.run(0, stream); one.wait(); one.download(); one.unit_path(0, 1)
        pcie_ms = (time.perf_counter() - t1) / reps3 * 1e3
        assert one.unit_path(0, 1).tolist() == batch.unit_path(0, 1).tolist()
        span = 0.0
        for _ in range(reps3):
            t1 = time.perf_counter()
            fresh = api.Batch(lib)
            fresh.add_chromosome_sol(graphs[0], 0, files[0][1][0])
            fresh.upload(); fresh.run(0, stream); fresh.fetch_paths(); p_new = fresh.unit_path(0, 1)
            span += time.perf_counter() - t1
            fresh.close()
        new_ms = span / reps3 * 1e3
        assert p_new.tolist() == batch.unit_path(0, 1).tolist()
        # (f) (c) as a C caller of the ABI sees it: ambigram_amd/bin/e2e_probe (csrc/e2e_probe.cpp) runs the same four calls natively
        #     in a process of its own, on the same sample
        c_probe = None
        try:
            import subprocess
            exe = os.path.join(ROOT, "ambigram_amd", "bin", "e2e_probe")
            if os.path.exists(exe):
                one.wait(); torch.cuda.synchronize()
                pr = subprocess.run([exe, files[0][0], files[0][1][0], str(args.single_reps)], capture_output=True, text=True, timeout=120)
                if pr.returncode == 0:
                    c_probe = json.loads(pr.stdout.strip().splitlines()[-1])
                    assert c_probe["path_len"] == len(p_e2e)
        except Exception as e:      # the probe is optional
            c_probe = {"error": str(e)}
        single = {"gpu_ms": gpu_ms, "gpu_ms_order_table_included": gpu_all_ms, "e2e_ms": e2e_ms,
                  # (rounds 1-2 meaning of this key: upload + run + wait + download of the whole result blob, order table complete)
                  "gpu_ms_with_upload_and_download": pcie_ms, "e2e_ms_mailbox_path": e2e_ms,
                  "e2e_ms_c_caller": (c_probe["e2e_us_mean"] / 1e3) if c_probe and "e2e_us_mean" in c_probe else None, "c_caller": c_probe,
                  "e2e_ms_blob_download_and_table": pcie_ms, "e2e_ms_new_batch_object_incl_sol_parse": new_ms, "reps": args.single_reps,
                  "what": "e2e_ms: packed unit on the host -> final path on the host (upload, run, fetch_paths, unit_path through ctypes; e2e_ms_c_caller: the same four calls from C; SURVEY 8d's region; the order table "
                          "of the sample is written behind it); gpu_ms: launch -> reconstruction results complete, inputs resident in HBM (ambi_batch_wait_results)",
                  "sample": "sample 0 of the batch (1 unit, R = %d orders)" % res[0]["num_orders"]}
        if cpu is not None:
            best = None
            for _ in range(5):
                r = oracle_py.run_bfb(files[0][0], files[0][1])
                best = r["recon_seconds"] if best is None else min(best, r["recon_seconds"])
            single["cpu_ms"] = best * 1e3
            single["speedup"] = best * 1e3 / e2e_ms
            single["e2e_speedup"] = best * 1e3 / e2e_ms
            single["speedup_inputs_resident"] = best * 1e3 / gpu_ms
            # the CPU side materialises all R orders (allTopologicalOrders, LGM.cpp:3380-3409) inside its time; the GPU side prints the
            # same path from order 0 and writes the table BEHIND the published result: with the table waited for the ratio is
            single["speedup_with_order_table"] = best * 1e3 / gpu_all_ms
            single["speedup_with_order_table_and_blob_download"] = best * 1e3 / pcie_ms
            single["speedup_accounting"] = "speedup / e2e_speedup: packed unit on the host -> final path on the host, order table written behind it; speedup_with_order_table: inputs resident, launch -> order table complete (the CPU time includes its order table)"
            if single.get("e2e_ms_c_caller"):
                single["e2e_speedup_c_caller"] = best * 1e3 / single["e2e_ms_c_caller"]
            single["cpu_kind"] = "port (oracle, 1 core, best of 5)"
        single["stream"] = "a stream of the caller's own" if stream != batch_stream else "default stream"
        stream = batch_stream

    # ---- the step WITH the ILP assembly ("reported twice").  BFB_ILP (LGM.cpp:4397-4752) is where the
    # reference spends its non-solver time (O(n * numPat^2) coefficient loop, 313 s at 256 segments);
    # here the same rows come from the closed form: on the host (O(nnz), one core) or written by ambi_ilp_fill_kernel.
    ilp = None
    if world == 1:
        try:
            prep = batch.unit_prepare(0, args.segs)
            g0 = graphs[0]
            max_cn = float(sum(prep["seg_cn"][1:]))
            t1 = time.perf_counter()
            mh = api.IlpModel(lib, g0, 0, prep["seg_cn"], prep["junc_cn"], res[0]["bias"], max_cn)
            host_ms = (time.perf_counter() - t1) * 1e3
            md = api.IlpModel(lib, g0, 0, prep["seg_cn"], prep["junc_cn"], res[0]["bias"], max_cn, device=True)
            nnz, rows = mh.nnz, mh.n_rows
            fill_ms = md.kernel_ms
            mh.close(); md.close()
            ilp_bytes = 12 * nnz + 16 * rows            # int32 column + f64 coefficient per entry, two f64 bounds per row
            ilp = {"sample": "sample 0 (%d segments): %d rows, %d non-zeros" % (args.segs, rows, nnz),
                   "host_generator_ms_per_sample": host_ms, "device_fill_kernel_ms_per_sample": fill_ms,
                   "device_fill_GBps": (12 * nnz / (fill_ms * 1e-3) / 1e9) if fill_ms else None, "algorithmic_bytes_per_sample": ilp_bytes,
                   "reconstructions_per_s_without_assembly": value,
                   "reconstructions_per_s_with_host_assembly_1core": units_total / (dt / args.steps + units_total * host_ms * 1e-3),
                   "reconstructions_per_s_with_device_fill": (units_total / (dt / args.steps + units_total * fill_ms * 1e-3)) if fill_ms else None,
                   "note": "assembly of ONE model timed, charged once per sample; the reference's own assembly loop takes 313 s per sample at this size (SURVEY.md section 6)"}
        except Exception as e:
            ilp = {"error": str(e)}

    # ---- --all (LGM.cpp:3672-3695): every order of every sample evaluated -- the mode in which the enumeration is consumed.
    # Fused unrank + evaluate kernel over all (sample, 64-order chunk) work items, one launch per orientation pass.
    all_mode = None
    if args.all_steps > 0 and world == 1:
        batch.set_timing(True)
        batch.run(api.FLAG_ALL, stream); batch.wait()          # untimed: bitmap allocation
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.all_steps):
            batch.run(api.FLAG_ALL, stream); batch.wait()
        torch.cuda.synchronize()
        dta = (time.perf_counter() - t1) / args.all_steps
        batch.download()
        ev = sum(batch.unit_result(u)["evaluated"] for u in range(B))
        kms = batch.kernel_times().get("ambi_all_kernel")
        batch.set_timing(False)
        nvalid = sum(len(batch.all_orders(u, 0)) for u in range(0, B, max(1, B // 8)))
        rows_bytes = sum(r["num_orders"] * r["n_nodes"] for r in res)
        all_mode = {"orders_evaluated_per_step": ev, "ms_per_step": dta * 1e3, "orders_evaluated_per_s": ev / dta,
                    "reconstructions_per_s": B / dta, "all_kernel_ms_both_passes": kms,
                    "orders_evaluated_per_s_kernel_only": (ev / (kms * 1e-3)) if kms else None,
                    "hbm_algorithmic_bytes": {"validity_bitmap": ev // 8, "order_table_not_read": 0, "note": "orders are unranked from the automaton in L2/LDS, breakpoint cells live in LDS: the kernel is bound by LDS latency / issue, not by HBM"},
                    "equivalent_table_read_GBps": (rows_bytes / (kms * 1e-3) / 1e9) if kms else None,
                    "valid_orders_in_sampled_units": nvalid, "steps": args.all_steps}

    out = {
        "metric": "BFB reconstructions/sec (synthetic .lh, 256 seg) at 1/2/4/8 MI355X",
        "value": value, "unit": "reconstructions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8/int16 (node ids, stored as 5-bit fields of the order table / breakpoint cells; f64 copy numbers)", "data": "synthetic",
        "config": {"workload": "synthetic %d-seg/%d-junc .lh, %s DAG tier K=%d, planted .sol, %s; every %s sample carries 2 deletions + 1 duplication that edit the path (indelBFB with edits inside the timed region); %d samples per GPU resident in HBM"
                               % (args.segs, args.juncs, args.tier, args.K, "--all mode" if args.mode == "all" else "default CLI mode",
                                  ("%d-th" % args.sv_every) if args.sv_every > 0 else "no", B),
                   "samples_per_gpu": B, "orders_per_sample": res[0]["num_orders"], "parallelism": "samples sharded over %d GPU(s), results stay in each GPU's HBM%s" % (world, {0: " (no data-path collective)", 1: "; ONE RCCL gather of the last batch's paths (run-length form, expanded on rank 0) at the end of the timed steps", 2: "; one RCCL gather of the paths (run-length form) to rank 0 at the end of every step"}[gather_mode])},
        "roofline": roofline, "cpu_baseline": cpu, "single_sample": single, "pipelined": pipelined,
        "paths_on_host": host_paths, "streams": streams_leg, "sharded_on_one_gpu": sharded,
        "own_stream_ms_per_step": streams_leg["own_stream_ms_per_step"] if streams_leg else None,
        "rccl_one_rank_ms_per_step": streams_leg["rccl_one_rank_ms_per_step"] if streams_leg else None,
        "step_without_table_ms": lazy["ms_per_step"] if lazy else None, "step_without_table": lazy,
        "ilp_assembly": ilp, "all_mode": all_mode,
    }
    print(json.dumps(out))
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
