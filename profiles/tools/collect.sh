#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box from the repo root):
#   gpurun --timeout 900 -- 'bash profiles/tools/collect.sh r01'
# Every pass is its own rocprofv3 run (counters never share a run with traces other than --kernel-trace); the rocpd
# databases go to gpurun_out/prof_<tag>/, the summaries (CSV / JSON) to gpurun_out/profiles_<tag>/ for copying into
# profiles/.
export AMBI_EXPERIMENTS=1   # the engine honours its AMBI_* switches only with this
set -eo pipefail
TAG=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
SUM=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$SUM"
export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --cpu-seconds 0 --single-reps 0 --pipelined 0 --all-steps 0 --lazy 0 --host-paths 0 --streams-leg 0 --sharded-leg 0 --steps 10 --warmup 3"   # batch leg only: the single-sample and two-batch legs would mix other launches into the per-kernel means

pass() {   # name, rocprofv3 flags ...
    local name=$1; shift
    rm -rf "$OUT/$name"
    (cd /tmp && timeout -k 10 240 rocprofv3 "$@" -d "$OUT/$name" -- $BENCH > "$OUT/$name.log" 2>&1)
    find "$OUT/$name" -name '*_results.db' | head -1
}

db=$(pass stats --kernel-trace --stats)
python3 profiles/summarize_rocpd.py stats "$db" "$SUM/${TAG}_bench_b4096_kernel_stats.csv"
db=$(AMBI_OVERLAP_BACK=0 AMBI_BUILD_IN_EMIT=0 pass serial --kernel-trace --stats)
python3 profiles/summarize_rocpd.py stats "$db" "$SUM/${TAG}_bench_b4096_serial_kernel_stats.csv"
wr=$(pass pmc_write --kernel-trace --pmc WRITE_SIZE)
python3 profiles/summarize_rocpd.py pmc "$wr" "$SUM/${TAG}_bench_b4096_pmc_write.csv" > /dev/null
rd=$(pass pmc_fetch --kernel-trace --pmc FETCH_SIZE)
python3 profiles/summarize_rocpd.py pmc "$rd" "$SUM/${TAG}_bench_b4096_pmc_fetch.csv" > /dev/null
python3 profiles/summarize_rocpd.py traffic "$wr" "$rd" "$SUM/traffic_${TAG}.json" --batch 4096 --workload "256/512/wide/K19/sv8"
db=$(pass pmc_sq --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS)
python3 profiles/summarize_rocpd.py pmc "$db" "$SUM/${TAG}_bench_b4096_pmc_sq.csv" > /dev/null
# the --all leg (fused unrank + evaluate kernel) and the ILP fill kernel: their own kernel-trace passes
BENCH="python3 $ROOT/bench.py --cpu-seconds 0 --single-reps 0 --pipelined 0 --lazy 0 --host-paths 0 --streams-leg 0 --sharded-leg 0 --all-steps 2 --steps 2 --warmup 1 --batch 1024"
db=$(pass all_mode --kernel-trace --stats)
python3 profiles/summarize_rocpd.py stats "$db" "$SUM/${TAG}_bench_b1024_allmode_kernel_stats.csv"
# group-memory side of the --all kernel (one thread per order): instruction counts and bank conflicts
db=$(pass all_mode_lds --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT)
python3 profiles/summarize_rocpd.py pmc "$db" "$SUM/${TAG}_bench_b1024_allmode_pmc_lds.csv" > /dev/null || true
BENCH="python3 $ROOT/profiles/tools/express_latency.py"
db=$(pass express --kernel-trace --stats)
python3 profiles/summarize_rocpd.py stats "$db" "$SUM/${TAG}_single_sample_express_kernel_stats.csv"
# timeline of a step (start / end of every kernel from the step's first one) from the kernel-trace pass
python3 profiles/tools/timeline.py "$(find "$OUT/stats" -name '*_results.db' | head -1)" 2 > "$SUM/${TAG}_timeline.txt" || true
# the same step on a stream of the caller's own (torch pool) and behind a one-rank RCCL group: timelines
BENCH="python3 $ROOT/profiles/tools/mode_steps.py plain 6 - 8 own"
db=$(pass own_stream --kernel-trace)
python3 profiles/tools/timeline.py "$db" 1 > "$SUM/${TAG}_timeline_own_stream.txt" || true
# the ILP fill kernel alone: duration, WRITE_SIZE, instruction counts
BENCH="python3 $ROOT/profiles/tools/ilp_fill_probe.py 2"
db=$(pass ilp_stats --kernel-trace --stats)
python3 profiles/summarize_rocpd.py stats "$db" "$SUM/${TAG}_ilp_fill_kernel_stats.csv" > /dev/null
db=$(pass ilp_write --kernel-trace --pmc WRITE_SIZE)
python3 profiles/summarize_rocpd.py pmc "$db" "$SUM/${TAG}_ilp_fill_pmc_write.csv" > /dev/null
db=$(pass ilp_sq --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY)
python3 profiles/summarize_rocpd.py pmc "$db" "$SUM/${TAG}_ilp_fill_pmc_sq.csv" > /dev/null
# the lean finish kernel alone at full residency (lazy steps, one workgroup per unit): what bounds it
BENCH="python3 $ROOT/profiles/tools/mode_steps.py lazy 4 - 0"
db=$(AMBI_FINISH_GRID=4096 pass lean_sq --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_WAIT_INST_ANY SQ_WAIT_ANY)
python3 profiles/summarize_rocpd.py pmc "$db" "$SUM/${TAG}_lean_full_grid_pmc_sq.csv" > /dev/null
tail -2 "$OUT/stats.log"
ls -la "$SUM"
