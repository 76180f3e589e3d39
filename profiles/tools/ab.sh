#!/bin/bash
# A/B runs of bench.py on ONE box (boxes of the pool differ by several per cent, so only runs of one call compare):
#   bash profiles/tools/ab.sh <reps> "ENV1=.. ENV2=.." "ENV.." ...
# prints reconstructions/s, ms per step and the enumerate / finish kernel times of every configuration, interleaved.
export AMBI_EXPERIMENTS=1   # the engine honours its AMBI_* switches only with this
reps=$1; shift
for r in $(seq 1 $reps); do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    env $cfg timeout -k 10 300 python3 bench.py --cpu-seconds 0 --single-reps 0 --all-steps 0 --pipelined 0 --host-paths 0 --streams-leg 0 --sharded-leg 0 $AB_ARGS > gpurun_out/ab_${i}_${r}.log 2>&1
    python3 - "$cfg" gpurun_out/ab_${i}_${r}.log <<'EOF'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = d["roofline"]["all_kernels_ms"]
print("%-60s %8d /s  %.4f ms  enum %.3f  finish %.3f  prepare %.3f" % (sys.argv[1], d["value"], d["ms_per_step"], k["ambi_enumerate_kernel"], k["ambi_finish_kernel"], k["ambi_prepare_kernel"]))
EOF
  done
done
