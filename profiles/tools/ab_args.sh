#!/bin/bash
# like ab.sh, with extra bench.py arguments:  bash profiles/tools/ab_args.sh <reps> "<bench args>" "ENV.." "ENV.." ...
export AMBI_EXPERIMENTS=1   # the engine honours its AMBI_* switches only with this
reps=$1; shift; args=$1; shift
for r in $(seq 1 $reps); do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    env $cfg timeout -k 10 400 python3 bench.py --cpu-seconds 0 --single-reps 0 $args > gpurun_out/abx_${i}_${r}.log 2>&1
    python3 - "$cfg" gpurun_out/abx_${i}_${r}.log <<'EOF'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = d["roofline"]["all_kernels_ms"]
print("%-40s %9d /s  %.4f ms  enum %.3f  build %.3f  GB/s %d" % (sys.argv[1], d["value"], d["ms_per_step"], k["ambi_enumerate_kernel"], k["ambi_blocks_build_kernel"], d["roofline"]["achieved"] or 0))
EOF
  done
done
