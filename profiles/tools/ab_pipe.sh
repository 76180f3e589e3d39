#!/bin/bash
# like ab.sh, for the two-resident-batches leg: prints plain and pipelined ms per step of every configuration, interleaved
export AMBI_EXPERIMENTS=1   # the engine honours its AMBI_* switches only with this
reps=$1; shift
for r in $(seq 1 $reps); do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    env $cfg timeout -k 10 300 python3 bench.py --cpu-seconds 0 --single-reps 0 --all-steps 0 --pipelined 1 $AB_ARGS > gpurun_out/abp_${i}_${r}.log 2>&1
    python3 - "$cfg" gpurun_out/abp_${i}_${r}.log <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-60s plain %.4f ms   pipelined %.4f ms" % (sys.argv[1], d["ms_per_step"], d["pipelined"]["ms_per_step"]))
PY
  done
done
