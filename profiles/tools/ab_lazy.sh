#!/bin/bash
# like ab.sh, printing also the step without order tables (AMBI_FLAG_LAZY_ORDERS) and the scan kernel's time
export AMBI_EXPERIMENTS=1   # the engine honours its AMBI_* switches only with this
reps=$1; shift
for r in $(seq 1 $reps); do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    env $cfg timeout -k 10 300 python3 bench.py --cpu-seconds 0 --single-reps 0 --all-steps 0 --pipelined 0 --lazy 1 > gpurun_out/abl_${i}_${r}.log 2>&1
    python3 - "$cfg" gpurun_out/abl_${i}_${r}.log <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = d["roofline"]["all_kernels_ms"]
print("%-50s step %.4f ms   without tables %.4f ms   first %.3f  enum %.3f" % (sys.argv[1], d["ms_per_step"], d["step_without_table_ms"], k.get("ambi_first_kernel", 0), k["ambi_enumerate_kernel"]))
PY
  done
done
