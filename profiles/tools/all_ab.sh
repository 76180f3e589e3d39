#!/bin/bash
# A/B of the --all leg of bench.py on ONE box:  bash profiles/tools/all_ab.sh "ENV=.." "ENV=.." ...
export AMBI_EXPERIMENTS=1   # the engine honours its AMBI_* switches only with this
for r in 1 2; do
for cfg in "$@"; do
env $cfg timeout -k 10 300 python3 bench.py --cpu-seconds 0 --single-reps 0 --pipelined 0 --steps 3 --all-steps 2 > gpurun_out/allb.log 2>&1
python3 -c "
import json
d=json.loads(open('gpurun_out/allb.log').read().strip().splitlines()[-1]); a=d['all_mode']; print('%-50s %7.1f M orders/s  %7.1f ms' % ('$cfg', a['orders_evaluated_per_s']/1e6, a['ms_per_step']))"
done
done
