#!/bin/bash
# A/B of one environment switch on the whole step, alternating rounds inside one call.
# usage: VAR=MSCL_PP VALS="0 1 2" N=2 [ARGS=--deterministic] bash tools/r03_step_ab.sh
set -o pipefail
mkdir -p gpurun_out
VAR=${VAR:-MSCL_PP}; VALS=${VALS:-"0 1"}; N=${N:-2}; ARGS=${ARGS:-}
for i in $(seq 1 $N); do
  for v in $VALS; do
    env $VAR=$v timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $ARGS 2> gpurun_out/ab_err.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('$VAR=$v $ARGS', round(d['value'], 1), 'clip-pairs/s', round(d['ms_per_step'], 3), 'ms', 'step_frac', round(r.get('step_frac', 0), 4))" || { tail -5 gpurun_out/ab_err.log; exit 1; }
  done
done
