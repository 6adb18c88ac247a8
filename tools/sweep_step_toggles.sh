#!/bin/bash
# A/B of the step-level tuning switches against the default, alternating inside ONE gpurun call (box-to-box spread is larger
# than most of these effects).  usage: bash tools/sweep_step_toggles.sh > gpurun_out/toggles.log
run() { env $1 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%-28s %7.1f clip-pairs/s  %6.3f ms' % ('$1' or 'default', d['value'], d['ms_per_step']))"; }
for v in "MSCL_WGRAD_TILE=64" "MSCL_WGRAD_TILE=128" "MSCL_IGEMM_WIDE=0" "MSCL_IGEMM_WIDE=2" "MSCL_LOSS_FORK=0" "MSCL_FAST_STAGES=3" "MSCL_PAIR_STEM=0" "MSCL_WGRAD_HALO=0" "MSCL_HALO_PERSIST=1" "MSCL_FUSE_BN_REDUCE=1" "MSCL_WGRAD_NCOL=192"; do
  run ""; run "$v"
done
run ""
