#!/bin/bash
# second sweep: grid caps of the BatchNorm passes, split-K rule, ReLU-mask source -- each against the default, alternating in one call
run() { env $1 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%-30s %7.1f clip-pairs/s  %6.3f ms' % ('$1' or 'default', d['value'], d['ms_per_step']))"; }
for v in "MSCL_BN_FWD_CAP=1024" "MSCL_BN_FWD_CAP=4096" "MSCL_BN_RED_CAP=512" "MSCL_BN_RED_CAP=2048" "MSCL_BN_APPLY_CAP=1024" "MSCL_BN_APPLY_CAP=4096" "MSCL_KSPLIT_TARGET=320" "MSCL_KSPLIT_TARGET=640" "MSCL_KSPLIT_MINSTEPS=8" "MSCL_KSPLIT_MINSTEPS=18" "MSCL_MASK_FROM_Y_MIN=0" "MSCL_MASK_FROM_Y_MIN=1073741824"; do
  run ""; run "$v"
done
run ""
