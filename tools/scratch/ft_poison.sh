#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/poison
fails=0
for i in $(seq 1 24); do
  MSCL_TEST_POISON=2 timeout -k 10 200 python3 -m pytest tests/test_finetune_gpu.py -q -m gpu -x > gpurun_out/poison/ftl_$i.log 2>&1 || { fails=$((fails+1)); echo "run $i FAILED"; grep -E "^(FAILED|E  )" gpurun_out/poison/ftl_$i.log | head -8; }
done
echo "finetune file under POISON: $fails failures of 24 fresh processes"
