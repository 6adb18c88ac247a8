#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/poison
fails=0
for i in $(seq 1 16); do
  MSCL_TEST_POISON=2 timeout -k 10 300 python3 -m pytest tests/test_data_gpu.py tests/test_finetune_gpu.py -q -m gpu > gpurun_out/poison/df_$i.log 2>&1 || { fails=$((fails+1)); echo "run $i FAILED"; grep -E "^(FAILED|E  )" gpurun_out/poison/df_$i.log | head -8; }
done
echo "data + finetune files under POISON: $fails failures of 16 fresh processes"
