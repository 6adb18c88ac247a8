#!/bin/bash
# round 4, GPU call 17: what the statistics epilogue of the stem kernel costs: no atomics (probe 1), 16 slots (probe 2), default, none
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c17; rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
for v in 1 2; do for l in default probe1 probe2; do
  if [ $l = default ]; then unset MSCL_LIB; else export MSCL_LIB=$R/mscl_amd/csrc/libmscl_hip_$l.so; fi
  echo "== $l"
  timeout -k 10 300 python -u tools/bench_conv.py --modes fwd --only stem_rgb_pairw 2>&1 | grep -v amdgpu | tee -a $O/conv_$l.log
  timeout -k 10 300 python -u tools/bench_conv.py --r50 --modes fwd --only r50_stem_pairw,r50_l1_c3_64_256 --iters 10 2>&1 | grep -v amdgpu | tee -a $O/conv_$l.log
done; done
unset MSCL_LIB
echo "== no statistics"
timeout -k 10 300 python -u tools/bench_conv.py --modes fwd --only stem_rgb_pairw --no-stats 2>&1 | grep -v amdgpu | tee -a $O/conv_nostats.log
timeout -k 10 300 python -u tools/bench_conv.py --r50 --modes fwd --only r50_stem_pairw,r50_l1_c3_64_256 --iters 10 --no-stats 2>&1 | grep -v amdgpu | tee -a $O/conv_nostats.log
