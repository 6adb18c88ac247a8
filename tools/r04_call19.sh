#!/bin/bash
# round 4, GPU call 19: stagger of the two layer-1 blocks of a CU
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c19; rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
timeout -k 10 300 python -u tools/bench_conv.py --modes fwd,dgrad --only l1_64_64 --sweep MSCL_HALO_STAGGER=0,100,200,300,400,600 2>&1 | grep -v amdgpu | tee $O/sweep.log
