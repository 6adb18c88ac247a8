#!/bin/bash
# round 3: parity of the ping-pong shared-tap conv kernel + A/B against the two-barrier kernels, one process per step
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "pp_forced or real_layer" > gpurun_out/pp_tests.log 2>&1
rc=$?; tail -5 gpurun_out/pp_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/bench_conv.py --sweep MSCL_PP=0,2 --only l2_128_128,l3_256_256,l4_512_512,fpn_133,neck_333 --modes fwd,dgrad --iters 20 > gpurun_out/pp_sweep.log 2>&1
rc=$?; cat gpurun_out/pp_sweep.log; [ $rc -ne 0 ] && exit $rc
