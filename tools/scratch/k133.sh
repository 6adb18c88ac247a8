#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -k "wgrad or halo" -x 2>&1 | tail -3 || exit 1
for s in r50_l1_c2_133 r50_l2_c2_133 r50_l3_c2_133 r50_l4_c2_133; do
  echo "new : $(python3 tools/bench_conv.py --r50 --only $s --iters 10 --modes wgrad 2>/dev/null | grep r50_)"
  echo "old : $(MSCL_WGRAD_HALO_MIN=100000000 python3 tools/bench_conv.py --r50 --only $s --iters 10 --modes wgrad 2>/dev/null | grep r50_)"
done
python3 tools/bench_trunk.py --r50 2>/dev/null | cut -c1-120
python3 tools/bench_trunk.py --r50 2>/dev/null | cut -c1-120
