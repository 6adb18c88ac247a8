#!/bin/bash
# round 4, GPU call 25: HIP path vs fp32 oracle over 200 optimizer steps on the final kernels (default, non-deterministic mode)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c25; rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
timeout -k 10 1100 python -u tools/train_curve.py --steps 200 --batch 2 --every 20 2> $O/curve.err | tee $O/training_curve.md | tail -14
