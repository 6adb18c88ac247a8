#!/bin/bash
# round 4, GPU call 14: window-resident stem kernel: parity, alone (A/B against the implicit-GEMM kernel), step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c14; rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 300 -k "stem" 2>&1 | tee $O/tests.log | tail -15; rc=${PIPESTATUS[0]}
echo "== tests rc $rc"; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -u tools/bench_conv.py --modes fwd --only stem_rgb_pairw --sweep MSCL_STEM=0,- 2>&1 | grep -v amdgpu | tee $O/sweep_r18.log
timeout -k 10 300 python -u tools/bench_conv.py --r50 --modes fwd --only r50_stem_pairw --sweep MSCL_STEM=0,- --iters 10 2>&1 | grep -v amdgpu | tee $O/sweep_r50.log
for v in 1 2 3; do for s in 0 -; do
  if [ $s = 0 ]; then export MSCL_STEM=0; else unset MSCL_STEM; fi
  timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-variants > $O/bench_${s}_$v.json 2> $O/bench_${s}_$v.err || exit 1
  python - $O/bench_${s}_$v.json $s <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('stem', sys.argv[2], round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['roofline']['also'][0]['frac'],4), d['final_loss'])
PY
done; done
