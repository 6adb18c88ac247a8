#!/bin/bash
# round 4, GPU call 20: layer-1 kernel with two-tap ring stages (one barrier per stage)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c20; rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 300 -k "halo_two_blocks" 2>&1 | tee $O/tests.log | tail -5; rc=${PIPESTATUS[0]}
echo "== tests rc $rc"; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -u tools/bench_conv.py --modes fwd,dgrad --only l1_64_64,l1n4_64_64 --sweep MSCL_HALO_TPS=1,2 2>&1 | grep -v amdgpu | tee $O/sweep.log
for v in 1 2 3; do for s in 1 2; do
  MSCL_HALO_TPS=$s timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-variants > $O/bench_${s}_$v.json 2> $O/bench_${s}_$v.err || exit 1
  python - $O/bench_${s}_$v.json $s <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('tps', sys.argv[2], round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['roofline']['also'][0]['frac'],4), d['final_loss'])
PY
done; done
