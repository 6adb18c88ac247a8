#!/bin/bash
# round 4, GPU call 11: conv_pp prologue reorder: parity + race screen, alone, step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c11; rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 300 -k "pp_forced or real_layer or fwd_dgrad_wgrad" 2>&1 | tee $O/tests.log | tail -4; rc=${PIPESTATUS[0]}
echo "== tests rc $rc"; [ $rc -eq 0 ] || exit 1
timeout -k 10 420 python -u tools/bench_conv.py --modes fwd,dgrad --only l2_128_128,l3_256_256,l4_512_512,sepc_128,fpn_133,neck_333,neck_133 2>&1 | grep -v amdgpu | tee $O/conv.log
for v in 1 2 3; do
  timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-variants > $O/bench_$v.json 2> $O/bench_$v.err || exit 1
  python - $O/bench_$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('bench', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['roofline']['also'][0]['frac'],4))
PY
done
