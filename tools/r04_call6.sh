#!/bin/bash
# round 4, GPU call 6: conv_pp2 (two blocks per CU): parity, alone, inside the step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c6; rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 300 -k "pp_forced or real_layer" 2>&1 | tee $O/tests.log | tail -15; rc=${PIPESTATUS[0]}
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out: stopping"; exit 1; fi
echo "== tests rc $rc"
[ $rc -eq 0 ] || exit 1
run() { name=$1; shift; echo "== $name"; timeout -k 10 420 "$@" 2>&1 | grep -v amdgpu | tee $O/$name.log; [ ${PIPESTATUS[0]} -eq 0 ] || exit 1; }
run sweep_pp_blocks python -u tools/bench_conv.py --sweep MSCL_PP_BLOCKS=1,2 --modes fwd,dgrad --only l2_128_128,l3_256_256,l4_512_512,sepc_128,fpn_133,neck_333_p1,neck_333_p2,neck_133_p1
for v in 1 2 1 2; do
  echo "== bench PP_BLOCKS=$v"
  MSCL_PP_BLOCKS=$v timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-variants > $O/bench_$v.json 2> $O/bench_$v.err || { tail -5 $O/bench_$v.err; exit 1; }
  python - $O/bench_$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('bench', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['roofline']['step_frac'],4), round(d['roofline']['also'][0]['frac'],4))
PY
done
