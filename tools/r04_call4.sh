#!/bin/bash
# round 4, GPU call 4: parity (kernel tests), 4- vs 8-wave two-block layer-1 kernel, 128 x 128 weight-gradient tile on small maps, step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c4; rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 300 2>&1 | tee $O/tests.log | tail -15; rc=${PIPESTATUS[0]}
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out: stopping"; exit 1; fi
echo "== tests rc $rc"
run() { name=$1; shift; echo "== $name"; timeout -k 10 420 "$@" 2>&1 | grep -v amdgpu | tee $O/$name.log; [ ${PIPESTATUS[0]} -eq 0 ] || exit 1; }
run sweep_halo_waves python -u tools/bench_conv.py --sweep MSCL_HALO_WAVES=8,4 --modes fwd,dgrad --only l1_64_64,l1n2,l1n4
run sweep_big_minm python -u tools/bench_conv.py --sweep MSCL_WGRAD_BIG_MINM=-,512 --modes wgrad --only l4_,l3_,neck_333,neck_133,fpn,flow_l4
for v in 8 4 8 4; do
  echo "== bench HALO_WAVES=$v"
  MSCL_HALO_WAVES=$v timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-variants > $O/bench_$v.json 2> $O/bench_$v.err || { tail -5 $O/bench_$v.err; exit 1; }
  python - $O/bench_$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('bench', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['roofline']['step_frac'],4))
PY
done
