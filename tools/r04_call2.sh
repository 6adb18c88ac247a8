#!/bin/bash
# round 4, GPU call 2: the sliced window-resident weight gradient (parity, A/B against the round-3 dispatch, step)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c2; rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -u -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_data_gpu.py -m gpu -x -q --timeout 300 2>&1 | tee $O/tests.log | tail -15; rc=${PIPESTATUS[0]}
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out: stopping"; exit 1; fi
echo "== tests rc $rc"
run() { name=$1; shift; echo "== $name"; timeout -k 10 420 "$@" 2>&1 | grep -v amdgpu | tee $O/$name.log; [ ${PIPESTATUS[0]} -eq 0 ] || exit 1; }
run sweep_halo_maxc python -u tools/bench_conv.py --sweep MSCL_WGRAD_HALO_MAXC=64,128,512 --modes wgrad --only l1_64_64,l2_128_128,l3_256_256,sepc_128,neck_333_p1
run chain_times python -u tools/chain_times.py
for v in 64 512 64 512; do
  echo "== bench MAXC=$v"
  MSCL_WGRAD_HALO_MAXC=$v timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-variants > $O/bench_$v.json 2> $O/bench_$v.err || { tail -5 $O/bench_$v.err; exit 1; }
  python - $O/bench_$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('bench', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['roofline']['step_frac'],4))
PY
done
