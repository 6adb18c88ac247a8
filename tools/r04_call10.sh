#!/bin/bash
# round 4, GPU call 10: slab sums of the window-resident weight gradient on the idle key stream: parity (model tests), step A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c10; rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
timeout -k 10 1000 python -u -m pytest tests/test_model_gpu.py tests/test_finetune_gpu.py -m gpu -x -q --timeout 400 2>&1 | tee $O/tests.log | tail -6; rc=${PIPESTATUS[0]}
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out: stopping"; exit 1; fi
echo "== tests rc $rc"
[ $rc -eq 0 ] || exit 1
for v in 0 1 0 1 0 1; do
  MSCL_AUX_REDUCE=$v timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-variants > $O/bench_$v.json 2> $O/bench_$v.err || { tail -5 $O/bench_$v.err; exit 1; }
  python - $O/bench_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('aux', sys.argv[2], round(d['value'],1), round(d['ms_per_step'],3), d['final_loss'], flush=True)
PY
done
MSCL_AUX_REDUCE=1 timeout -k 10 300 python -u bench.py --no-cpu-baseline --deterministic --no-variants > $O/bench_det.json 2> $O/bench_det.err && python - $O/bench_det.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('det', round(d['value'],1), d['final_loss'])
PY
run() { name=$1; shift; echo "== $name"; timeout -k 10 420 "$@" 2>&1 | grep -v amdgpu | tee $O/$name.log; }
run chain_times python -u tools/chain_times.py
