#!/bin/bash
# round 4, GPU call 8: the whole GPU suite, smoke, the bench line's rccl leg on a one-rank RCCL group (both gradient collectives)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c8; rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 400 2>&1 | tee $O/tests.log | tail -8; rc=${PIPESTATUS[0]}
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out: stopping"; exit 1; fi
echo "== tests rc $rc"
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
for coll in all_reduce rs_ag; do
  echo "== forced one-rank RCCL group, MSCL_GRAD_COLLECTIVE=$coll"
  MSCL_FORCE_DIST=1 MSCL_GRAD_COLLECTIVE=$coll timeout -k 10 300 python -u bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_forced_$coll.json 2> $O/bench_forced_$coll.err || { tail -8 $O/bench_forced_$coll.err; exit 1; }
  python - $O/bench_forced_$coll.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('bench', round(d['value'],1), d['config']['launch'][:60], json.dumps(d.get('rccl')))
PY
done
for v in 1 2 3; do
  timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-variants > $O/bench_$v.json 2> $O/bench_$v.err || exit 1
  python - $O/bench_$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('bench', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['roofline']['step_frac'],4))
PY
done
