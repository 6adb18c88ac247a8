#!/bin/bash
# round 4, GPU call 12: full suite on the final code, smoke, pool split A/B in the step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c12; rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 400 2>&1 | tee $O/tests.log | tail -4; rc=${PIPESTATUS[0]}
echo "== tests rc $rc"; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
for v in 1 2 3; do for s in 0 1; do
  MSCL_POOL_SPLIT=$s timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-variants > $O/bench_${s}_$v.json 2> $O/bench_${s}_$v.err || exit 1
  python - $O/bench_${s}_$v.json $s <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('pool split', sys.argv[2], round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['roofline']['also'][0]['frac'],4))
PY
done; done
