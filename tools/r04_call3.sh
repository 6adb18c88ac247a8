#!/bin/bash
# round 4, GPU call 3: full parity after the deletions, the two-blocks-per-CU layer-1 kernel (parity, alone, inside the step)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c3; rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tee $O/tests.log | tail -15; rc=${PIPESTATUS[0]}
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out: stopping"; exit 1; fi
echo "== tests rc $rc"
run() { name=$1; shift; echo "== $name"; timeout -k 10 420 "$@" 2>&1 | grep -v amdgpu | tee $O/$name.log; [ ${PIPESTATUS[0]} -eq 0 ] || exit 1; }
for r in 4 3 2; do
MSCL_HALO_RING=$r run sweep_halo_blocks_ring$r python -u tools/bench_conv.py --sweep MSCL_HALO_BLOCKS=1,2 --modes fwd,dgrad --only l1_64_64,l1n2,l1n4
done
for v in 1 2 1 2; do
  echo "== bench HALO_BLOCKS=$v"
  MSCL_HALO_BLOCKS=$v timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-variants > $O/bench_$v.json 2> $O/bench_$v.err || { tail -5 $O/bench_$v.err; exit 1; }
  python - $O/bench_$v.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('bench', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['roofline']['step_frac'],4))
PY
done
MSCL_HALO_BLOCKS=2 run chain_times_b2 python -u tools/chain_times.py
