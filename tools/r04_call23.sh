#!/bin/bash
# round 4, GPU call 23: loss-phase kernels with their loads issued ahead of the staging / barriers
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c23; rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 300 -k "${TESTS:-nce or lmcl or linear or l2norm or enqueue}" 2>&1 | tee $O/tests.log | tail -5; rc=${PIPESTATUS[0]}
echo "== tests rc $rc"; [ $rc -eq 0 ] || exit 1
for l in prev new prev new; do
  if [ $l = prev ]; then export MSCL_LIB=$R/mscl_amd/csrc/libmscl_hip_ab.so; else unset MSCL_LIB; fi
  echo "== $l"; timeout -k 10 300 python -u tools/bench_conv.py --modes fwd,dgrad --only l3_256_256,l4_512_512,l4_256_512_s2,neck_333_p1,neck_lat_l4 2>&1 | grep -v amdgpu | tee -a $O/conv_$l.log
done
for v in 1 2 3; do for l in prev new; do
  if [ $l = prev ]; then export MSCL_LIB=$R/mscl_amd/csrc/libmscl_hip_ab.so; else unset MSCL_LIB; fi
  timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-variants > $O/bench_${l}_$v.json 2> $O/bench_${l}_$v.err || exit 1
  python - $O/bench_${l}_$v.json $l <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['roofline']['also'][0]['frac'],4), d['final_loss'])
PY
done; done
