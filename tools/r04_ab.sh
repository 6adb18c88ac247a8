#!/bin/bash
# Round-4 A/B driver (one gpurun call): the current build against another build of the library, alternating in one call.
#   cp mscl_amd/csrc/libmscl_hip.so mscl_amd/csrc/libmscl_hip_ab.so        (the "prev" arm: the build BEFORE the change)
#   ... edit, bash mscl_amd/csrc/build.sh ...
#   gpurun --timeout 1200 -- 'TESTS="stem or halo" SHAPES=l1_64_64,stem_rgb_pairw MODES=fwd,dgrad STEP=1 bash tools/r04_ab.sh'
# TESTS: pytest -k filter for tests/test_kernels_gpu.py (run first; a failure stops the call); SHAPES / MODES: tools/bench_conv.py
# shapes and directions timed alone for both arms, twice; STEP=1: three alternating pairs of bench.py runs.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 300 -k "${TESTS:-conv}" 2>&1 | tee $O/tests.log | tail -5; rc=${PIPESTATUS[0]}
echo "== tests rc $rc"; [ $rc -eq 0 ] || exit 1
if [ -n "$SHAPES" ]; then
for v in 1 2; do for l in prev new; do
  if [ $l = prev ]; then export MSCL_LIB=$R/mscl_amd/csrc/libmscl_hip_ab.so; else unset MSCL_LIB; fi
  echo "== $l"
  timeout -k 10 300 python -u tools/bench_conv.py --modes ${MODES:-fwd} --only $SHAPES 2>&1 | grep -v amdgpu | tee -a $O/conv_$l.log
done; done
fi
if [ -n "$STEP" ]; then
for v in 1 2 3; do for l in prev new; do
  if [ $l = prev ]; then export MSCL_LIB=$R/mscl_amd/csrc/libmscl_hip_ab.so; else unset MSCL_LIB; fi
  timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-variants > $O/bench_${l}_$v.json 2> $O/bench_${l}_$v.err || exit 1
  python - $O/bench_${l}_$v.json $l <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['roofline']['also'][0]['frac'],4), d['final_loss'])
PY
done; done
fi
