#!/bin/bash
# round 4, GPU call 1: parity of the new cases / schedules, then the A/B sweeps that decide them, the edge-shape diagnostic and a
# baseline bench line of this box.  Everything is written under gpurun_out/c1/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c1; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?
tail -5 $O/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out: stopping"; exit 1; fi
echo "== tests rc $rc"
PPS=l2_128_128,l3_256_256,l4_512_512,sepc_128,fpn_133,neck_333_p1,neck_333_p2,neck_133_p1
timeout -k 10 300 python tools/bench_conv.py --sweep MSCL_PP_LATE=0,1,2 --modes fwd,dgrad --only $PPS > $O/sweep_pp_late.log 2>&1 || exit 1
cat $O/sweep_pp_late.log | grep -v amdgpu
MSCL_WGRAD_PP_LATE=1 timeout -k 10 300 python tools/bench_conv.py --sweep MSCL_WGRAD_PP=0,2 --modes wgrad --only $PPS > $O/sweep_wgrad_pp_late1.log 2>&1 || exit 1
cat $O/sweep_wgrad_pp_late1.log | grep -v amdgpu
MSCL_WGRAD_PP_LATE=0 timeout -k 10 300 python tools/bench_conv.py --sweep MSCL_WGRAD_PP=0,2 --modes wgrad --only $PPS > $O/sweep_wgrad_pp_late0.log 2>&1 || exit 1
cat $O/sweep_wgrad_pp_late0.log | grep -v amdgpu
timeout -k 10 300 python tools/bench_conv.py --sweep MSCL_WGRAD_ONE_SPLIT=-,200,400,1000 --modes wgrad --only l3_,l4_,neck_,fpn_,flow_l3,flow_l4 > $O/sweep_one_split.log 2>&1 || exit 1
cat $O/sweep_one_split.log | grep -v amdgpu
timeout -k 10 600 python tools/other_shapes_diag.py > $O/other_shapes.log 2>&1 || exit 1
grep -v amdgpu $O/other_shapes.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_line.json 2> $O/bench_line.err || exit 1
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/c1/bench_line.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('step_frac'))
PY
