#!/bin/bash
# round 4, GPU call 1: parity of the new cases / schedules, then the A/B sweeps that decide them, the edge-shape diagnostic and a
# baseline bench line of this box.  Everything is written under gpurun_out/c1/ (unbuffered: a silent run is killed as hung).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c1; rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tee $O/tests.log | tail -15; rc=${PIPESTATUS[0]}
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out: stopping"; exit 1; fi
echo "== tests rc $rc"
PPS=l2_128_128,l3_256_256,l4_512_512,sepc_128,fpn_133,neck_333_p1,neck_333_p2,neck_133_p1
run() { name=$1; shift; echo "== $name"; timeout -k 10 420 "$@" 2>&1 | grep -v amdgpu | tee $O/$name.log; [ ${PIPESTATUS[0]} -eq 0 ] || exit 1; }
run sweep_pp_late python -u tools/bench_conv.py --sweep MSCL_PP_LATE=0,1,2 --modes fwd,dgrad --only $PPS
MSCL_WGRAD_PP_LATE=1 run sweep_wgrad_pp_late1 python -u tools/bench_conv.py --sweep MSCL_WGRAD_PP=0,2 --modes wgrad --only $PPS
MSCL_WGRAD_PP_LATE=0 run sweep_wgrad_pp_late0 python -u tools/bench_conv.py --sweep MSCL_WGRAD_PP=0,2 --modes wgrad --only $PPS
run sweep_one_split python -u tools/bench_conv.py --sweep MSCL_WGRAD_ONE_SPLIT=-,200,400,1000 --modes wgrad --only l3_,l4_,neck_,fpn_,flow_l3,flow_l4
run sweep_wgrad_halo python -u tools/bench_conv.py --sweep MSCL_WGRAD_HALO_WAVES=4,8 --modes wgrad --only l1_64_64,l1n2
run bench_bn python -u tools/bench_bn.py
run chain_times python -u tools/chain_times.py
run other_shapes python -u tools/other_shapes_diag.py
echo "== bench"
timeout -k 10 300 python -u bench.py --no-cpu-baseline > $O/bench_line.json 2> $O/bench_line.err || { tail -5 $O/bench_line.err; exit 1; }
python - <<'PY'
import json,os
d=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/c1/bench_line.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('step_frac'), d.get('variants'))
PY
