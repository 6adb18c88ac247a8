#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/pp_prof
rm -f gpurun_out/pp_prof/*
export MSCL_PP=2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/pp_prof -o ppexp -- python3 $R/tools/bench_conv.py --sweep MSCL_PP_EXP=${EXPS:-0,7,8,9,11} --only ${ONLY:-l2_128_128} --modes fwd,dgrad --iters 30 --rounds 2 > $R/gpurun_out/pp_exp.log 2>&1
rc=$?; grep -v "^[WE]2026" $R/gpurun_out/pp_exp.log | tail -6
cd $R; python3 tools/rocpd_stats.py gpurun_out/pp_prof/ppexp_results.db conv_ > gpurun_out/pp_prof/summary.txt; cat gpurun_out/pp_prof/summary.txt
exit $rc
