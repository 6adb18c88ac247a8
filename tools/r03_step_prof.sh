#!/bin/bash
# per-kernel time of the step, single stream, eager (kernels attributed one by one): rocprofv3 kernel trace -> tools/rocpd_stats.py
# usage: [ARGS=--deterministic] [TAG=det] bash tools/r03_step_prof.sh
set -o pipefail
R=$GRAFT_REPO_ROOT; TAG=${TAG:-step}; O=$R/gpurun_out/${TAG}_prof; mkdir -p $O; rm -f $O/*
cd /tmp && export TMPDIR=/tmp
MSCL_STREAMS=1 rocprofv3 --kernel-trace -d $O -o step -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-graph $ARGS > $O/bench.json 2> $O/bench.err
rc=$?
cd $R; python3 tools/rocpd_stats.py $O/step_results.db > $O/summary.txt; head -45 $O/summary.txt; exit $rc
