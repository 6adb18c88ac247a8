#!/bin/bash
# round 4, GPU call 22: kernel timeline of the loss phase inside the graph-replayed step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c22; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-variants > $O/bench.json 2> $O/bench.err
cd $R
T=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python3 tools/loss_phase_trace.py $T > $O/loss_phase.txt 2>&1
head -5 $T | cut -c1-400 > $O/trace_head.txt
find $O/trace -name '*kernel_trace.csv' -delete
tail -5 $O/loss_phase.txt
