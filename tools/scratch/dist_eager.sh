#!/bin/bash
# the launch mode world size > 1 runs by default (eager launches + collective-free branches from sub-graphs), plain and with every
# collective forced through a one-rank RCCL group, beside the whole-step graph, in one call
cd $GRAFT_REPO_ROOT
O=gpurun_out/dist2; mkdir -p $O
run() { name=$1; shift; flags=$1; shift; env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants $flags > $O/$name.json 2> $O/$name.err
  python3 -c "
import json
d=json.loads([l for l in open('$O/$name.json') if l.startswith('{')][-1])
print('$name', round(d['value'],1), 'clip-pairs/s', round(d['ms_per_step'],3), 'ms/step |', d['config']['launch'][:90])"; }
run plain_graph "" A=1
run plain_eager "--no-graph" A=1
run forced_graph "" MSCL_FORCE_DIST=1
run forced_eager "--no-graph" MSCL_FORCE_DIST=1
run plain_graph_2 "" A=1
run plain_eager_2 "--no-graph" A=1
run forced_graph_2 "" MSCL_FORCE_DIST=1
run forced_eager_2 "--no-graph" MSCL_FORCE_DIST=1
