#!/bin/bash
# What ONE GPU can show about the world-size > 1 path (profiles/r06_dist_one_rank.md): the bench line with every collective forced
# through a one-rank RCCL group (MSCL_FORCE_DIST=1), eager + sub-graphs and whole-step capture, beside the plain line, in one call.
cd "$(dirname "$0")/.."
O=gpurun_out/dist; mkdir -p $O
line() { python3 -c "
import json,sys
d=json.load(open('$1'))
r=d.get('rccl') or {}
print('$2', round(d['value'],1), 'clip-pairs/s', round(d['ms_per_step'],3), 'ms/step', d.get('config',{}).get('launch','')[:34], {k:r[k] for k in ('grad_buckets_MB','exposed_wire_ms_model','exposed_wire_ms_measured','grad_collective','grad_transport') if k in r})"; }
run() { name=$1; shift; env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $O/$name.json 2> $O/$name.err; line $O/$name.json $name; }
run plain_1 A=1
run forced_eager_1 MSCL_FORCE_DIST=1
run forced_graph_1 MSCL_FORCE_DIST=1 MSCL_GRAPH_DP=1
run plain_2 A=1
run forced_eager_2 MSCL_FORCE_DIST=1
run forced_graph_2 MSCL_FORCE_DIST=1 MSCL_GRAPH_DP=1
run forced_graph_ch4 MSCL_FORCE_DIST=1 MSCL_GRAPH_DP=1 NCCL_MAX_NCHANNELS=4
run forced_graph_ch16 MSCL_FORCE_DIST=1 MSCL_GRAPH_DP=1 NCCL_MIN_NCHANNELS=16
run forced_graph_bf16 MSCL_FORCE_DIST=1 MSCL_GRAPH_DP=1 MSCL_GRAD_TRANSPORT=bf16
