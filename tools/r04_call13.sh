#!/bin/bash
# round 4, GPU call 13: pool split A/B in the step; SlowOnly-50 (config 5) stage table and per-kernel time of its trunk
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c13; rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
for v in 1 2 3; do for s in 0 1; do
  MSCL_POOL_SPLIT=$s timeout -k 10 300 python -u bench.py --no-cpu-baseline --no-variants > $O/bench_${s}_$v.json 2> $O/bench_${s}_$v.err || exit 1
  python - $O/bench_${s}_$v.json $s <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('pool split', sys.argv[2], round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['roofline']['also'][0]['frac'],4))
PY
done; done
timeout -k 10 420 python -u tools/bench_conv.py --r50 --iters 10 2>&1 | grep -v amdgpu | tee $O/conv_r50.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r50 -- python3 $R/tools/bench_trunk.py --r50 --iters 10 > $O/trunk_r50.json 2> $O/trunk_r50.err
cd $R; T=$(find $O/prof_r50 -name '*kernel_trace.csv' | head -1); python3 tools/prof_summary.py $T --by-grid --top 60 > $O/prof_r50_summary.txt 2>&1; find $O/prof_r50 -name '*kernel_trace.csv' -delete; find $O/prof_r50 -name '*kernel_stats.csv' -exec cp {} $O/r50_kernel_stats.csv \;
cat $O/trunk_r50.json | cut -c1-300
