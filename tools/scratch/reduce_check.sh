#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/red; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -k "wgrad or real_layer" -x 2>&1 | tail -3 || exit 1
python3 tools/bench_conv.py --only l1_64_64 --modes wgrad --iters 20 2>/dev/null | grep l1_
python3 tools/bench_conv.py --only l2_128_128 --modes wgrad --iters 20 2>/dev/null | grep l2_
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $O/line.json 2> $O/err.txt
f=$(ls $O/st/*/*kernel_stats.csv | head -1); grep "wgrad_halo64\|Name" $f | cut -d, -f1-8 | cut -c1-200
cd $R; for i in 1 2; do python3 bench.py --no-cpu-baseline --no-variants 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'])"; done
rm -rf $O/st
