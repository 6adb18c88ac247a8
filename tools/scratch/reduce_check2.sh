#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/red; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for s in l1_64_64 l2_128_128 l3_256_256; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/$s -- python3 $R/tools/bench_conv.py --only $s --modes wgrad --iters 10 > $O/$s.log 2>&1
f=$(ls $O/$s/*/*kernel_stats.csv | head -1); grep "wgrad_halo64" $f | awk -F'",' '{print $1}' | cut -c1-40 | paste - <(grep "wgrad_halo64" $f | awk -F'",' '{print $2}')
done
rm -rf $O/*/
