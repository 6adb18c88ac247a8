#!/bin/bash
cd $GRAFT_REPO_ROOT
for s in fpn_133 neck_133_p1; do
  echo "new : $(python3 tools/bench_conv.py --only $s --iters 20 --modes wgrad 2>/dev/null | grep $s)"
  echo "old : $(MSCL_WGRAD_HALO_K1=0 python3 tools/bench_conv.py --only $s --iters 20 --modes wgrad 2>/dev/null | grep $s)"
done
for i in 1 2 3; do
  echo "new  $(python3 bench.py --no-cpu-baseline --no-variants 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1))")"
  echo "old  $(MSCL_WGRAD_HALO_K1=0 python3 bench.py --no-cpu-baseline --no-variants 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'],1))")"
done
