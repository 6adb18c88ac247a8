#!/bin/bash
cd $GRAFT_REPO_ROOT
for n in 0 2 4 6 8 12 16 0; do
  echo "dephase $n: $(MSCL_HALO_DEPHASE=$n python3 tools/bench_conv.py --only l1_64_64 --modes fwd,dgrad --iters 20 2>/dev/null | grep l1_64_64)"
done
