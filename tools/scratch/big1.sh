#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== default"; python3 tools/bench_conv.py --r50 --iters 10 --modes wgrad 2>/dev/null | grep r50_
echo "== MSCL_WGRAD_BIG1=1 (1-tap)"; MSCL_WGRAD_BIG1=1 python3 tools/bench_conv.py --r50 --iters 10 --modes wgrad 2>/dev/null | grep r50_
echo "== MSCL_WGRAD_BIG1=3 (<= 3 taps)"; MSCL_WGRAD_BIG1=3 python3 tools/bench_conv.py --r50 --iters 10 --modes wgrad 2>/dev/null | grep r50_
