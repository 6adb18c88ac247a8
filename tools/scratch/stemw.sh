#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "new: $(python3 tools/bench_conv.py --only stem_rgb_pairw --modes wgrad --iters 20 2>/dev/null | grep stem_)"
echo "old: $(MSCL_WGRAD_STEM=0 python3 tools/bench_conv.py --only stem_rgb_pairw --modes wgrad --iters 20 2>/dev/null | grep stem_)"
echo "new: $(python3 tools/bench_conv.py --r50 --only r50_stem_pairw --modes wgrad --iters 10 2>/dev/null | grep stem_)"
echo "old: $(MSCL_WGRAD_STEM=0 python3 tools/bench_conv.py --r50 --only r50_stem_pairw --modes wgrad --iters 10 2>/dev/null | grep stem_)"
echo "new: $(python3 tools/bench_conv.py --r50 --only r50_stem_177_pairw --modes wgrad --iters 10 2>/dev/null | grep stem_)"
echo "old: $(MSCL_WGRAD_STEM=0 python3 tools/bench_conv.py --r50 --only r50_stem_177_pairw --modes wgrad --iters 10 2>/dev/null | grep stem_)"
