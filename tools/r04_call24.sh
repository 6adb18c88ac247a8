#!/bin/bash
# round 4, GPU call 24: full -m gpu suite + smoke on the current code
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c24; rm -rf $O; mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 400 2>&1 | tee $O/tests.log | tail -4; rc=${PIPESTATUS[0]}
echo "== tests rc $rc"; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
