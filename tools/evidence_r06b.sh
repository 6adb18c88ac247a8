#!/bin/bash
# Round-6 evidence, second call (the first, tools/evidence_r06.sh, fills its 20 minutes): SlowOnly-50 per-kernel table + grouping-cap sweep,
# the in-place strided-shortcut gradient off / on (R3D-18 step and both trunks).  Output: gpurun_out/ev_r06b
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ev_r06b; rm -rf $O; mkdir -p $O
H=${GIT_HEAD:-unknown}; echo "$H" > $O/HEAD
cd $R
GIT_HEAD=$H bash tools/r50_kernels.sh > $O/r50_kernels.log 2>&1
cp gpurun_out/r50k/kernels.md gpurun_out/r50k/group_rows.txt $O/
{ echo "# commit $H: the strided 1x1x1 shortcut's input gradient as a map of its own (False) / added into the entry conv's gradient in place (True)"
  echo "## R3D-18 step, two captured whole-step graphs replayed alternately (tools/ab_step.py, clip-pairs/s)"
  python3 tools/ab_step.py "nn.SHORTCUT_INTO_DX[0]=False" "nn.SHORTCUT_INTO_DX[0]=True" --rounds 5 2>/dev/null | grep -v amdgpu
  echo "## trunks alone (tools/bench_trunk.py, eager, one stream, clips/s): R3D-18 map / in place / map / in place, then SlowOnly-50 the same"
  for a in "" "--r50"; do for f in "--shortcut-map" "" "--shortcut-map" ""; do
    python3 tools/bench_trunk.py $a $f 2>/dev/null | python3 -c "import sys, json; d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-60s into_dx=%s  %.1f clips/s  %.3f ms' % (d['metric'][:60], d['shortcut_into_dx'], d['value'], d['ms_per_iter']))"
  done; done; } > $O/ab_shortcut.txt 2>&1
cat $O/ab_shortcut.txt; echo finished
