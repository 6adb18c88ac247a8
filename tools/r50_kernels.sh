#!/bin/bash
# SlowOnly-50 trunk (BASELINE configs[4], 8 x 32 x 224^2): per-kernel table of one forward + backward, and the sweep of the row cap
# under which weight gradients join a grouped launch (nn.GROUP_MAX_ROWS).  Output: gpurun_out/r50k/{kernels.md,group_rows.txt}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r50k; rm -rf $O; mkdir -p $O
H=${GIT_HEAD:-unknown}
cd $R
{ echo "# commit $H: tools/bench_trunk.py --r50 --iters 10 --group-rows N (eager, one stream), clips/s; layer-4 maps have 6272 output positions, layer 3: 25088, layer 2: 100352, layer 1: 401408"
  for n in 16384 0 32768 131072 1073741824 16384; do
    python3 tools/bench_trunk.py --r50 --iters 10 --group-rows $n 2>/dev/null | python3 -c "import sys, json; d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('group_max_rows %10d  %.1f clips/s  %.3f ms  peak %.1f GB' % (d['group_max_rows'], d['value'], d['ms_per_iter'], d['peak_mem_gb']))"
  done; } > $O/group_rows.txt 2>&1
cat $O/group_rows.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 $R/tools/bench_trunk.py --r50 --iters 10 > $O/st.log 2>&1
f=$(ls $O/st/*/*kernel_stats.csv | head -1)
python3 - "$f" "$O/st.log" "$H" > $O/kernels.md <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
line = json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1])
iters = 15.0                                          # 5 warm-up + 10 timed iterations traced
lib = [r for r in rows if not r['Name'].startswith(('void at::', 'at::', '__amd_rocclr'))]
tot = sum(float(r['TotalDurationNs']) for r in lib) / iters / 1e6
print(f"# commit {sys.argv[3]}: ResNet3dSlowOnly-50 trunk at 8 x 32 x 224^2 (BASELINE configs[4]), per-kernel time of one forward + backward")
print(f"`rocprofv3 --kernel-trace --stats -- python3 tools/bench_trunk.py --r50 --iters 10` (15 iterations traced; eager, one stream; {line['ms_per_iter']:.1f} ms per")
print(f"iteration = {line['value']:.0f} clips/s under the tracer).  Library kernels only ({tot:.2f} ms per iteration; the harness's aten fills / copies are left out).\n")
print("| kernel | launches / iteration | ms / iteration | avg us |\n|---|---|---|---|")
for r in sorted(lib, key=lambda r: -float(r['TotalDurationNs'])):
    name = r['Name'].split('(')[0].replace('void ', '')
    print(f"| `{name}` | {int(r['Calls']) / iters:.1f} | {float(r['TotalDurationNs']) / iters / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} |")
bn = sum(float(r['TotalDurationNs']) for r in lib if r['Name'].startswith('bn_')) / iters / 1e6
print(f"\nBatchNorm passes: {bn:.2f} ms per iteration ({100 * bn / tot:.0f} % of the library kernel time).")
PY
cat $O/kernels.md
rm -rf $O/st/*/*.db
