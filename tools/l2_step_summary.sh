#!/bin/bash
# Per-kernel L2 request traffic of the eager single-stream step (TCC_REQ_sum: requests at the XCDs' L2s, 128 B each at most) next to
# each kernel's time: which kernels lean on the L2 -> LDS path hardest while they run (profiles/r06_l2_traffic.md).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/l2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MSCL_STREAMS=1 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum --kernel-trace --output-format csv -d $O/pmc -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-graph --no-variants > $O/pmc.json 2> $O/pmc.err
python3 - $O <<'PY'
import csv, glob, sys, re, collections
O = sys.argv[1]
def short(n):
    n = re.sub(r'^void\s+', '', n); return re.sub(r'\(.*$', '', n)[:56]
f = glob.glob(O + '/pmc/*/*counter_collection.csv')[0]
req = collections.defaultdict(float); hit = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    k = short(r['Kernel_Name'])
    if r['Counter_Name'] == 'TCC_REQ_sum': req[k] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'TCC_HIT_sum': hit[k] += float(r['Counter_Value'])
f = glob.glob(O + '/pmc/*/*kernel_trace.csv')[0]
dur = collections.defaultdict(float); calls = collections.Counter()
for r in csv.DictReader(open(f)):
    k = short(r['Kernel_Name']); dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; calls[k] += 1
steps = 8.0      # 2 warm-up + 4 timed + 2 profiled eager steps (bench.py --no-graph): all eager
rows = sorted(((req[k], k) for k in req), reverse=True)
print('| kernel | launches / step | us / step | L2 requests / step (M) | requests per us | L2 hit |'); print('|---|---|---|---|---|---|')
for v, k in rows[:28]:
    print(f'| `{k}` | {calls[k] / steps:.0f} | {dur[k] / steps:.0f} | {v / steps / 1e6:.1f} | {v / max(dur[k], 1e-9):.0f} | {100 * hit[k] / max(v, 1):.0f} % |')
PY
rm -rf $O/pmc/*/*.db
