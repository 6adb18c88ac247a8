#!/bin/bash
# Where the one-rank RCCL step spends its extra time (profiles/r06_dist_one_rank.md): per-kernel totals of the captured step with every
# collective forced through a one-rank RCCL group against the plain step, from two rocprofv3 kernel traces in one call.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/distprof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/plain -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants > $O/plain.json 2> $O/plain.err
MSCL_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/forced -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants > $O/forced.json 2> $O/forced.err
python3 - $O <<'PY'
import csv, glob, sys, re
O = sys.argv[1]
def load(d):
    f = glob.glob(f'{O}/{d}/*/*kernel_stats.csv')[0]
    out = {}
    for r in csv.DictReader(open(f)):
        n = re.sub(r'\(.*$', '', r['Name'])[:70]
        out[n] = (int(r['Calls']), float(r['TotalDurationNs']) / 1e3)
    return out
a, b = load('plain'), load('forced')
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    rows.append((tb - ta, k, ca, ta, cb, tb))
rows.sort(reverse=True)
print(f'{"kernel":70s} {"plain calls":>11s} {"plain us":>10s} {"forced calls":>12s} {"forced us":>10s} {"delta us":>10s}   (whole runs: 13 captured steps + capture warm-up; same step count in both)')
for d, k, ca, ta, cb, tb in rows[:25]:
    print(f'{k:70s} {ca:11d} {ta:10.0f} {cb:12d} {tb:10.0f} {d:10.0f}')
print('total kernel time: plain %.0f us, forced %.0f us' % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values())))
PY
rm -rf $O/*/*/*.db
