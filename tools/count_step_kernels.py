"""Launches per kernel name in ONE graph-replayed step of a rocprofv3 --kernel-trace --output-format csv run of bench.py (the last full
step between two sgd_kernel groups), aggregated by family.  usage: python tools/count_step_kernels.py DIR"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted((int(r['Start_Timestamp']), re.sub(r'\(.*$', '', re.sub(r'^void\s+', '', r['Kernel_Name']))[:48]) for r in csv.DictReader(open(f)))
sgd = [i for i, r in enumerate(rows) if r[1].startswith('sgd_kernel')]
end = sgd[-1]
beg = max(i for i in sgd if rows[end][0] - rows[i][0] > 3_000_000)
step = rows[beg + 1:end + 1]
c = collections.Counter(n for _, n in step)
print(f'{len(step)} launches in the step')
for n, k in c.most_common():
    if k >= 3 or 'copy' in n.lower() or 'fill' in n.lower() or 'at::' in n:
        print(f'{k:5d}  {n}')
