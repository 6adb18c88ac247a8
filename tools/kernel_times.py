"""Per-launch durations of every kernel in a rocprofv3 --kernel-trace --output-format csv directory, in launch order (every third launch).
usage: python tools/kernel_times.py DIR"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"][:44]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    print(k, len(v), [round(x, 1) for x in v[:70:3]])
