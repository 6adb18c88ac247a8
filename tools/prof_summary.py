"""Summarise a rocprofv3 kernel trace (rocpd .db or *_kernel_trace.csv) per (kernel, grid) so that template
instantiations shared by several layers can be told apart.
usage: python tools/prof_summary.py RESULTS.db|kernel_trace.csv [--steps N] [--top K]"""
import argparse
import collections
import csv
import re
import sqlite3


def short(name):
    name = re.sub(r'^void\s+', '', name)
    name = re.sub(r'\(.*$', '', name)
    return name[:110]


def rows_from_db(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    q = 'select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels' if 'grid_x' in cols else None
    if q is None:
        raise SystemExit(f'unexpected schema: {cols}')
    for name, s, e, gx, gy, gz, wx in db.execute(q):
        yield name, (e - s) / 1e3, (gx // max(1, wx)) * gy * gz


def rows_from_csv(path):
    for r in csv.DictReader(open(path)):
        g = int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
        yield r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('path')
    ap.add_argument('--steps', type=int, default=1, help='divide totals by this many steps')
    ap.add_argument('--top', type=int, default=40)
    ap.add_argument('--by-grid', action='store_true')
    a = ap.parse_args()
    rows = rows_from_db(a.path) if a.path.endswith('.db') else rows_from_csv(a.path)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, us, blocks in rows:
        k = (short(name), blocks) if a.by_grid else (short(name),)
        agg[k][0] += 1
        agg[k][1] += us
    tot = sum(v[1] for v in agg.values())
    print(f'total kernel time {tot / a.steps / 1e3:.3f} ms/step over {a.steps} steps; {sum(v[0] for v in agg.values()) / a.steps:.0f} launches/step')
    print(f'{"kernel":112s} {"blocks":>8s} {"calls/step":>10s} {"avg us":>9s} {"ms/step":>8s} {"%":>6s}')
    for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print(f'{k[0]:112s} {str(k[1]) if a.by_grid else "":>8s} {n / a.steps:10.1f} {us / n:9.1f} {us / a.steps / 1e3:8.3f} {100 * us / tot:6.1f}')


if __name__ == '__main__':
    main()
