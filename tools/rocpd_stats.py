"""Per-kernel duration summary of a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace` on ROCm 7.2).
usage: python tools/rocpd_stats.py results.db [name-filter]"""
import sqlite3
import statistics
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    filt = sys.argv[2] if len(sys.argv) > 2 else ''
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith('rocpd_kernel_dispatch'))
    ks = next(t for t in tabs if t.startswith('rocpd_info_kernel_symbol'))
    cols = [r[1] for r in c.execute(f'pragma table_info({ks})')]
    name_col = 'display_name' if 'display_name' in cols else 'kernel_name'
    rows = c.execute(f'select s.{name_col}, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id').fetchall()
    by = {}
    for n, dur in rows:
        if filt in n:
            by.setdefault(n, []).append(dur / 1e3)
    print(f'{"kernel":90s} {"n":>6s} {"avg us":>9s} {"med us":>9s} {"min us":>9s} {"total ms":>9s}')
    for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print(f'{n[:90]:90s} {len(v):6d} {sum(v)/len(v):9.2f} {statistics.median(v):9.2f} {min(v):9.2f} {sum(v)/1e3:9.3f}')


if __name__ == '__main__':
    main()
