"""Idle time of the device inside one graph-replayed step: union of all kernel intervals between two consecutive lmcl_kernel launches,
the gaps in it (no kernel running on any queue) and the kernels either side of the large ones.
usage: python tools/step_idle.py results.db [min_gap_us]"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith('rocpd_kernel_dispatch'))
    ks = next(t for t in tabs if t.startswith('rocpd_info_kernel_symbol'))
    cols = [r[1] for r in c.execute(f'pragma table_info({ks})')]
    name_col = 'display_name' if 'display_name' in cols else 'kernel_name'
    rows = c.execute(f'select s.{name_col}, d.start, d.end, d.queue_id from {kd} d join {ks} s on d.kernel_id = s.id order by d.start').fetchall()
    idx = [i for i, r in enumerate(rows) if r[0].startswith('lmcl_kernel')]
    a, b = idx[-3], idx[-2]
    step = rows[a:b]
    t0, t1 = step[0][1], rows[b][1]
    print(f'step {(t1 - t0) / 1e3:.1f} us, {len(step)} kernels, queues {sorted(set(r[3] for r in step))}')
    busy_q = {}
    for n, s, e, q in step:
        busy_q[q] = busy_q.get(q, 0) + (e - s)
    print('busy per queue (us):', {q: round(v / 1e3, 1) for q, v in busy_q.items()})
    cur_end, idle, gaps = step[0][1], 0, []
    last = step[0]
    for r in step:
        n, s, e, q = r
        if s > cur_end:
            idle += s - cur_end
            if (s - cur_end) / 1e3 >= min_gap:
                gaps.append(((cur_end - t0) / 1e3, (s - cur_end) / 1e3, last[0][:50], n[:50]))
        if e > cur_end:
            cur_end, last = e, r
    print(f'device idle (no kernel on any queue): {idle / 1e3:.1f} us = {100.0 * idle / (t1 - t0):.1f} % of the step')
    print(f'gaps >= {min_gap} us: {len(gaps)}, total {sum(g[1] for g in gaps):.1f} us')
    for at, g, before, after in gaps:
        print(f'  at {at:8.1f}  gap {g:6.1f}   {before}  ->  {after}')


if __name__ == '__main__':
    main()
