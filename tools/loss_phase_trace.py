"""Timeline of the loss phase of one graph-replayed step from a rocprofv3 kernel trace (csv): every kernel between the first
l2norm / linear launch that follows the last forward convolution and the first backward BatchNorm pass, with its stream (queue),
start offset and duration.
usage: python tools/loss_phase_trace.py kernel_trace.csv"""
import csv
import re
import sys


def short(n):
    n = re.sub(r'^void\s+', '', n)
    return re.sub(r'\(.*$', '', n)[:60]


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id', '?')))
    rows.sort()
    # the last full step: between the last two sgd_kernel groups
    sgd = [i for i, r in enumerate(rows) if r[2].startswith('sgd_kernel')]
    end = sgd[-1]
    beg = max(i for i in sgd if rows[end][0] - rows[i][0] > 3_000_000)       # an sgd launch of the previous step (> 3 ms earlier)
    step = rows[beg + 1:end + 1]
    t0 = step[0][0]
    print(f'step span {(step[-1][1] - t0) / 1e3:.1f} us, {len(step)} kernels')
    # loss phase: from the loss_pack launch to the loss_unpack launch (the tracer serialises the streams of the captured step, so the
    # offsets are those of the serialised order; the durations are the kernels' own)
    keys = ('nce_', 'rowdot', 'lmcl', 'loss_pack', 'loss_unpack', 'enqueue', 'step_logs', 'linear_')
    a = max(i for i, r in enumerate(step) if r[2].startswith('loss_pack'))
    b = max(i for i, r in enumerate(step) if r[2].startswith('loss_unpack'))
    lo = step[a][0]
    print(f'loss-phase kernels span {(step[b][1] - lo) / 1e3:.1f} us (from {(lo - t0) / 1e3:.1f} us into the step); '
          f'sum of their durations {sum(e - s for s, e, n, q in step[a:b + 1]) / 1e3:.1f} us over {b + 1 - a} launches')
    for s, e, n, q in step[a:b + 1]:
        mark = '*' if n.startswith(keys) else ' '
        print(f'{mark} q{q:>3} +{(s - lo) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  {n}')


if __name__ == '__main__':
    main()
