"""Per-kernel MFMA utilisation and HBM traffic of the eager single-stream step from three rocprofv3 PMC passes
(SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE; FETCH_SIZE; WRITE_SIZE) -> markdown table.
usage: python tools/pmc_step_summary.py DIR_MFMA DIR_FETCH DIR_WRITE N_STEPS_TOTAL"""
import collections
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r'^void\s+', '', n)
    return re.sub(r'\(.*$', '', n)[:64]


def counters(d):
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    seen = set()
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        out[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (r['Dispatch_Id'], k)
        if key not in seen:
            seen.add(key); calls[k] += 1
    return out, calls


def durations(d):
    f = glob.glob(d + '/*/*kernel_trace.csv')[0]
    dur = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        dur[short(r['Kernel_Name'])] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    return dur


def main():
    dm, df, dw, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    cm, calls = counters(dm)
    cf, _ = counters(df)
    cw, _ = counters(dw)
    dur = durations(df)          # the FETCH pass perturbs least; durations of the MFMA pass are used for its own ratio
    dur_m = durations(dm)
    rows = []
    for k in cm:
        if 'at::native' in k and 'direct_copy' in k:
            continue
        us = dur.get(k, 0.0) / steps
        gui = cm[k].get('GRBM_GUI_ACTIVE', 0.0) / 8            # summed over 8 XCDs
        busy = cm[k].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / 1024   # per SIMD
        util = busy / gui if gui else 0.0
        rd = 2 * cf[k].get('FETCH_SIZE', 0.0) * 1024 if k in cf else 0.0
        wr = cw[k].get('WRITE_SIZE', 0.0) * 1024 if k in cw else 0.0
        gbs = (rd + wr) / (dur.get(k, 1.0) * 1e-6) / 1e9 if dur.get(k) else 0.0
        rows.append((us, k, calls[k] / steps, util, (rd + wr) / steps / 1e6, gbs))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    tot_busy = sum(cm[k].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for k in cm) / 1024
    tot_gui = sum(cm[k].get('GRBM_GUI_ACTIVE', 0.0) for k in cm if not ('at::native' in k and 'direct_copy' in k)) / 8
    tot_bytes = sum(r[4] for r in rows)
    print(f'kernel time {tot / 1e3:.2f} ms/step (single stream, eager); MFMA busy over all kernel cycles {100 * tot_busy / tot_gui:.1f} %; '
          f'HBM-side traffic {tot_bytes / 1e3:.2f} GB/step = {tot_bytes / 1e3 / (tot / 1e6) / 1e3:.2f} TB/s averaged over kernel time\n')
    print('| kernel | launches/step | us/step | MFMA busy | HBM MB/step | GB/s while running |')
    print('|---|---|---|---|---|---|')
    for us, k, n, util, mb, gbs in rows[:26]:
        print(f'| `{k}` | {n:.0f} | {us:.0f} | {100 * util:.0f} % | {mb:.0f} | {gbs:.0f} |')


if __name__ == '__main__':
    main()
