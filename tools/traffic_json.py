"""HBM bytes per launch of the roofline kernel from the two PMC passes of tools/evidence_r04.sh (FETCH_SIZE and WRITE_SIZE in separate
rocprofv3 runs of `tools/bench_conv.py --only l1_64_64 --modes fwd`), corrected as MI355X_MICROARCH.md prescribes for gfx950
(FETCH_SIZE counts 128-B read requests as 64 B: reads x 2; WRITE_SIZE exact; both in KB).
usage: python tools/traffic_json.py gpurun_out/ev_r03/l1_FETCH_SIZE gpurun_out/ev_r03/l1_WRITE_SIZE [commit] > profiles/r03_traffic_layer1.json"""
import csv
import glob
import json
import sys

KERNEL = 'conv_halo64'           # conv_halo64b_kernel (two blocks per CU, the round-4 default) or conv_halo64_kernel


def per_launch(d, counter):
    vals = []
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] == counter and KERNEL in row['Kernel_Name']:
                vals.append(float(row['Counter_Value']))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def wgrad(commit):
    """the layer-1 weight gradient: wgrad_halo64_kernel + its reduce kernel (both counted), algorithmic = x + dy read once + dW"""
    global KERNEL
    KERNEL = 'wgrad_halo64'
    fetch, nf = per_launch(sys.argv[1], 'FETCH_SIZE')
    write, nw = per_launch(sys.argv[2], 'WRITE_SIZE')
    alg = 2 * 8 * 16 * 56 * 56 * 64 * 2 + 27 * 64 * 64 * 4
    # per_launch averages over the dispatches of BOTH kernels (main + reduce, alternating): a conv = one of each = 2 x the mean
    tot = 2 * (2 * fetch + write) * 1024 if fetch is not None and write is not None else None
    print(json.dumps({
        'commit': commit, 'kernel': 'wgrad_halo64_kernel<8> + wgrad_halo64_reduce_kernel, 3x3x3 64->64 on (8,16,56,56,64) bf16',
        'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 tools/bench_conv.py --only l1_64_64 --iters 3 --modes wgrad',
        'FETCH_SIZE_KB_mean_per_dispatch': fetch, 'WRITE_SIZE_KB_mean_per_dispatch': write, 'dispatches_per_pass': [nf, nw],
        'correction': 'reads x2 (gfx950 FETCH_SIZE), WRITE_SIZE exact; one weight gradient = one dispatch of each of the two kernels',
        'traffic_bytes_per_conv': tot, 'algorithmic_bytes_per_conv': alg, 'ratio': (tot / alg) if tot else None}, indent=1))


def main():
    fetch, nf = per_launch(sys.argv[1], 'FETCH_SIZE')
    write, nw = per_launch(sys.argv[2], 'WRITE_SIZE')
    commit = sys.argv[3] if len(sys.argv) > 3 else 'unknown'
    if len(sys.argv) > 4 and sys.argv[4] == 'wgrad_halo64':
        return wgrad(commit)
    alg = 2 * 8 * 16 * 56 * 56 * 64 * 2 + 27 * 64 * 64 * 2          # map in + map out (bf16) + the kernel once
    out = {
        'commit': commit,
        'kernel': 'conv_halo64b_kernel forward (two blocks per CU), 3x3x3 64->64 on (8,16,56,56,64) bf16',
        'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 tools/bench_conv.py --only l1_64_64 '
                   '--iters 3 --modes fwd (two separate passes; tools/evidence_r04.sh)',
        'FETCH_SIZE_KB_per_launch': fetch, 'WRITE_SIZE_KB_per_launch': write, 'dispatches_per_pass': [nf, nw],
        'correction': 'gfx950 FETCH_SIZE counts 128-B read requests as 64 B: reads x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact',
        'traffic_bytes_per_launch': (2 * fetch + write) * 1024 if fetch is not None and write is not None else None,
        'algorithmic_bytes_per_launch': alg,
    }
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
