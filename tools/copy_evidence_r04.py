"""Copy gpurun_out/ev_r04/* (tools/evidence_r04.sh) to profiles/r04_* and rewrite the Results paragraph of DESIGN.md section 6 from
them.  usage (dev container, after the gpurun call): python tools/copy_evidence_r04.py"""
import json
import os
import re
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = os.path.join(R, 'gpurun_out', 'ev_r04')
P = os.path.join(R, 'profiles')
MAP = {'bench_line.json': 'r04_bench_line.json', 'bench_kernel_stats.csv': 'r04_bench_kernel_stats.csv',
       'bench_line_under_rocprof.json': 'r04_bench_line_under_rocprof.json', 'bench_line_deterministic.json': 'r04_bench_line_deterministic.json',
       'bn_passes.txt': 'r04_bn_passes.md', 'chain_times.txt': 'r04_chain_times.txt', 'conv_stage.log': 'r04_conv_stage_roofline.md',
       'conv_stage_r50.log': 'r04_conv_stage_r50.md', 'glue_launches.txt': 'r04_glue_launches.txt', 'pp_stamps.txt': 'r04_pp_stamps_run.txt',
       'step_config5_r50_32x224.json': 'r04_step_config5_r50_32x224.json', 'step_config5_r50_8x224.json': 'r04_step_config5_r50_8x224.json',
       'step_utilisation.md': 'r04_step_utilisation.md', 'traffic_layer1.json': 'r04_traffic_layer1.json',
       'traffic_layer1_wgrad.json': 'r04_traffic_layer1_wgrad.json', 'trunk_r18.json': 'r04_trunk_config2.json', 'trunk_r50.json': 'r04_trunk_config5_r50.json'}


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def stage(lines, name):
    for ln in lines:
        if ln.startswith(name + ' '):
            tf = re.findall(r'([0-9.]+) TF', ln)
            return ' / '.join(f'{float(t):.0f}' for t in tf)
    return '?'


def main():
    for a, b in MAP.items():
        shutil.copy(os.path.join(E, a), os.path.join(P, b))
    head = open(os.path.join(E, 'HEAD')).read().strip()
    d = last_json(os.path.join(P, 'r04_bench_line.json'))
    rf = d['roofline']
    util = open(os.path.join(P, 'r04_step_utilisation.md')).read()
    m = re.search(r'kernel time ([0-9.]+) ms/step.*?MFMA busy over all kernel cycles ([0-9.]+) %; HBM-side traffic ([0-9.]+) GB', util)
    st = open(os.path.join(P, 'r04_conv_stage_roofline.md')).read().splitlines()
    chain = open(os.path.join(P, 'r04_chain_times.txt')).read()
    g3 = re.search(r'whole step, graph, 3 stream\(s\)\s+([0-9.]+) ms', chain).group(1)
    g1 = re.search(r'whole step, graph, 1 stream\(s\)\s+([0-9.]+) ms', chain).group(1)
    tr = json.load(open(os.path.join(P, 'r04_traffic_layer1.json')))
    t18 = last_json(os.path.join(P, 'r04_trunk_config2.json'))['value']
    t50 = last_json(os.path.join(P, 'r04_trunk_config5_r50.json'))['value']
    s50 = last_json(os.path.join(P, 'r04_step_config5_r50_32x224.json'))['value']
    s50s = last_json(os.path.join(P, 'r04_step_config5_r50_8x224.json'))['value']
    det = d['variants']['deterministic']['value']
    txt = (f"**Results** (one MI355X, `profiles/r04_*`, every file stamped with its commit, `{head}`; box-to-box spread is several per cent, A/B pairs\n"
           f"are made inside one call). Headline: **{d['value']:.1f} clip-pairs/s** ({d['ms_per_step']:.2f} ms per step of 8 clip-pairs; round-3 driver line 1075.5),\n"
           f"deterministic mode {det:.0f}, CPU baseline (oracle, {d['cpu_baseline']['cores']} threads) {d['cpu_baseline']['value']:.2f}. Dominant kernel: layer-1 forward "
           f"{rf['avg_launch_ms'] * 1e3:.1f} µs by events =\n{rf['achieved']:.0f} TFLOP/s = **{rf['frac']:.3f} of the MFMA peak** (round 3: 0.345), HBM traffic "
           f"{tr['traffic_bytes_per_launch'] / 1e6:.1f} MB per launch = {tr['traffic_bytes_per_launch'] / tr['algorithmic_bytes_per_launch']:.2f} × algorithmic; 128→128\n"
           f"forward of the ping-pong kernel {rf['also'][0]['frac']:.3f}. Whole step: {m.group(1)} ms of kernel time on one stream under the counters (round 3: 10.76),\n"
           f"{m.group(2)} % MFMA busy, {m.group(3)} GB of HBM traffic; {g1} ms replayed on one stream, {g3} ms on three. Stages alone (TFLOP/s fwd / dgrad /\n"
           f"wgrad): layer 1 {stage(st, 'l1_64_64')}, 128→128 {stage(st, 'l2_128_128')}, 256→256 {stage(st, 'l3_256_256')}, 512→512 {stage(st, 'l4_512_512')}, paired stem\n"
           f"{stage(st, 'stem_rgb_pairw')} (fwd / wgrad). R3D-18 trunk {t18:.0f} clips/s; SlowOnly-50 trunk {t50:.0f} clips/s, mscl_r50 step {s50:.0f} / {s50s:.0f} clip-pairs/s at\n"
           f"32 × 224² / 8 × 224². What moved the step this round, each measured as an alternating A/B: sliced window-resident weight\n"
           f"gradient +2 %, two layer-1 blocks per CU +1.7 %, 16-byte epilogue stores +0.9 %, reduce-scatter statistics +0.6 %, BatchNorm\n"
           f"constant loads issued together +1.0 %, stem kernel / two-tap ring stages / split-K finalize +0.2…0.5 % each.\n")
    dp = os.path.join(R, 'DESIGN.md')
    s = open(dp).read()
    a = s.index('**Results**')
    b = s.index('## 7. Status against')
    s = s[:a] + txt + '\n' + s[b:]
    open(dp, 'w').write(s)
    print(txt)


if __name__ == '__main__':
    main()
