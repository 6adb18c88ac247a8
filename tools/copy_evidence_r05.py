"""Copy gpurun_out/ev_r05/* (tools/evidence_r05.sh) to profiles/r05_* and rewrite the Results paragraph of DESIGN.md section 6 from
them.  usage (dev container, after the gpurun call): python tools/copy_evidence_r05.py"""
import json
import os
import re
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = os.path.join(R, 'gpurun_out', 'ev_r05')
P = os.path.join(R, 'profiles')
MAP = {'bench_line.json': 'r05_bench_line.json', 'bench_kernel_stats.csv': 'r05_bench_kernel_stats.csv',
       'bench_line_under_rocprof.json': 'r05_bench_line_under_rocprof.json', 'bench_line_deterministic.json': 'r05_bench_line_deterministic.json',
       'bn_passes.txt': 'r05_bn_passes.md', 'chain_times.txt': 'r05_chain_times.txt', 'conv_stage.log': 'r05_conv_stage_roofline.md',
       'conv_stage_r50.log': 'r05_conv_stage_r50.md', 'glue_launches.txt': 'r05_glue_launches.txt',
       'step_config5_r50_32x224.json': 'r05_step_config5_r50_32x224.json', 'step_config5_r50_8x224.json': 'r05_step_config5_r50_8x224.json',
       'step_utilisation.md': 'r05_step_utilisation.md', 'traffic_layer1.json': 'r05_traffic_layer1.json',
       'trunk_r18.json': 'r05_trunk_config2.json', 'trunk_r50.json': 'r05_trunk_config5_r50.json',
       'nce_passes.md': 'r05_nce_passes.md', 'loss_phase.txt': 'r05_loss_phase.txt', 'ab_group_wgrad.txt': 'r05_ab_group_wgrad.txt',
       'training_curve.md': 'r05_training_curve.md'}


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
        if os.path.exists(os.path.join(E, a)):
            shutil.copy(os.path.join(E, a), os.path.join(P, b))
    head = open(os.path.join(E, 'HEAD')).read().strip()
    d = last_json(os.path.join(P, 'r05_bench_line.json'))
    rf = d['roofline']
    util = open(os.path.join(P, 'r05_step_utilisation.md')).read()
    m = re.search(r'kernel time ([0-9.]+) ms/step.*?MFMA busy over all kernel cycles ([0-9.]+) %; HBM-side traffic ([0-9.]+) GB', util)
    st = open(os.path.join(P, 'r05_conv_stage_roofline.md')).read().splitlines()
    chain = open(os.path.join(P, 'r05_chain_times.txt')).read()
    g3 = re.search(r'whole step, graph, 3 stream\(s\)\s+([0-9.]+) ms', chain).group(1)
    g1 = re.search(r'whole step, graph, 1 stream\(s\)\s+([0-9.]+) ms', chain).group(1)
    tr = json.load(open(os.path.join(P, 'r05_traffic_layer1.json')))
    t18 = last_json(os.path.join(P, 'r05_trunk_config2.json'))['value']
    t50 = last_json(os.path.join(P, 'r05_trunk_config5_r50.json'))['value']
    s50 = last_json(os.path.join(P, 'r05_step_config5_r50_32x224.json'))['value']
    s50s = last_json(os.path.join(P, 'r05_step_config5_r50_8x224.json'))['value']
    det = d['variants']['deterministic']['value']
    ab = open(os.path.join(P, 'r05_ab_group_wgrad.txt')).read()
    med = re.findall(r'median\s+([0-9.]+)', ab)
    nce = open(os.path.join(P, 'r05_nce_passes.md')).read()
    nf = re.search(r'nce_fwd_kernel<24>.*?([0-9.]+) us', nce); nb = re.search(r'nce_bwd_kernel<24>.*?([0-9.]+) us', nce)
    txt = (f"**Results** (one MI355X, `profiles/r05_*`, every file stamped with its commit, `{head}`; box-to-box spread is several per cent -- the\n"
           f"boxes of this round's calls read 1099-1188 clip-pairs/s on one and the same build -- so A/B pairs are made inside one call; the number\n"
           f"to quote is the DRIVER's: round 4 1095.8). This evidence run: **{d['value']:.1f} clip-pairs/s** ({d['ms_per_step']:.2f} ms per step of 8\n"
           f"clip-pairs), deterministic mode {det:.0f}, CPU baseline (oracle, {d['cpu_baseline']['cores']} threads) {d['cpu_baseline']['value']:.2f}. Dominant kernel: layer-1 forward "
           f"{rf['avg_launch_ms'] * 1e3:.1f} µs by events =\n{rf['achieved']:.0f} TFLOP/s = **{rf['frac']:.3f} of the MFMA peak**, HBM traffic "
           f"{tr['traffic_bytes_per_launch'] / 1e6:.1f} MB per launch = {tr['traffic_bytes_per_launch'] / tr['algorithmic_bytes_per_launch']:.2f} × algorithmic; 128→128\n"
           f"forward of the ping-pong kernel {rf['also'][0]['frac']:.3f}. Whole step: {m.group(1)} ms of kernel time on one stream under the counters (round 4: 9.90),\n"
           f"{m.group(2)} % MFMA busy, {m.group(3)} GB of HBM traffic; {g1} ms replayed on one stream, {g3} ms on three. Stages alone (TFLOP/s fwd / dgrad /\n"
           f"wgrad): layer 1 {stage(st, 'l1_64_64')}, 128→128 {stage(st, 'l2_128_128')}, 256→256 {stage(st, 'l3_256_256')}, 512→512 {stage(st, 'l4_512_512')}, paired stem\n"
           f"{stage(st, 'stem_rgb_pairw')} (fwd / wgrad). R3D-18 trunk {t18:.0f} clips/s; SlowOnly-50 trunk {t50:.0f} clips/s, mscl_r50 step {s50:.0f} / {s50s:.0f} clip-pairs/s at\n"
           f"32 × 224² / 8 × 224². What moved the step this round, each an alternating A/B in one process (`profiles/r05_ab_sweeps.md`): the grouped\n"
           f"weight-gradient launch of the small maps {med[0] if med else '?'} → {med[1] if len(med) > 1 else '?'} clip-pairs/s; InfoNCE passes (24 rows) "
           f"28.6 / 38.2 → {nf.group(1) if nf else '?'} / {nb.group(1) if nb else '?'} µs per launch.\n")
    dp = os.path.join(R, 'DESIGN.md')
    s = open(dp).read()
    a = s.index('**Results**')
    b = s.index('## 7. Status against')
    s = s[:a] + txt + '\n' + s[b:]
    open(dp, 'w').write(s)
    print(txt)


if __name__ == '__main__':
    main()
