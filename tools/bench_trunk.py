"""BASELINE.json configs[1]: the R3D-18 trunk alone (Conv3d / BatchNorm3d HIP kernels only), forward + backward on one
(8, 3, 16, 112, 112) batch, loss = mean of the layer-4 map, eager launches on one stream.
Prints one JSON line: clips/s, ms per iteration and the fraction of the dense bf16 MFMA peak over the whole iteration
(algorithmic 241.3 GFLOP per clip: forward 81.39 + input gradients without the stem's + weight gradients).
--r50: BASELINE.json configs[4], the ResNet3dSlowOnly-50 trunk of mscl_r50_cosm_lr3e-2.py on one (8, 3, 32, 224, 224) batch (the
deep / large-activation stress case); its algorithmic FLOPs are counted from the conv descriptors of the run itself.
--group-rows N: nn.GROUP_MAX_ROWS for this run (output positions up to which a weight gradient joins a grouped launch; 0 = none).
usage: python tools/bench_trunk.py [--iters N] [--warmup W] [--r50] [--batch B] [--group-rows N]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GFLOP_FWD_PER_CLIP = 81.39
GFLOP_STEM_FWD_PER_CLIP = 2.832


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--r50', action='store_true')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--group-rows', type=int, default=-1)
    ap.add_argument('--shortcut-map', action='store_true', help='A/B: strided shortcut gradients as maps of their own (nn.SHORTCUT_INTO_DX off)')
    a = ap.parse_args()
    from mscl_amd import Config, build_model
    from mscl_amd.fill import fill_module
    from mscl_amd.nn import Conv3dHip
    from mscl_amd import nn as nn_hip
    if a.group_rows >= 0:
        nn_hip.GROUP_MAX_ROWS = a.group_rows
    nn_hip.SHORTCUT_INTO_DX[0] = not a.shortcut_map
    from mscl_amd.synthetic import synthetic_batch
    dev = torch.device('cuda:0')
    B, T, H = (a.batch, 32, 224) if a.r50 else (a.batch, 16, 112)
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/recognition/moco/' + ('mscl_r50_cosm_lr3e-2.py' if a.r50 else 'mscl_r18_cosm_lr2e-2.py')))
    cfg.model.sup_head.t = T // 4 if a.r50 else T // 2
    model = build_model(cfg.model)
    fill_module(model)
    model.materialize(dev).train()
    x = synthetic_batch(B, T, H, H, 0, 0, device=dev)['imgs'][0]
    trunk = model.recognizer.encoder_q

    def it():
        model.zero_grad()
        maps = trunk(model.aug_gpu.pack_rgb(x))
        maps[-1].float().mean().backward()
    for _ in range(a.warmup):
        it()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        it()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    if a.r50:
        fwd = stem = 0.0
        hbm = 0.0       # bytes the passes of this implementation must move when nothing stays in cache between kernels: per
                        # conv + BatchNorm unit with X input and Y output elements (bf16): forward conv X + Y, BN apply 2Y (+Y
                        # residual); backward BN reduce 2Y, BN apply 3Y, weight gradient X + Y, input gradient Y + X = 3X + 10Y
        for name, m in trunk.named_modules():
            if isinstance(m, Conv3dHip):
                d = next(iter(m._descs.values()))
                f = 2.0 * d.N * d.To * d.Ho * d.Wo * m.out_channels * m.kernel_size[0] * m.kernel_size[1] * m.kernel_size[2] * m.in_channels / 1e9
                fwd += f
                X = d.N * d.T * d.H * d.W * d.C
                Y = d.N * d.To * d.Ho * d.Wo * m.out_channels
                hbm += 2.0 * (3 * X + 10 * Y)
                if m.in_channels == 3:
                    stem += f                       # no input gradient through the stem
        gflop = 3 * fwd - stem
        extra = {'hbm_roofline': {'bound': 'hbm', 'unit': 'GB/s', 'peak': 8000.0, 'pass_bytes_per_iter_gb': hbm / 1e9,
                                  'achieved': hbm / 1e9 / (ms / 1e3), 'frac': hbm / 1e9 / (ms / 1e3) / 8000.0,
                                  'note': 'sum over conv+BatchNorm units of 3X + 10Y bf16 elements (the passes as implemented, every '
                                          'map read from / written to HBM once per pass): the 205-MB maps of layers 1-2 make this '
                                          'configuration HBM-bound, not MFMA-bound'}}
        name = f'clips/sec (ResNet3dSlowOnly-50 trunk fwd+bwd, {T}x{H}^2, bs{B}, 1 GPU)'
    else:
        gflop = B * (3 * GFLOP_FWD_PER_CLIP - GFLOP_STEM_FWD_PER_CLIP)
        name = f'clips/sec (R3D-18 trunk fwd+bwd, 16x112^2, bs{B}, 1 GPU)'
        extra = {}
    tf = gflop / ms
    print(json.dumps({'metric': name, 'value': B / ms * 1e3, 'unit': 'clips/s', 'peak_mem_gb': torch.cuda.max_memory_allocated() / 2 ** 30,
                      'ms_per_iter': ms, 'iters': a.iters, 'group_max_rows': nn_hip.GROUP_MAX_ROWS, 'shortcut_into_dx': nn_hip.SHORTCUT_INTO_DX[0], 'dtype': 'bf16', 'launch': 'eager, one stream',
                      'roofline': {'bound': 'mfma', 'achieved': tf, 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': tf / 2500.0,
                                   'algorithmic_gflop_per_iter': gflop,
                                   'note': 'whole iteration incl. BatchNorm passes and launch gaps, not one kernel'}, **extra}))


if __name__ == '__main__':
    main()
