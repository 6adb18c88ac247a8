"""Micro-benchmark of the conv kernels on the shapes of one mscl_r18 step (B=8, T=16, 112^2).
usage: python tools/bench_conv.py [--only NAME] [--iters N] [--modes fwd,dgrad,wgrad]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mscl_amd import kernels as K, lib  # noqa: E402

SHAPES = [
    # name, (N,T,H,W,C), K, kernel, stride, pad
    ('l1_64_64', (8, 16, 56, 56, 64), 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('l1n1_64_64', (1, 16, 56, 56, 64), 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),      # 207 / 414 / 827 tiles: per-tile cost of the persistent layer-1 kernel
    ('l1n2_64_64', (2, 16, 56, 56, 64), 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('l1n4_64_64', (4, 16, 56, 56, 64), 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('l2_64_128_s2', (8, 16, 56, 56, 64), 128, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    ('l2_128_128', (8, 8, 28, 28, 128), 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('l3_128_256_s2', (8, 8, 28, 28, 128), 256, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    ('l3_256_256', (8, 4, 14, 14, 256), 256, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('l4_256_512_s2', (8, 4, 14, 14, 256), 512, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    ('l4_512_512', (8, 2, 7, 7, 512), 512, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('stem_rgb', (8, 16, 112, 112, 8), 64, (3, 7, 7), (1, 2, 2), (1, 3, 3)),
    ('stem_rgb_pairw', (8, 16, 112, 57, 8), 64, (3, 7, 4), (1, 2, 1), (1, 3, 1)),     # the same stem on W-paired input (kernels.pair_w)
    ('sepc_128', (8, 8, 28, 28, 128), 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('fpn_133', (8, 8, 28, 28, 128), 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('neck_lat_l3', (8, 4, 14, 14, 256), 128, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ('neck_lat_l4', (8, 2, 7, 7, 512), 128, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ('neck_lat_l2', (8, 8, 28, 28, 128), 128, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ('neck_333_p1', (8, 4, 14, 14, 128), 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('neck_333_p2', (8, 2, 7, 7, 128), 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('neck_133_p1', (8, 4, 14, 14, 128), 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('flow_stem', (8, 16, 112, 112, 8), 16, (1, 7, 7), (2, 2, 2), (0, 3, 3)),
    ('flow_l1', (8, 8, 56, 56, 16), 16, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('flow_l2', (8, 8, 28, 28, 32), 32, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('flow_l3', (8, 8, 14, 14, 64), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('flow_l4', (8, 8, 7, 7, 128), 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
]


# ResNet3dSlowOnly-50 at 8 x 32 x 224^2 (BASELINE configs[4]; --r50): thin-K convs on 205-MB maps, HBM-bound
SHAPES_R50 = [
    ('r50_stem_pairw', (8, 32, 224, 113, 8), 64, (5, 7, 4), (2, 2, 1), (2, 3, 1)),       # conv1 (5,7,7) / (2,2,2) of mscl_r50_cosm_lr3e-2.py:18 on W-paired input (implicit-GEMM kernel: conv_stem.hip covers kT <= 3)
    ('r50_stem_177_pairw', (8, 32, 224, 113, 8), 64, (1, 7, 4), (1, 2, 1), (0, 3, 1)),   # the (1,7,7) / (1,2,2) stem of the reference's default SlowOnly (conv_stem.hip)
    ('r50_l1_c1_64_64', (8, 16, 56, 56, 64), 64, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ('r50_l1_c2_133', (8, 16, 56, 56, 64), 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('r50_l1_c3_64_256', (8, 16, 56, 56, 64), 256, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ('r50_l1_c1_256_64', (8, 16, 56, 56, 256), 64, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ('r50_l2_c2_133_s2', (8, 16, 56, 56, 128), 128, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ('r50_l2_c2_133', (8, 16, 28, 28, 128), 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('r50_l3_c2_133', (8, 16, 14, 14, 256), 256, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('r50_l4_c2_133', (8, 16, 7, 7, 512), 512, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('r50_l2_c3_128_512', (8, 16, 28, 28, 128), 512, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ('r50_l2_c1_512_128', (8, 16, 28, 28, 512), 128, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ('r50_l3_c1_311', (8, 16, 14, 14, 1024), 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ('r50_l3_c3_256_1024', (8, 16, 14, 14, 256), 1024, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ('r50_l4_c1_311', (8, 16, 7, 7, 2048), 512, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ('r50_l4_c3_512_2048', (8, 16, 7, 7, 512), 2048, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
]


_STATS_POOL = None


def stats_slice(Kc, dev):
    """[2][K] statistics rows cut from a 16-MB buffer, as the step cuts them from kernels.ZEROS (what the statistics epilogue costs the
    layer-1 forward, and what does not cause it: conv_halo.hip)"""
    global _STATS_POOL
    if _STATS_POOL is None:
        _STATS_POOL = torch.zeros((4 << 20,), device=dev)
    return _STATS_POOL[:K.STAT_SLOTS * 2 * Kc].view(K.STAT_SLOTS, 2, Kc)[0]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def sweep(a):
    dev = torch.device('cuda:0')
    var, vals = a.sweep.split('=')
    vals = vals.split(',')
    modes = a.modes.split(',')
    print(f'# sweep {var} over {vals}: min of {a.rounds} rounds x {a.iters} launches, us per launch')
    for name, xs, Kc, kern, st, pad in SHAPES:
        if a.only and not any(o in name for o in a.only.split(',')):
            continue
        C = xs[-1]
        d = K.conv_desc(xs, Kc, kern, st, pad)
        x = torch.randn(xs, device=dev).to(torch.bfloat16)
        w = (torch.randn((Kc, *kern, C), device=dev) * 0.05).to(torch.bfloat16)
        wT = w.permute(4, 1, 2, 3, 0).contiguous()
        dy = torch.randn(K.out_shape(d), device=dev).to(torch.bfloat16)
        dw = torch.zeros((Kc, *kern, C), device=dev)
        stats = stats_slice(Kc, dev)
        fns = {'fwd': lambda: K.conv3d_fwd(x, w, d, stats=(stats[0], stats[1])), 'dgrad': lambda: K.conv3d_dgrad(dy, wT, d),
               'wgrad': lambda: K.conv3d_wgrad(x, dy, d, dw)}
        for m in modes:
            if m == 'dgrad' and C < 16:
                continue
            best = {v: 1e9 for v in vals}
            for _ in range(a.rounds):
                for v in vals:
                    lib.tune(**{var: None if v == '-' else v})     # (the library caches its switches: set + re-read)
                    best[v] = min(best[v], timeit(fns[m], a.iters) * 1e3)
            lib.tune(**{var: None})
            print(f'{name:16s} {m:6s} ' + '  '.join(f'{v}: {best[v]:7.1f}' for v in vals), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--modes', default='fwd,dgrad,wgrad')
    ap.add_argument('--sweep', default=None, help='NAME=v1,v2,..: A/B an environment tuning hook that the library reads per '
                    'launch (e.g. MSCL_PP=0,1,2), variants interleaved per shape in ONE process')
    ap.add_argument('--rounds', type=int, default=3, help='rounds per measurement (min is reported)')
    ap.add_argument('--r50', action='store_true', help='the ResNet3dSlowOnly-50 shapes at 8 x 32 x 224^2 instead (adds a GB/s column: x + y bytes)')
    ap.add_argument('--no-stats', action='store_true', help='forward without the BatchNorm statistics epilogue')
    a = ap.parse_args()
    if a.r50:
        SHAPES[:] = SHAPES_R50
    if a.sweep:
        return sweep(a)
    dev = torch.device('cuda:0')
    modes = a.modes.split(',')
    for name, xs, Kc, kern, st, pad in SHAPES:
        if a.only and not any(o in name for o in a.only.split(',')):
            continue
        C = xs[-1]
        d = K.conv_desc(xs, Kc, kern, st, pad)
        x = torch.randn(xs, device=dev).to(torch.bfloat16)
        w = (torch.randn((Kc, *kern, C), device=dev) * 0.05).to(torch.bfloat16)
        wT = w.permute(4, 1, 2, 3, 0).contiguous()
        dy = torch.randn(K.out_shape(d), device=dev).to(torch.bfloat16)
        dw = torch.zeros((Kc, *kern, C), device=dev)
        stats = stats_slice(Kc, dev)
        flops = 2.0 * d.N * d.To * d.Ho * d.Wo * Kc * kern[0] * kern[1] * kern[2] * C
        out = [f'{name:16s} {flops/1e9:7.2f} GF']
        nbytes = 2.0 * (x.numel() + dy.numel())
        gbs = (lambda ms: f' {nbytes/ms/1e6:6.0f} GB/s') if a.r50 else (lambda ms: '')
        # min over --rounds rounds of the mean of --iters launches: the first rounds of a process run at ramping clocks
        best = lambda fn: min(timeit(fn, a.iters) for _ in range(a.rounds))
        if 'fwd' in modes:
            st = None if a.no_stats else (stats[0], stats[1])
            ms = best(lambda: K.conv3d_fwd(x, w, d, stats=st))
            out.append(f'fwd {ms*1e3:8.1f} us {flops/ms/1e9:7.1f} TF' + gbs(ms))
        if 'dgrad' in modes and C >= 16:
            ms = best(lambda: K.conv3d_dgrad(dy, wT, d))
            out.append(f'dgrad {ms*1e3:8.1f} us {flops/ms/1e9:7.1f} TF' + gbs(ms))
        if 'wgrad' in modes:
            ms = best(lambda: K.conv3d_wgrad(x, dy, d, dw))
            out.append(f'wgrad {ms*1e3:8.1f} us {flops/ms/1e9:7.1f} TF' + gbs(ms))
        print('  '.join(out), flush=True)


if __name__ == '__main__':
    main()
