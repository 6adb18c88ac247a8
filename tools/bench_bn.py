"""Micro-benchmark of the BatchNorm passes on the maps of one mscl_r18 step (B=8, T=16, 112^2).
usage: python tools/bench_bn.py [--iters N]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mscl_amd import kernels as K  # noqa: E402
from mscl_amd.kernels import _bnp  # noqa: E402

MAPS = [('l1 56x56x64', (8, 16, 56, 56, 64)), ('l2 28x28x128', (8, 8, 28, 28, 128)), ('l3 14x14x256', (8, 4, 14, 14, 256)),
        ('l4 7x7x512', (8, 2, 7, 7, 512)), ('flow l1 56x56x16', (8, 8, 56, 56, 16))]


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
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=30)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    for name, shape in MAPS:
        C = shape[-1]
        n = 1
        for s in shape:
            n *= s
        mb = n * 2 / 1e6
        y = torch.randn(shape, device=dev).to(torch.bfloat16)
        res = torch.randn(shape, device=dev).to(torch.bfloat16)
        dout = torch.randn(shape, device=dev).to(torch.bfloat16)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        f = y.float().reshape(-1, C)
        st = torch.zeros((K.STAT_SLOTS, 2, C), device=dev)
        st[0, 0] = f.sum(0); st[0, 1] = (f * f).sum(0)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        nbt = torch.zeros((), dtype=torch.long, device=dev)
        sm, si = torch.empty(C, device=dev), torch.empty(C, device=dev)
        bn = _bnp((st[0, 0], st[0, 1]), g, b, rm, rv, nbt, sm, si)
        out = K.bn_act_fwd(y, bn, relu=True)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        scratch = torch.zeros(K.STAT_SLOTS * 4 * C, device=dev)
        rows = [
            ('fwd', lambda: K.bn_act_fwd(y, bn, relu=True), 2),
            ('fwd+res', lambda: K.bn_act_fwd(y, bn, residual=res, relu=True), 3),
            ('bwd mask(out)', lambda: K.bn_act_bwd(dout, out, y, g, sm, si, dg, db, True, scratch), 3 + 4),
            ('bwd mask(y)', lambda: K.bn_act_bwd(dout, None, y, g, sm, si, dg, db, True, scratch, beta=b), 2 + 3),
            ('bwd +dres', lambda: K.bn_act_bwd(dout, out, y, g, sm, si, dg, db, True, scratch, want_identity_dres=True), 3 + 5),
        ]
        for what, fn, maps in rows:
            us = timeit(fn, a.iters)
            print(f'{name:18s} {what:14s} {us:7.1f} us  {maps * mb / us * 1e3:7.0f} GB/s ({maps} maps of {mb:.1f} MB)')


if __name__ == '__main__':
    main()
