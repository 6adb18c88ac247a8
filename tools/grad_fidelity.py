"""Where does the per-tensor gradient cosine of a bf16 pipeline against the fp32 reference go?  CPU experiment on the oracle
(test infrastructure): the fp32 step is re-run with bf16 rounding switched on at one more place per variant --
  ops    conv operands (x, w, dy) rounded to bf16, fp32 accumulate, every stored map fp32          (what MFMA alone costs)
  act    ops + forward maps stored in bf16 (conv outputs, BatchNorm outputs)
  grad   act + gradient maps stored in bf16 (every conv's input gradient)                          (= the HIP pipeline's storage)
  fix12  grad, but the input gradients of the stem / layer1 / layer2 convs stay fp32                 (the proposed remedy)
and the per-tensor cosine of the RGB trunk's conv-weight gradients against fp32 is printed per variant.
usage: python tools/grad_fidelity.py [--B 2] [--T 8] [--H 112]"""
import argparse
import os
import statistics
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mscl_amd.synthetic import synthetic_batch      # noqa: E402
from oracle import fill as ofill, mscl as om        # noqa: E402

MODE = {'ops': False, 'act': False, 'grad': False, 'keep_fp32': ()}


def r16(t):
    return t.to(torch.bfloat16).float()


class ConvBF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, pad, keep):
        xb, wb = r16(x), r16(w)
        y = F.conv3d(xb, wb, b, stride, pad)
        ctx.save_for_backward(xb, wb)
        ctx.cfg = (stride, pad, keep, b is not None)
        return r16(y) if MODE['act'] else y

    @staticmethod
    def backward(ctx, dy):
        xb, wb = ctx.saved_tensors
        stride, pad, keep, has_b = ctx.cfg
        dyb = r16(dy)
        dx = torch.nn.grad.conv3d_input(xb.shape, wb, dyb, stride, pad)
        dw = torch.nn.grad.conv3d_weight(xb, wb.shape, dyb, stride, pad)
        if MODE['grad'] and not keep:
            dx = r16(dx)
        return dx, dw, (dyb.sum(dim=(0, 2, 3, 4)) if has_b else None), None, None, None


_orig_conv_forward = nn.Conv3d.forward
_orig_bn_forward = nn.BatchNorm3d.forward


def conv_forward(self, x):
    if not MODE['ops']:
        return _orig_conv_forward(self, x)
    return ConvBF.apply(x, self.weight, self.bias, self.stride, self.padding, getattr(self, '_keep_fp32', False))


def bn_forward(self, x):
    y = _orig_bn_forward(self, x)
    return r16(y) if MODE['act'] else y


def run(batch, T, K, mode, keep=()):
    MODE.update(ops=False, act=False, grad=False)
    MODE.update(mode)
    m = om.MSCLWithAug(num_frames=T, K=K); ofill.fill_module(m); m.train()
    for n, mod in m.named_modules():
        if isinstance(mod, nn.Conv3d):
            mod._keep_fp32 = any(n.startswith(k) for k in keep)
    torch.manual_seed(100)
    out = m.train_step(batch)
    out['loss'].backward()
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}, float(out['loss'])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=2); ap.add_argument('--T', type=int, default=8); ap.add_argument('--H', type=int, default=112)
    a = ap.parse_args()
    batch = synthetic_batch(a.B, a.T, a.H, a.H, 0, 0)
    nn.Conv3d.forward = conv_forward
    nn.BatchNorm3d.forward = bn_forward
    ref, lref = run(batch, a.T, 64, {})
    early = ('recognizer.encoder_q.stem', 'recognizer.encoder_q.layer1', 'recognizer.encoder_q.layer2')
    variants = [('ops', dict(ops=True), ()), ('act', dict(ops=True, act=True), ()), ('grad', dict(ops=True, act=True, grad=True), ()),
                ('fix12', dict(ops=True, act=True, grad=True), early)]
    cos = torch.nn.functional.cosine_similarity
    print(f'B={a.B} T={a.T} H={a.H}; fp32 loss {lref:.5f}')
    print(f'{"variant":8s} {"loss":>10s} {"min":>7s} {"median":>7s}   per-stage median of the RGB trunk conv-weight cosines (stem, l1, l2, l3, l4)')
    for name, mode, keep in variants:
        g, l = run(batch, a.T, 64, mode, keep)
        per = {}
        for n, v in g.items():
            if n.startswith('recognizer.encoder_q.') and v.dim() == 5:
                st = n.split('.')[2]
                per.setdefault(st, []).append(float(cos(v.flatten(), ref[n].flatten(), dim=0)))
        allc = [c for v in per.values() for c in v]
        stages = '  '.join(f'{k}:{statistics.median(v):.4f}' for k, v in per.items())
        print(f'{name:8s} {l:10.5f} {min(allc):7.4f} {statistics.median(allc):7.4f}   {stages}', flush=True)


if __name__ == '__main__':
    main()
