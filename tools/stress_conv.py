"""Stress probe for a producer -> consumer pair on ONE stream: a split-K conv (conv + finalize launch) followed by the trilinear
up-sampling that reads its output -- the pair behind the one-row differences tools/flake_det.py found -- launched many times with the
input ALTERNATING between two tensors, every up-sampled map compared bitwise with the reference of its input.  A consumer that sees
anything but the producer's finished output (the previous iteration's values, a half-written map) shows as a changed row.
usage: python tools/stress_conv.py [--iters N] [--graph 0|1] [--side 0|1]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mscl_amd import kernels as K  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20000)
    ap.add_argument('--graph', type=int, default=0)
    ap.add_argument('--side', type=int, default=0)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(1)
    xs, Kc, kern = (2, 2, 4, 4, 128), 128, (3, 3, 3)                      # PConv3D level 1 of the small test step: 64 rows, split-K
    w = (torch.randn((Kc, *kern, xs[-1]), generator=g) * (2.0 / (xs[-1] * np.prod(kern))) ** 0.5).to(dev).bfloat16()
    b = torch.randn((Kc,), generator=g).to(dev)
    d = K.conv_desc(xs, Kc, kern, (1, 1, 1), (1, 1, 1))
    x2 = [torch.randn(xs, generator=g).to(dev).bfloat16() for _ in range(2)]
    big = (2, 4, 8, 8, 128)

    def pair(x):
        y = K.conv3d_fwd(x, w, d, bias=b)
        up = torch.empty(big, dtype=torch.bfloat16, device=dev)
        K.upsample_add(y, up, True, accumulate=False)
        return up
    refs = [pair(x).clone() for x in x2]
    torch.cuda.synchronize()
    assert not torch.equal(refs[0], refs[1])
    side = torch.cuda.Stream()
    noise = torch.randn(1 << 20, device=dev)
    bad = 0
    if a.graph:
        # 64 iterations per replay: the input alternates INSIDE the captured graph through one static buffer
        xin = torch.empty(xs, dtype=torch.bfloat16, device=dev)
        outs = []
        gr = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            with torch.cuda.graph(gr):
                for i in range(64):
                    xin.copy_(x2[i % 2])
                    outs.append(pair(xin))
        torch.cuda.current_stream().wait_stream(st)
        for it in range(a.iters // 64):
            if a.side:
                with torch.cuda.stream(side):
                    for _ in range(8):
                        noise.mul_(1.0001)
            gr.replay()
            torch.cuda.synchronize()
            for i, o in enumerate(outs):
                if not torch.equal(o, refs[i % 2]):
                    bad += 1
                    idx = (o != refs[i % 2]).flatten().nonzero().flatten()
                    print(f'replay {it} pair {i}: {len(idx)} elements differ, rows {sorted({int(j) // 128 for j in idx})[:8]}', flush=True)
    else:
        for it in range(a.iters):
            if a.side and it % 4 == 0:
                with torch.cuda.stream(side):
                    noise.mul_(1.0001)
            o = pair(x2[it % 2])
            if not torch.equal(o, refs[it % 2]):          # (synchronises: the next launch starts on an idle device)
                bad += 1
                idx = (o != refs[it % 2]).flatten().nonzero().flatten()
                print(f'iter {it}: {len(idx)} elements differ, rows {sorted({int(j) // 128 for j in idx})[:8]}', flush=True)
    print(f'done: {bad} changed output(s) in {a.iters} iterations (graph={a.graph}, side={a.side})', flush=True)


if __name__ == '__main__':
    main()
