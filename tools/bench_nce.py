"""Micro-benchmark of the queue passes (K=65536, dim 128): GB/s of queue reads against the 8 TB/s HBM roofline."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mscl_amd import kernels as K  # noqa: E402

dev = torch.device('cuda:0')
Kq, dim = 65536, 128
queue = torch.nn.functional.normalize(torch.randn(dim, Kq, device=dev), dim=0)
count = torch.randint(1, 5000, (Kq,), device=dev, dtype=torch.long)
for R in (8, 24):
    q = torch.nn.functional.normalize(torch.randn(R, dim, device=dev), dim=1)
    pos = torch.rand(R, device=dev)
    ones = torch.ones(R, device=dev)

    def fwd():
        return K.nce_forward(queue, count, q, pos, 1 / 0.07)

    lse, _, _ = fwd()

    def bwd():
        return K.nce_backward(queue, count, q, lse, ones, 1 / 0.07)

    for name, fn in (('fwd', fwd), ('bwd', bwd)):
        # device time: 20 calls captured into ONE HIP graph and replayed (eager calls are host-paced: two launches and four
        # allocations per call cost the host more than the kernels take)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g):
                for _ in range(20):
                    fn()
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 100 * 1e3
        print(f'R={R:2d} {name}: {us:7.1f} us   {dim * Kq * 4 / us / 1e3:7.1f} GB/s of queue reads (graph replay, incl. the finish / slab-sum launch)')
