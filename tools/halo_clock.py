"""In-kernel clock of the layer-1 halo conv (diagnostic build libprobe_clock.so, -DHALO_CLOCK): per-block cycles and GHz."""
import ctypes
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mscl_amd import kernels as K, lib  # noqa: E402

dev = torch.device('cuda:0')
xs = (8, 16, 56, 56, 64)
d = K.conv_desc(xs, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1))
x = torch.randn(xs, device=dev).to(torch.bfloat16)
w = (torch.randn((64, 3, 3, 3, 64), device=dev) * 0.05).to(torch.bfloat16)
wT = w.permute(4, 1, 2, 3, 0).contiguous()
zeros = '--zeros' in sys.argv
if zeros:
    x.zero_(); w.zero_(); wT.zero_()
t_end = time.time() + 2.0
while time.time() < t_end:                      # >= 2 s of back-to-back launches on the data under test
    for _ in range(50):
        K.conv3d_dgrad(x, wT, d)
    torch.cuda.synchronize()
h = lib.load()
n = 256 if os.environ.get('MSCL_HALO_PERSIST') == '1' else 1664
buf = (ctypes.c_ulonglong * (2 * n))()
rc = h.mscl_halo_clock_read(buf, 2 * n)
cyc = [buf[2 * i] for i in range(n)]
ghz = [buf[2 * i] / buf[2 * i + 1] * 0.1 for i in range(n) if buf[2 * i + 1]]
print(f'rc {rc} data {"zeros" if zeros else "random"}: block cycles median {statistics.median(cyc):.0f}, in-kernel clock median {statistics.median(ghz):.3f} GHz '
      f'(min {min(ghz):.3f}, max {max(ghz):.3f}); MFMA-only bound 13824 cycles per plane tile; blocks {n}')

if hasattr(h, 'mscl_halo_seg_read') and os.environ.get('HALO_SEG') == '1':
    sb = (ctypes.c_ulonglong * (5 * 256))()
    h.mscl_halo_seg_read(sb, 5 * 256)
    names = ['k-step-0 MFMA issue', 'wait vmcnt/lgkmcnt', 'barrier', 'operand reads + DMA issue', 'k-step-1 MFMA issue']
    for q in range(5):
        vals = [sb[5 * b + q] / 27 for b in range(256)]
        print(f'  {names[q]:28s} {statistics.median(vals):7.0f} cycles per tap (median over 256 blocks, wave 0)')
