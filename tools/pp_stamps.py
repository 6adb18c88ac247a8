"""Where a phase of conv_pp_kernel spends its cycles: run with the -DPP_STAMP build of the library (mscl_amd/csrc/build/
libmscl_hip_stamp.so, see conv_pp.hip) and print, per shape, the mean cycles per phase of the L section, the barrier wait behind
it, the M section and the barrier wait behind that, for waves 0-3 and waves 4-7.
usage: MSCL_LIB=mscl_amd/csrc/build/libmscl_hip_stamp.so python tools/pp_stamps.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mscl_amd import kernels as K, lib  # noqa: E402

SHAPES = [('l2_128_128', (8, 8, 28, 28, 128), 128), ('l3_256_256', (8, 4, 14, 14, 256), 256), ('l4_512_512', (8, 2, 7, 7, 512), 512)]


def main():
    dev = torch.device('cuda:0')
    h = lib.load()
    if not hasattr(h, 'mscl_debug_pp_stamps'):
        raise SystemExit('this library has no stamps: build conv_pp.hip with -DPP_STAMP and point MSCL_LIB at it')
    h.mscl_debug_pp_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for name, xs, Kc in SHAPES:
        d = K.conv_desc(xs, Kc, (3, 3, 3), (1, 1, 1), (1, 1, 1))
        x = torch.randn(xs, device=dev).to(torch.bfloat16)
        w = (torch.randn((Kc, 3, 3, 3, xs[-1]), device=dev) * 0.05).to(torch.bfloat16)
        for _ in range(5):
            K.conv3d_fwd(x, w, d)
        torch.cuda.synchronize()
        rows = d.N * d.To * d.Ho * (d.Wo + 2)
        mt = (rows + 253) // 254
        nt = Kc // 128
        blocks = mt * nt
        ng = 9 * (xs[-1] // 64)
        ks = 1
        if blocks <= 128:
            ks = min(256 // blocks, ng // 2, 16)
        nblk = min(2048, blocks * ks)
        buf = (ctypes.c_ulonglong * (nblk * 16))()
        rc = h.mscl_debug_pp_stamps(buf, nblk * 16)
        assert rc == 0, rc
        t = torch.tensor(list(buf), dtype=torch.float64).view(nblk, 2, 8)
        phases = 3.0 * ng / ks
        m = t.mean(0) / phases
        print(f'{name}: {nblk} blocks, {phases:.0f} phases per block; cycles per phase  L / wait / M / wait:  '
              f'waves 0-3 {m[0, 0]:.0f} / {m[0, 1]:.0f} / {m[0, 2]:.0f} / {m[0, 3]:.0f}   '
              f'waves 4-7 {m[1, 0]:.0f} / {m[1, 1]:.0f} / {m[1, 2]:.0f} / {m[1, 3]:.0f}   (M alone = 512 cycles of MFMA issue)', flush=True)
        wm = t.mean(0)
        print(f'    whole block, cycles (waves 0-3): prologue {wm[0, 4]:.0f}, loop {wm[0, 6] - wm[0, 4] - wm[0, 5]:.0f}, epilogue {wm[0, 5]:.0f}, '
              f'total {wm[0, 6]:.0f};  block totals min / max over blocks {t[:, 0, 6].min():.0f} / {t[:, 0, 6].max():.0f}', flush=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            K.conv3d_fwd(x, w, d)
        e1.record(); torch.cuda.synchronize()
        print(f'    launch (stamp build, incl. split-K finalize where used): {e0.elapsed_time(e1) * 100:.1f} us', flush=True)


if __name__ == '__main__':
    main()
