"""Per-kernel parity: every HIP kernel (through the C ABI) against plain PyTorch fp32 on the CPU.

The CPU side computes in fp32 from the SAME bf16-rounded inputs, so the only differences are the
accumulation order (fp32) and the final rounding of bf16 outputs: tolerance = 2^-7 relative to the
tensor's max magnitude for bf16 outputs, 2e-4 relative for fp32 outputs (stated per test).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BF16_TOL = 2.0 ** -7
F32_TOL = 3e-4


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale)


def bf(x):
    return x.to(torch.bfloat16)


def close(got, want, tol, what=''):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = want.abs().max().item() + 1e-12
    err = (got - want).abs().max().item()
    assert err <= tol * scale, f'{what}: max err {err:.4g} vs scale {scale:.4g} (rel {err/scale:.3g} > {tol:.3g})'


CONV_CASES = [
    # name, N,T,H,W, C, K, kernel, stride, pad
    ('r3d_64_64_s1', 2, 4, 12, 12, 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('r3d_64_128_s2', 2, 4, 12, 12, 64, 128, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    ('r3d_128_128_s1', 1, 3, 9, 10, 128, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('r3d_256_256_s1', 1, 2, 7, 7, 256, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('ds_64_128', 2, 4, 12, 12, 64, 128, (1, 1, 1), (2, 2, 2), (0, 0, 0)),
    ('flow_16_16', 2, 3, 14, 14, 16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('flow_16_32_s2', 2, 3, 14, 14, 16, 32, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ('flow_32_64_s2', 2, 3, 14, 14, 32, 64, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ('flow_ds_16_32', 2, 3, 14, 14, 16, 32, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
    ('fpn_133', 1, 4, 10, 10, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('lat_512_128', 1, 2, 7, 7, 512, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ('stem_rgb', 1, 4, 20, 20, 8, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3)),
    ('stem_flow', 1, 4, 20, 20, 8, 16, (1, 7, 7), (2, 2, 2), (0, 3, 3)),
    ('big_m_tail', 3, 5, 13, 11, 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('l1_plane_56', 1, 3, 56, 56, 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),      # real layer-1 plane: 13 halo tiles, last partial
    ('lat_bias_long', 2, 4, 48, 48, 16, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0)),    # 18 432 rows: the bias-gradient column sum takes a full and a partial trip
    # Bottleneck trunks (mscl_r50): 1x1x1 widening / narrowing, inflated 3x1x1, 1x3x3 with the spatial stride, 8-channel r2d_50 layers
    ('r50_64_256', 2, 3, 12, 12, 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ('r50_1024_256_311', 1, 4, 6, 6, 1024, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ('r50_512_2048', 1, 2, 5, 5, 512, 2048, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ('r50_128_128_133_s2', 2, 3, 14, 14, 128, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ('r50_ds_256_512_s2', 2, 3, 14, 14, 256, 512, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
    ('r2d50_8_8_133', 2, 3, 14, 14, 8, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('r2d50_8_32', 2, 3, 14, 14, 8, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ('r2d50_32_8', 2, 3, 14, 14, 32, 8, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
]


def _conv_ref(x, w, stride, pad, bias=None):
    """x (N,T,H,W,C) float, w (K,kT,kH,kW,C) float -> (N,To,Ho,Wo,K) float"""
    y = F.conv3d(x.permute(0, 4, 1, 2, 3), w.permute(0, 4, 1, 2, 3), bias=bias, stride=stride, padding=pad)
    return y.permute(0, 2, 3, 4, 1).contiguous()


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(case, dev):
    from mscl_amd import kernels as K_
    name, N, T, H, W, C, K, kern, stride, pad = case
    x = bf(rnd((N, T, H, W, C), 1)); w = bf(rnd((K, *kern, C), 2, scale=(2.0 / (C * np.prod(kern))) ** 0.5))
    if name.startswith('stem'):
        x[..., 3:] = 0
    d = K_.conv_desc(x.shape, K, kern, stride, pad)
    xg, wg = x.to(dev), w.to(dev)
    # forward + BN statistics
    st = torch.zeros((K_.STAT_SLOTS, 2, K), device=dev)          # [slots][2][C]: producers spread over the slots
    y = K_.conv3d_fwd(xg, wg, d, stats=(st[0, 0], st[0, 1]))
    ssum, ssq = st[:, 0].sum(0), st[:, 1].sum(0)
    xr = x.float().requires_grad_(True); wr = w.float().requires_grad_(True)
    yr = _conv_ref(xr, wr, stride, pad)
    close(y, yr, BF16_TOL, 'conv fwd')
    close(ssum, yr.sum(dim=(0, 1, 2, 3)), 2e-3, 'bn sum')
    close(ssq, (yr * yr).sum(dim=(0, 1, 2, 3)), 2e-3, 'bn sumsq')
    # forward with bias + addend + relu
    b = rnd((K,), 3); a = bf(rnd(tuple(yr.shape), 4))
    y2 = K_.conv3d_fwd(xg, wg, d, bias=b.to(dev), addend=a.to(dev), relu=True)
    close(y2, F.relu(yr.detach() + b + a.float()), BF16_TOL, 'conv fwd epilogue')
    # backward
    dy = bf(rnd(tuple(yr.shape), 5))
    yr.backward(dy.float())
    taps = int(np.prod(kern))
    if not name.startswith('stem'):
        wT = torch.empty((C, *kern, K), dtype=torch.bfloat16, device=dev)
        K_.weight_transpose(wg, wT, K, taps, C)
        assert torch.equal(wT.cpu(), w.permute(4, 1, 2, 3, 0).contiguous())
        dx = K_.conv3d_dgrad(dy.to(dev), wT, d)
        close(dx, xr.grad, BF16_TOL, 'conv dgrad')
        add = bf(rnd(tuple(x.shape), 6))
        dx2 = K_.conv3d_dgrad(dy.to(dev), wT, d, addend=add.to(dev))
        close(dx2, xr.grad + add.float(), BF16_TOL, 'conv dgrad+addend')
    dw = torch.zeros((K, *kern, C), dtype=torch.float32, device=dev)
    db = torch.zeros((K,), dtype=torch.float32, device=dev)
    K_.conv3d_wgrad(xg, dy.to(dev), d, dw, db)
    close(dw, wr.grad, F32_TOL, 'conv wgrad')
    close(db, dy.float().sum(dim=(0, 1, 2, 3)), F32_TOL, 'conv dbias')
    K_.conv3d_wgrad(xg, dy.to(dev), d, dw, None)           # accumulates
    close(dw, 2 * wr.grad, F32_TOL, 'conv wgrad accumulate')


# The layer shapes of one mscl_r18 step (B=8, T=16, 112^2) that select the big-map kernel instantiations (the 8-wave 256 x 128
# tile, the ping-pong shared-tap kernel, conv_wgrad_kernel<128,128,2>, the parity-class input gradient at full depth): the small
# CONV_CASES never reach them.  CPU fp32 F.conv3d is the reference (a few seconds each on the box's host cores).
REAL_CASES = [
    # the layer that carries 46 % of the step's FLOPs, at its real map: the DEFAULT window-resident kernels (conv_halo64_kernel
    # forward / input gradient, wgrad_halo64_kernel), asserted by their launch counters below
    ('real_l1_64_64', 8, 16, 56, 56, 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),           # r3d.py:16-34 layer1 convs
    ('real_l2_128_128', 8, 8, 28, 28, 128, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),        # r3d.py:16-34 layer2 convs, sepc.py Pconv level 0
    ('real_l2_entry_64_128_s2', 8, 16, 56, 56, 64, 128, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    ('real_l3_256_256', 8, 4, 14, 14, 256, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('real_l4_512_512', 8, 2, 7, 7, 512, 512, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('real_fpn_133', 8, 8, 28, 28, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
]


@pytest.mark.parametrize('case', REAL_CASES, ids=[c[0] for c in REAL_CASES])
def test_conv_real_layer_shapes(case, dev):
    from mscl_amd import lib
    n_halo, n_wh = lib.call_raw('mscl_debug_halo_launches'), lib.call_raw('mscl_debug_wgrad_halo_launches')
    n_s2 = lib.call_raw('mscl_debug_dgrad_s2_launches')
    test_conv_fwd_dgrad_wgrad(case, dev)
    if case[0] in ('real_l2_128_128', 'real_l3_256_256'):
        # since round 4 the 128- / 256-channel 3x3x3 layers run the window-resident weight gradient as 64 x 64 channel slices
        assert lib.call_raw('mscl_debug_wgrad_halo_launches') == n_wh + 2, f'{case[0]} did not take the window-resident weight gradient'
    if case[0] == 'real_l2_entry_64_128_s2':
        assert lib.call_raw('mscl_debug_dgrad_s2_launches') == n_s2 + 2, 'the layer-2 entry input gradient did not take the window-resident kernel'
    if case[0] == 'real_l1_64_64':
        # the forward with statistics, 2 input gradients (plain / + addend), 2 weight gradients (the forward with bias + ReLU is the
        # implicit-GEMM family's: the window-resident kernel has no such epilogue)
        assert lib.call_raw('mscl_debug_halo_launches') == n_halo + 3, 'layer 1 did not take the window-resident conv kernel'
        assert lib.call_raw('mscl_debug_wgrad_halo_launches') == n_wh + 2, 'layer 1 did not take the window-resident weight gradient'


# conv_k1.hip (persistent thin-K 1x1x1 streaming kernel; round 5): the forward of a widening conv K -> 4 K (64 -> 256, 128 -> 512, 256 -> 1024, the
# `conv3` of the Bottleneck blocks, resnet3d.py:262-296) and the input gradient of a narrowing one (256 -> 64, 512 -> 128: `conv1`), with a
# row count that is a multiple of the 128-row tile, one that leaves a tail of ONE row, one that gives some blocks one tile and others
# two, and a map long enough for every block to walk several tiles.  Launch counters assert the kernel family.
K1_CASES = [
    # name, N,T,H,W, C (in), K (out)
    ('k1_64_256', 1, 2, 56, 56, 64, 256),               # 6272 rows = 49 tiles
    ('k1_64_256_tail', 1, 1, 65, 65, 64, 256),          # 4225 rows: 33 tiles + 1 row
    ('k1_128_512', 1, 2, 56, 56, 128, 512),             # two channel tiles share the rows
    ('k1_128_512_tail', 1, 3, 41, 43, 128, 512),        # 5289 rows
    ('k1_128_256', 2, 2, 40, 40, 128, 256),             # the first block of layer 2: 256 -> 128 narrowing, gradient 128 -> 256
    ('k1_64_256_long', 2, 8, 56, 56, 64, 256),          # 50176 rows = 392 tiles over 192 blocks
    ('k1_256_1024', 2, 4, 28, 28, 256, 1024),           # K = 256: 64-row tiles, four channel tiles share the rows
    ('k1_256_1024_tail', 1, 3, 37, 41, 256, 1024),      # 4551 rows: 71 tiles + 7 rows
]


@pytest.mark.parametrize('case', K1_CASES, ids=[c[0] for c in K1_CASES])
def test_conv_k1(case, dev):
    from mscl_amd import kernels as K_, lib
    name, N, T, H, W, C, K = case
    kern, one, zero = (1, 1, 1), (1, 1, 1), (0, 0, 0)
    x = bf(rnd((N, T, H, W, C), 21)); w = bf(rnd((K, *kern, C), 22, scale=(2.0 / C) ** 0.5))
    d = K_.conv_desc(x.shape, K, kern, one, zero)
    xg, wg = x.to(dev), w.to(dev)
    n0 = lib.call_raw('mscl_debug_k1_launches')
    st = torch.zeros((K_.STAT_SLOTS, 2, K), device=dev)
    y = K_.conv3d_fwd(xg, wg, d, stats=(st[0, 0], st[0, 1]))
    assert lib.call_raw('mscl_debug_k1_launches') == n0 + 1, 'the forward did not take the thin-K kernel'
    yr = _conv_ref(x.float(), w.float(), one, zero)
    close(y, yr, BF16_TOL, 'k1 fwd')
    close(st[:, 0].sum(0), yr.sum(dim=(0, 1, 2, 3)), 2e-3, 'k1 bn sum')
    close(st[:, 1].sum(0), (yr * yr).sum(dim=(0, 1, 2, 3)), 2e-3, 'k1 bn sumsq')
    a = bf(rnd(tuple(yr.shape), 24))
    close(K_.conv3d_fwd(xg, wg, d, addend=a.to(dev)), yr + a.float(), BF16_TOL, 'k1 fwd + addend')
    assert lib.call_raw('mscl_debug_k1_launches') == n0 + 2
    # the narrowing conv K -> C whose input gradient is this GEMM: dx (.., K) = dy (.., C) . w2[C][K]
    w2 = bf(rnd((C, *kern, K), 25, scale=(2.0 / K) ** 0.5))
    d2 = K_.conv_desc((N, T, H, W, K), C, kern, one, zero)
    wT = torch.empty((K, *kern, C), dtype=torch.bfloat16, device=dev)
    K_.weight_transpose(w2.to(dev), wT, C, 1, K)
    dy = bf(rnd((N, T, H, W, C), 26)); add = bf(rnd((N, T, H, W, K), 27))
    n1 = lib.call_raw('mscl_debug_k1_launches')
    dx = K_.conv3d_dgrad(dy.to(dev), wT, d2)
    dx2 = K_.conv3d_dgrad(dy.to(dev), wT, d2, addend=add.to(dev))
    assert lib.call_raw('mscl_debug_k1_launches') == n1 + 2, 'the input gradient did not take the thin-K kernel'
    dxr = dy.float().reshape(-1, C) @ w2.float().reshape(C, K)
    close(dx.reshape(-1, K), dxr, BF16_TOL, 'k1 dgrad')
    close(dx2.reshape(-1, K), dxr + add.float().reshape(-1, K), BF16_TOL, 'k1 dgrad + addend')
    # the same launch again, many times: a race between the staged tiles and the fragment reads shows as a changed output
    y0 = K_.conv3d_fwd(xg, wg, d)
    for _ in range(10):
        assert torch.equal(K_.conv3d_fwd(xg, wg, d), y0)


# conv_pp.hip (ping-pong, shared W taps) forced onto small shapes: row tails, a map smaller than one tile, split-K over the
# (kt, kh, channel part) groups with the finalize pass, kT = 1, strides along T / H (forward only: the strided input gradient is
# the parity-class kernel's), 256 / 512 channels (2 / 4 channel tiles, 4 / 8 channel parts per tap).
PP_CASES = [
    ('pp_128_128', 1, 3, 9, 10, 128, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1), 1),
    ('pp_128_128_tail', 2, 3, 13, 11, 128, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1), 1),
    ('pp_64_128', 2, 4, 12, 12, 64, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1), 1),
    ('pp_256_256_split', 1, 2, 7, 7, 256, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1), 0),
    ('pp_256_256_split3', 1, 2, 7, 7, 256, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1), 3),
    ('pp_512_512', 1, 2, 7, 7, 512, 512, (3, 3, 3), (1, 1, 1), (1, 1, 1), 0),
    ('pp_fpn_133', 1, 4, 10, 10, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), 1),
    ('pp_128_256_s221', 2, 4, 12, 12, 128, 256, (3, 3, 3), (2, 2, 1), (1, 1, 1), 1),
    ('pp_w28', 1, 2, 28, 28, 128, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1), 1),
    ('pp_64_64', 2, 3, 13, 11, 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), 1),            # 64 output channels: the 8 x 1 wave layout
    ('pp_64_64_plane56', 1, 3, 56, 56, 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), 1),
    ('pp_128_64', 1, 3, 9, 10, 128, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), 1),
]


@pytest.mark.parametrize('case', PP_CASES, ids=[c[0] for c in PP_CASES])
def test_conv_pp_forced(case, dev, monkeypatch):
    from mscl_amd import kernels as K_, lib
    name, N, T, H, W, C, K, kern, stride, pad, ksplit = case
    monkeypatch.setenv('MSCL_PP', '2')
    monkeypatch.setenv('MSCL_HALO', '0')          # (the window-resident layer-1 kernel would take the 64 -> 64 cases first)
    if ksplit:
        monkeypatch.setenv('MSCL_PP_KSPLIT', str(ksplit))
    lib.tune()                                    # the library caches its switches: re-read (conftest re-reads after the test)
    x = bf(rnd((N, T, H, W, C), 11)); w = bf(rnd((K, *kern, C), 12, scale=(2.0 / (C * np.prod(kern))) ** 0.5))
    d = K_.conv_desc(x.shape, K, kern, stride, pad)
    xg, wg = x.to(dev), w.to(dev)
    n0 = lib.call_raw('mscl_debug_pp_launches')
    st = torch.zeros((K_.STAT_SLOTS, 2, K), device=dev)
    y = K_.conv3d_fwd(xg, wg, d, stats=(st[0, 0], st[0, 1]))
    assert lib.call_raw('mscl_debug_pp_launches') == n0 + 1, 'the forward did not take the ping-pong kernel'
    xr = x.float().requires_grad_(True); wr = w.float()
    yr = _conv_ref(xr, wr, stride, pad)
    close(y, yr, BF16_TOL, 'pp fwd')
    close(st[:, 0].sum(0), yr.sum(dim=(0, 1, 2, 3)), 2e-3, 'pp bn sum')
    close(st[:, 1].sum(0), (yr * yr).sum(dim=(0, 1, 2, 3)), 2e-3, 'pp bn sumsq')
    b = rnd((K,), 13); a = bf(rnd(tuple(yr.shape), 14))
    y2 = K_.conv3d_fwd(xg, wg, d, bias=b.to(dev), addend=a.to(dev), relu=True)
    close(y2, F.relu(yr.detach() + b + a.float()), BF16_TOL, 'pp fwd epilogue')
    if stride == (1, 1, 1):
        dy = bf(rnd(tuple(yr.shape), 15))
        yr.backward(dy.float())
        wT = torch.empty((C, *kern, K), dtype=torch.bfloat16, device=dev)
        K_.weight_transpose(wg, wT, K, int(np.prod(kern)), C)
        n1 = lib.call_raw('mscl_debug_pp_launches')
        add = bf(rnd(tuple(x.shape), 16))
        dx = K_.conv3d_dgrad(dy.to(dev), wT, d, addend=add.to(dev))
        assert lib.call_raw('mscl_debug_pp_launches') == n1 + (C % 64 == 0), 'the input gradient did not take the ping-pong kernel'
        close(dx, xr.grad + add.float(), BF16_TOL, 'pp dgrad+addend')
    # the same launches again, many times: a race between the LDS-DMA ring and the fragment reads shows as a changed output
    y0 = K_.conv3d_fwd(xg, wg, d)
    for _ in range(20):
        assert torch.equal(K_.conv3d_fwd(xg, wg, d), y0)


# conv_halo64b_kernel (conv_halo.hip: two blocks per CU, one window slot, half-tap operand pipeline, ring of two 2-tap weight stages):
# planes smaller than a tile and of several tiles with a ragged last one, the widest plane the window takes (W = 61), T = 1 / 2 (a
# missing neighbour plane on one or both sides: a zero window), forward with statistics and with an addend, input gradient with and
# without addend, and the same launch many times over (a race between the window / ring DMA and the fragment reads shows as a
# changed output).
HALO2_CASES = [('h2_small', 2, 4, 12, 12), ('h2_plane56', 1, 3, 56, 56), ('h2_tail', 3, 5, 13, 11), ('h2_w61', 1, 2, 9, 61), ('h2_T1', 2, 1, 20, 20)]


@pytest.mark.parametrize('kt', [3, 1], ids=['k333', 'k133'])
@pytest.mark.parametrize('case', HALO2_CASES, ids=[c[0] for c in HALO2_CASES])
def test_conv_halo_two_blocks(case, kt, dev, monkeypatch):
    """the window-resident layer-1 conv (the A/B arms of round 4 -- one block per CU, one-tap ring stages of depth 2 / 3 / 4 -- were
    deleted in round 5 with their switches); kt = 1 (round 6): the same kernel on ONE source plane, the 1x3x3 / pad (0,1,1) conv2 of
    the 64-channel Bottlenecks"""
    from mscl_amd import kernels as K_, lib
    name, N, T, H, W = case
    C = K = 64
    kern, stride, pad = (kt, 3, 3), (1, 1, 1), (kt // 2, 1, 1)
    monkeypatch.setenv('MSCL_HALO', '1')
    lib.tune()
    x = bf(rnd((N, T, H, W, C), 51)); w = bf(rnd((K, *kern, C), 52, scale=(2.0 / (C * 9 * kt)) ** 0.5))
    d = K_.conv_desc(x.shape, K, kern, stride, pad)
    xg, wg = x.to(dev), w.to(dev)
    n0 = lib.call_raw('mscl_debug_halo_launches')
    st = torch.zeros((K_.STAT_SLOTS, 2, K), device=dev)
    y = K_.conv3d_fwd(xg, wg, d, stats=(st[0, 0], st[0, 1]))
    assert lib.call_raw('mscl_debug_halo_launches') == n0 + 1, 'the forward did not take the window-resident kernel'
    xr = x.float().requires_grad_(True); wr = w.float()
    yr = _conv_ref(xr, wr, stride, pad)
    close(y, yr, BF16_TOL, 'halo2 fwd')
    close(st[:, 0].sum(0), yr.sum(dim=(0, 1, 2, 3)), 2e-3, 'halo2 bn sum')
    close(st[:, 1].sum(0), (yr * yr).sum(dim=(0, 1, 2, 3)), 2e-3, 'halo2 bn sumsq')
    a = bf(rnd(tuple(yr.shape), 54))
    close(K_.conv3d_fwd(xg, wg, d, addend=a.to(dev)), yr.detach() + a.float(), BF16_TOL, 'halo2 fwd + addend')
    dy = bf(rnd(tuple(yr.shape), 55))
    yr.backward(dy.float())
    wT = torch.empty((C, *kern, K), dtype=torch.bfloat16, device=dev)
    K_.weight_transpose(wg, wT, K, 9 * kt, C)
    n1 = lib.call_raw('mscl_debug_halo_launches')
    dx = K_.conv3d_dgrad(dy.to(dev), wT, d)
    add = bf(rnd(tuple(x.shape), 56))
    dx2 = K_.conv3d_dgrad(dy.to(dev), wT, d, addend=add.to(dev))
    assert lib.call_raw('mscl_debug_halo_launches') == n1 + 2, 'the input gradient did not take the window-resident kernel'
    close(dx, xr.grad, BF16_TOL, 'halo2 dgrad')
    close(dx2, xr.grad + add.float(), BF16_TOL, 'halo2 dgrad + addend')
    y0 = K_.conv3d_fwd(xg, wg, d)
    for _ in range(20):
        assert torch.equal(K_.conv3d_fwd(xg, wg, d), y0)


# conv_wgrad_halo.hip (window-resident weight gradient on 64 x 64 channel slices) forced onto small shapes (MSCL_WGRAD_HALO_MIN=1: a
# plane smaller than one 256-position tile is a tile of mostly zero rows): one and several channel slices each way, more blocks
# than slots and fewer, several items per block (tile-major ranges that cross sample and tile boundaries), T = 1 / 2 (planes
# without a neighbour on one or both sides: every item of kt = 0 / 2 skipped), a plane of several tiles with a ragged last one,
# accumulation into a non-zero dw with a bias gradient, and the same bits on every run (slot-ordered slab sums).
WGRAD_HALO_CASES = [
    ('wh_64_64', 2, 4, 12, 12, 64, 64),
    ('wh_64_64_T1', 3, 1, 10, 9, 64, 64),
    ('wh_64_64_plane56', 1, 3, 56, 56, 64, 64),
    ('wh_128_128', 1, 3, 9, 10, 128, 128),
    ('wh_128_64', 2, 3, 13, 11, 128, 64),
    ('wh_64_256_T2', 2, 2, 11, 13, 64, 256),
    ('wh_256_256', 1, 2, 7, 7, 256, 256),
    ('wh_128_128_plane28', 2, 3, 28, 28, 128, 128),
    ('wh_256_128_many_items', 8, 4, 14, 14, 256, 128),
]


@pytest.mark.parametrize('kt', [3, 1], ids=['k333', 'k133'])
@pytest.mark.parametrize('case', WGRAD_HALO_CASES, ids=[c[0] for c in WGRAD_HALO_CASES])
def test_conv_wgrad_halo_forced(case, kt, dev, monkeypatch):
    """kt = 1 (round 6): the same kernel with ONE temporal tap / pad 0 -- the 1x3x3 conv2 of the Bottleneck trunks"""
    from mscl_amd import kernels as K_, lib
    name, N, T, H, W, C, K = case
    kern, stride, pad = (kt, 3, 3), (1, 1, 1), (kt // 2, 1, 1)
    monkeypatch.setenv('MSCL_WGRAD_HALO_MIN', '1')            # planes of any size
    monkeypatch.setenv('MSCL_WGRAD_HALO_ITEMS', '0')          # ... and any number of them
    lib.tune()
    x = bf(rnd((N, T, H, W, C), 41)); w = rnd((K, *kern, C), 42)
    d = K_.conv_desc(x.shape, K, kern, stride, pad)
    xr = x.float(); wr = w.clone().requires_grad_(True)
    yr = _conv_ref(xr, wr, stride, pad)
    dy = bf(rnd(tuple(yr.shape), 43))
    yr.backward(dy.float())
    dw = torch.zeros((K, *kern, C), dtype=torch.float32, device=dev)
    db = torch.zeros((K,), dtype=torch.float32, device=dev)
    n0 = lib.call_raw('mscl_debug_wgrad_halo_launches')
    K_.conv3d_wgrad(x.to(dev), dy.to(dev), d, dw, db)
    assert lib.call_raw('mscl_debug_wgrad_halo_launches') == n0 + 1, 'the weight gradient did not take the window-resident kernel'
    close(dw, wr.grad, F32_TOL, 'wgrad_halo')
    close(db, dy.float().sum(dim=(0, 1, 2, 3)), F32_TOL, 'wgrad_halo dbias')
    first = dw.clone()
    K_.conv3d_wgrad(x.to(dev), dy.to(dev), d, dw, None)           # accumulates; slot-ordered slab sums: the same bits every run
    assert torch.equal(dw, 2 * first)
    for _ in range(5):
        dw2 = torch.zeros_like(dw)
        K_.conv3d_wgrad(x.to(dev), dy.to(dev), d, dw2, None)
        assert torch.equal(dw2, first)


# conv_thin.hip (window-resident 1x3x3 kernel of the 16- / 32-channel maps): every channel pairing, bands that divide the plane
# by 8, by 7 and not at all (a short last band), planes narrower than a 16-position tile row, the flow trunk's real layer1 / layer2
# maps with two statistics groups.
THIN_CASES = [
    ('thin_16_16', 2, 3, 16, 20, 16, 16, 1),
    ('thin_16_32', 2, 2, 14, 14, 16, 32, 1),
    ('thin_32_16', 1, 3, 9, 10, 32, 16, 1),
    ('thin_32_32', 2, 2, 21, 12, 32, 32, 2),
    ('thin_short_band', 2, 1, 11, 5, 16, 16, 2),
    ('thin_flow_l1', 4, 8, 56, 56, 16, 16, 2),
    ('thin_flow_l2', 4, 8, 28, 28, 32, 32, 2),
]


@pytest.mark.parametrize('case', THIN_CASES, ids=[c[0] for c in THIN_CASES])
def test_conv_thin(case, dev):
    import ctypes
    from mscl_amd import kernels as K_, lib
    name, N, T, H, W, C, K, groups = case
    kern, stride, pad = (1, 3, 3), (1, 1, 1), (0, 1, 1)
    x = bf(rnd((N, T, H, W, C), 61)); w = bf(rnd((K, *kern, C), 62, scale=(2.0 / (C * 9)) ** 0.5))
    d = K_.conv_desc(x.shape, K, kern, stride, pad)
    xg, wg = x.to(dev), w.to(dev)
    n0 = lib.call_raw('mscl_debug_thin_launches')
    st = torch.zeros((groups, K_.STAT_SLOTS, 2, K), device=dev)
    y = torch.empty(K_.out_shape(d), dtype=torch.bfloat16, device=dev)
    lib.call('mscl_conv3d_fwd_groups', ctypes.byref(d), xg.data_ptr(), wg.data_ptr(), y.data_ptr(), None, None, 0,
             st.data_ptr(), st.data_ptr() + 4 * K, groups, None, 0, lib.stream_ptr())
    assert lib.call_raw('mscl_debug_thin_launches') == n0 + 1, 'the forward did not take the window-resident kernel'
    xr = x.float().requires_grad_(True); wr = w.float()
    yr = _conv_ref(xr, wr, stride, pad)
    close(y, yr, BF16_TOL, 'thin fwd')
    yg = yr.detach().view(groups, N // groups, T, H, W, K)
    close(st[:, :, 0].sum(1), yg.sum(dim=(1, 2, 3, 4)), 2e-3, 'thin bn sum')
    close(st[:, :, 1].sum(1), (yg * yg).sum(dim=(1, 2, 3, 4)), 2e-3, 'thin bn sumsq')
    b = rnd((K,), 63); a = bf(rnd(tuple(yr.shape), 64))
    y2 = K_.conv3d_fwd(xg, wg, d, bias=b.to(dev), addend=a.to(dev), relu=True)
    close(y2, F.relu(yr.detach() + b + a.float()), BF16_TOL, 'thin fwd epilogue')
    dy = bf(rnd(tuple(yr.shape), 65))
    yr.backward(dy.float())
    wT = torch.empty((C, *kern, K), dtype=torch.bfloat16, device=dev)
    K_.weight_transpose(wg, wT, K, 9, C)
    add = bf(rnd(tuple(x.shape), 66))
    n1 = lib.call_raw('mscl_debug_thin_launches')
    dx = K_.conv3d_dgrad(dy.to(dev), wT, d, addend=add.to(dev))
    assert lib.call_raw('mscl_debug_thin_launches') == n1 + 1, 'the input gradient did not take the window-resident kernel'
    close(dx, xr.grad + add.float(), BF16_TOL, 'thin dgrad+addend')
    # weight gradient: accumulates into a non-zero dw, fixed-order slab sums (the same bits every run), bias gradient beside it
    dw = torch.zeros((K, *kern, C), dtype=torch.float32, device=dev); db = torch.zeros((K,), dtype=torch.float32, device=dev)
    n2 = lib.call_raw('mscl_debug_thin_wgrad_launches')
    K_.conv3d_wgrad(xg, dy.to(dev), d, dw, db)
    assert lib.call_raw('mscl_debug_thin_wgrad_launches') == n2 + 1, 'the weight gradient did not take the window-resident kernel'
    wr2 = w.float().requires_grad_(True)
    _conv_ref(x.float(), wr2, stride, pad).backward(dy.float())
    close(dw, wr2.grad, F32_TOL, 'thin wgrad'); close(db, dy.float().sum(dim=(0, 1, 2, 3)), F32_TOL, 'thin dbias')
    first = dw.clone()
    K_.conv3d_wgrad(xg, dy.to(dev), d, dw, None)
    assert torch.equal(dw, 2 * first)
    # against the implicit-GEMM kernel on the same inputs: same products, fp32 sums in another order
    import os
    lib.tune(MSCL_THIN=0)
    try:
        close(K_.conv3d_fwd(xg, wg, d), y, 2.0 ** -7, 'thin vs implicit GEMM')
    finally:
        lib.tune(MSCL_THIN=None)


@pytest.mark.parametrize('W', [24, 23])
def test_stem_w_paired_equals_plain_stem(W, dev):
    """RGB stem (r3d.py:176-184: Conv3d(3,64,(3,7,7),(1,2,2),(1,3,3))) run on W-paired input (mscl_pair_w): same outputs,
    same weight gradient after folding the paired staging buffer back to (64,3,7,7,3)"""
    from mscl_amd import kernels as K_
    N, T, H, Co = 2, 4, 20, 64
    Wp = (W + 1) // 2 + 1
    x3 = bf(rnd((N, T, H, W, 3), 1)); w = bf(rnd((Co, 3, 7, 7, 3), 2, scale=(2.0 / (3 * 147)) ** 0.5))      # physical (Cout,kT,kH,kW,Cin)
    x8 = torch.zeros((N, T, H, W, 8), dtype=torch.bfloat16); x8[..., :3] = x3
    xr = x3.float().requires_grad_(True); wr = w.float().requires_grad_(True)
    yr = _conv_ref(xr, wr, (1, 2, 2), (1, 3, 3))
    xp = K_.pair_w(x8.to(dev))
    assert tuple(xp.shape) == (N, T, H, Wp, 8)
    exp = torch.zeros((N, T, H, Wp, 2, 3), dtype=torch.bfloat16)                  # pair j = pixels 2j-1, 2j
    odd, even = x3[:, :, :, 1::2], x3[:, :, :, 0::2]
    exp[:, :, :, 1:1 + odd.shape[3], 0] = odd; exp[:, :, :, :even.shape[3], 1] = even
    assert torch.equal(xp.cpu()[..., :6], exp.reshape(N, T, H, Wp, 6)) and not xp.cpu()[..., 6:].any()
    w8 = torch.zeros((Co, 3, 7, 4, 8), dtype=torch.bfloat16, device=dev)
    K_.pair_w_weight(w.to(dev), w8)
    d = K_.conv_desc(tuple(xp.shape), Co, (3, 7, 4), (1, 2, 1), (1, 3, 1))
    st = torch.zeros((K_.STAT_SLOTS, 2, Co), device=dev)
    y = K_.conv3d_fwd(xp, w8, d, stats=(st[0, 0], st[0, 1]))
    close(y, yr, BF16_TOL, 'paired stem fwd')
    close(st[:, 0].sum(0), yr.sum(dim=(0, 1, 2, 3)), 2e-3, 'paired stem bn sum')
    dy = bf(rnd(tuple(yr.shape), 5)); yr.backward(dy.float())
    dw8 = torch.zeros((Co, 3, 7, 4, 8), dtype=torch.float32, device=dev)
    K_.conv3d_wgrad(xp, dy.to(dev), d, dw8, None)
    g = torch.zeros((Co, 3, 7, 7, 3), dtype=torch.float32, device=dev)
    K_.pair_w_grad_fold(dw8, g)
    close(g, wr.grad, F32_TOL, 'paired stem wgrad')
    assert float(dw8[..., 6:].abs().max()) == 0.0          # (slot j = 3, p = 1 is kw = 7: real pixels, no kernel column -- dropped by the fold)


# conv_stem.hip (window-resident forward of the W-paired RGB stems): the real planes of both configurations (default dispatch), an
# odd width, a temporal stride, a plane smaller than one tile (forced), a tile tail behind the plane
STEM_CASES = [
    # name, N, T, H, W, kT, sT, pT, forced
    ('r18_112', 2, 4, 112, 112, 3, 1, 1, False),
    ('r50_224', 1, 3, 224, 224, 1, 1, 0, False),
    ('edge_90', 1, 3, 90, 90, 3, 1, 1, False),
    ('odd_w_70x57', 2, 3, 70, 57, 3, 1, 1, False),
    ('stride_t2', 1, 5, 64, 64, 3, 2, 1, False),
    ('small_20x24', 2, 4, 20, 24, 3, 1, 1, True),
    ('small_kt1_22x23', 2, 2, 22, 23, 1, 1, 0, True),
    ('r50_cfg_kt5_224', 1, 6, 224, 224, 5, 2, 2, False),      # conv1 of mscl_r50_cosm_lr3e-2.py:18: (5,7,7) / (2,2,2) / pad (2,3,3): three window pieces per plane
    ('kt5_112_tail', 2, 5, 112, 96, 5, 2, 2, False),         # two pieces per plane, an odd frame count
    ('small_kt5_20x24', 1, 7, 20, 24, 5, 2, 2, True),
]


@pytest.mark.parametrize('case', STEM_CASES, ids=[c[0] for c in STEM_CASES])
def test_conv_stem_window_resident(case, dev):
    """the stems of r3d.py:176-184 (3,7,7) and ResNet3dSlowOnly conv1 (1,7,7) through mscl_conv3d_fwd on W-paired input: the
    window-resident kernel takes the launch (counter), equals CPU fp32 F.conv3d on the 3-channel clip, and repeats bit for bit"""
    from mscl_amd import kernels as K_
    from mscl_amd import lib
    name, N, T, H, W, kT, sT, pT, forced = case
    Co = 64
    x3 = bf(rnd((N, T, H, W, 3), 1)); w = bf(rnd((Co, kT, 7, 7, 3), 2, scale=(2.0 / (3 * 49 * kT)) ** 0.5))
    x8 = torch.zeros((N, T, H, W, 8), dtype=torch.bfloat16); x8[..., :3] = x3
    yr = _conv_ref(x3.float(), w.float(), (sT, 2, 2), (pT, 3, 3))
    xp = K_.pair_w(x8.to(dev))
    w8 = torch.zeros((Co, kT, 7, 4, 8), dtype=torch.bfloat16, device=dev)
    K_.pair_w_weight(w.to(dev), w8)
    d = K_.conv_desc(tuple(xp.shape), Co, (kT, 7, 4), (sT, 2, 1), (pT, 3, 1))
    if forced:
        lib.tune(MSCL_STEM=1)
    try:
        n0 = lib.call_raw('mscl_debug_stem_launches')
        st = torch.zeros((K_.STAT_SLOTS, 2, Co), device=dev)
        y = K_.conv3d_fwd(xp, w8, d, stats=(st[0, 0], st[0, 1]))
        assert lib.call_raw('mscl_debug_stem_launches') == n0 + 1, 'the window-resident stem kernel did not take the launch'
        assert tuple(y.shape) == tuple(yr.shape)
        close(y, yr, BF16_TOL, f'stem fwd {name}')
        close(st[:, 0].sum(0), yr.sum(dim=(0, 1, 2, 3)), 2e-3, 'stem bn sum')
        close(st[:, 1].sum(0), (yr * yr).sum(dim=(0, 1, 2, 3)), 2e-3, 'stem bn sumsq')
        for _ in range(3):                                   # DMA ring / window reuse: repeated launches are bit-identical
            assert torch.equal(K_.conv3d_fwd(xp, w8, d), y)
        # epilogue variants the kernel does not cover stay with the implicit-GEMM kernel
        b = rnd((Co,), 3)
        y2 = K_.conv3d_fwd(xp, w8, d, bias=b.to(dev), relu=True)
        assert lib.call_raw('mscl_debug_stem_launches') == n0 + 4
        close(y2, F.relu(yr + b), BF16_TOL, 'stem fwd with bias + relu (implicit GEMM)')
        lib.tune(MSCL_STEM=0)
        close(K_.conv3d_fwd(xp, w8, d), y, 2.0 ** -7, 'window-resident stem vs implicit GEMM')
        assert lib.call_raw('mscl_debug_stem_launches') == n0 + 4
    finally:
        lib.tune(MSCL_STEM=None)


@pytest.mark.parametrize('C,relu,resmode', [(64, True, 'none'), (64, True, 'identity'), (128, True, 'bn'),
                                            (16, False, 'none'), (32, True, 'identity'), (512, True, 'bn')])
def test_bn_act_fwd_bwd(C, relu, resmode, dev):
    from mscl_amd import kernels as K_
    from mscl_amd.kernels import _bnp
    shape = (2, 3, 5, 7, C)
    rows = 2 * 3 * 5 * 7
    y = bf(rnd(shape, 1) * 1.5 + 0.3)
    gamma = 1 + 0.1 * rnd((C,), 2); beta = 0.1 * rnd((C,), 3)
    res = bf(rnd(shape, 4)) if resmode != 'none' else None
    rgamma = 1 + 0.1 * rnd((C,), 5); rbeta = 0.1 * rnd((C,), 6)
    # reference: torch BatchNorm in training mode on NCTHW fp32
    def bn_ref(inp, g, b_):
        m = torch.nn.BatchNorm3d(C)
        m.weight.data.copy_(g); m.bias.data.copy_(b_)
        m.train()
        return m, m(inp.permute(0, 4, 1, 2, 3)).permute(0, 2, 3, 4, 1)
    yr = y.float().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    m1, z = bn_ref(yr, gamma, beta)
    rr = None
    if resmode == 'identity':
        rr = res.float().requires_grad_(True); z = z + rr
    elif resmode == 'bn':
        rr = res.float().requires_grad_(True); m2, z2 = bn_ref(rr, rgamma, rbeta); z = z + z2
    out_ref = F.relu(z) if relu else z
    # device
    def stats_of(t):                       # [slots][2][C], the sums split unevenly over two of the active slots
        from mscl_amd.kernels import STAT_ACTIVE, STAT_SLOTS
        f = t.float().to(dev).reshape(-1, C)
        st = torch.zeros((STAT_SLOTS, 2, C), device=dev)
        h = f.shape[0] // 3
        st[0, 0], st[0, 1] = f[:h].sum(0), (f[:h] * f[:h]).sum(0)
        st[STAT_ACTIVE - 1, 0], st[STAT_ACTIVE - 1, 1] = f[h:].sum(0), (f[h:] * f[h:]).sum(0)
        return st[0, 0], st[0, 1], st
    mk = lambda: dict(rm=torch.zeros(C, device=dev), rv=torch.ones(C, device=dev),
                      nbt=torch.zeros((), dtype=torch.long, device=dev), sm=torch.empty(C, device=dev), si=torch.empty(C, device=dev))
    s1 = mk(); g1, b1 = gamma.to(dev), beta.to(dev)
    st1 = stats_of(y)                      # keep alive: BnParams only holds raw pointers
    bn = _bnp(st1, g1, b1, s1['rm'], s1['rv'], s1['nbt'], s1['sm'], s1['si'])
    rbn = None
    if resmode == 'bn':
        s2 = mk(); g2, b2 = rgamma.to(dev), rbeta.to(dev)
        st2 = stats_of(res)
        rbn = _bnp(st2, g2, b2, s2['rm'], s2['rv'], s2['nbt'], s2['sm'], s2['si'])
    out = K_.bn_act_fwd(y.to(dev), bn, residual=res.to(dev) if res is not None else None, res_bn=rbn, relu=relu)
    close(out, out_ref, BF16_TOL, 'bn fwd')
    close(s1['rm'], m1.running_mean, 1e-4, 'running_mean'); close(s1['rv'], m1.running_var, 1e-4, 'running_var')
    assert int(s1['nbt']) == 1
    if resmode == 'bn':
        close(s2['rm'], m2.running_mean, 1e-4, 'res running_mean'); close(s2['rv'], m2.running_var, 1e-4, 'res running_var')
    # backward: use the device's own bf16 `out` for the relu mask on both sides
    dout = bf(rnd(shape, 7))
    mask = (out.float().cpu() > 0).float() if relu else torch.ones(shape)
    z.backward(dout.float() * mask)
    dgamma = torch.zeros(C, device=dev); dbeta = torch.zeros(C, device=dev)
    scratch = torch.zeros(K_.STAT_SLOTS * 4 * C, device=dev)
    resd = None
    if resmode == 'bn':
        rdg = torch.zeros(C, device=dev); rdb = torch.zeros(C, device=dev)
        resd = dict(y=res.to(dev), gamma=g2, mean=s2['sm'], invstd=s2['si'], dgamma=rdg, dbeta=rdb)
    dy, dres = K_.bn_act_bwd(dout.to(dev), out, y.to(dev), g1, s1['sm'], s1['si'], dgamma, dbeta, relu, scratch,
                             res=resd, want_identity_dres=(resmode == 'identity'))
    close(dy, yr.grad, 2 * BF16_TOL, 'bn dy')
    close(dgamma, m1.weight.grad, 2e-3, 'dgamma'); close(dbeta, m1.bias.grad, 2e-3, 'dbeta')
    if relu and resmode == 'none':          # same pass with the ReLU mask recomputed from y (no read of `out`)
        dg2 = torch.zeros(C, device=dev); db2 = torch.zeros(C, device=dev)
        dy2, _ = K_.bn_act_bwd(dout.to(dev), None, y.to(dev), g1, s1['sm'], s1['si'], dg2, db2, relu,
                               torch.zeros(K_.STAT_SLOTS * 4 * C, device=dev), beta=b1)
        # the two masks differ only where bn(y) is within bf16 rounding of 0
        frac = (dy2.float() != dy.float()).float().mean().item()
        assert frac < 0.02, frac
        close(dy2, yr.grad, 3 * BF16_TOL, 'bn dy (mask from y)')
        close(dg2, m1.weight.grad, 4e-3, 'dgamma (mask from y)'); close(db2, m1.bias.grad, 4e-3, 'dbeta (mask from y)')
    if resmode == 'identity':
        close(dres, rr.grad, BF16_TOL, 'identity dres')
    if resmode == 'bn':
        close(dres, rr.grad, 2 * BF16_TOL, 'bn dres')
        close(rdg, m2.weight.grad, 2e-3, 'res dgamma'); close(rdb, m2.bias.grad, 2e-3, 'res dbeta')


def test_bn_act_fwd_large_map(dev):
    """a map larger than one sweep of the 2048-block grid: the unrolled loop takes a full and a partial trip"""
    from mscl_amd import kernels as K_
    from mscl_amd.kernels import STAT_SLOTS, _bnp
    C, shape = 64, (2, 4, 96, 96, 64)                 # 589 824 granules > 2048 * 256
    y = bf(rnd(shape, 11) * 1.5 + 0.3).to(dev); res = bf(rnd(shape, 12)).to(dev)
    gamma = (1 + 0.1 * rnd((C,), 2)).to(dev); beta = (0.1 * rnd((C,), 3)).to(dev)
    f = y.float().reshape(-1, C)
    st = torch.zeros((STAT_SLOTS, 2, C), device=dev)
    st[0, 0] = f.sum(0); st[0, 1] = (f * f).sum(0)
    mean = f.mean(0); var = f.var(0, unbiased=False)
    ref = torch.relu((y.float() - mean) * torch.rsqrt(var + 1e-5) * gamma + beta + res.float())
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    nbt = torch.zeros((), dtype=torch.long, device=dev)
    sm, si = torch.empty(C, device=dev), torch.empty(C, device=dev)
    bn = _bnp((st[0, 0], st[0, 1]), gamma, beta, rm, rv, nbt, sm, si)
    out = K_.bn_act_fwd(y, bn, residual=res, relu=True)
    close(out, ref, BF16_TOL, 'bn_act_fwd large map')
    out = K_.bn_act_fwd(y, bn, relu=False)
    close(out, (y.float() - mean) * torch.rsqrt(var + 1e-5) * gamma + beta, BF16_TOL, 'bn_act_fwd large map, plain')


def test_pack_add_relu_pool(dev):
    from mscl_amd import kernels as K_
    x = torch.rand(2, 3, 4, 6, 5, generator=torch.Generator().manual_seed(0))
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    out = K_.pack_input(x.to(dev), mean, std)
    ref = torch.zeros(2, 4, 6, 5, 8)
    ref[..., :3] = ((x - torch.tensor(mean).view(1, 3, 1, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1, 1)).permute(0, 2, 3, 4, 1)
    close(out, ref, BF16_TOL, 'pack_input')
    assert torch.equal(out[..., 3:].cpu().float(), torch.zeros(2, 4, 6, 5, 5))
    out2 = K_.pack_input(x.to(dev))
    close(out2[..., :3], x.permute(0, 2, 3, 4, 1), BF16_TOL, 'pack_input raw')
    out3 = K_.pack_input(x.to(dev), t_off=2, T=2)            # second half of the frames, no copy
    close(out3[..., :3], x[:, :, 2:].permute(0, 2, 3, 4, 1), BF16_TOL, 'pack_input window')
    a, b, c = (bf(rnd((2, 3, 4, 5, 16), s)) for s in (1, 2, 3))
    close(K_.add_relu(a.to(dev), b.to(dev), c.to(dev), relu=True), F.relu(a.float() + b.float() + c.float()), BF16_TOL, 'add3 relu')
    close(K_.add_relu(a.to(dev), b.to(dev)), a.float() + b.float(), BF16_TOL, 'add2')
    close(K_.relu_bwd(a.to(dev), b.to(dev)), a.float() * (b.float() > 0), BF16_TOL, 'relu bwd')
    xm = bf(rnd((4, 37, 128), 4))
    p = K_.pool_fwd(xm.to(dev), 4, 37)
    close(p, xm.float().mean(1), 1e-5, 'pool fwd')
    dp = rnd((4, 128), 5)
    dx = K_.pool_bwd(dp.to(dev), (4, 37, 128), 4, 37)
    close(dx, (dp / 37).unsqueeze(1).expand(4, 37, 128), BF16_TOL, 'pool bwd')
    base = bf(rnd((4, 37, 128), 6)).to(dev)
    dx2 = K_.pool_bwd(dp.to(dev), (4, 37, 128), 4, 37, into=base.clone())
    close(dx2, base.float().cpu() + (dp / 37).unsqueeze(1), BF16_TOL, 'pool bwd accumulate')


@pytest.mark.parametrize('tri', [False, True])
def test_upsample(tri, dev):
    from mscl_amd import kernels as K_
    mode = 'trilinear' if tri else 'nearest'
    # scales of exactly 2 (the necks), 1 along some axes, between 1 and 2, and above 2 (the general backward kernel)
    for (ss, ds) in (((2, 7, 7), (4, 14, 14)), ((4, 14, 14), (8, 28, 28)), ((2, 3, 5), (3, 7, 9)), ((2, 7, 7), (4, 7, 7)),
                     ((3, 5, 6), (5, 9, 11)), ((1, 1, 3), (2, 2, 6))):
        src = bf(rnd((2, *ss, 16), 1)); dst = bf(rnd((2, *ds, 16), 2))
        sr = src.float().requires_grad_(True)
        up = F.interpolate(sr.permute(0, 4, 1, 2, 3), size=ds, mode=mode).permute(0, 2, 3, 4, 1)
        got = K_.upsample_add(src.to(dev), dst.to(dev).clone(), tri, accumulate=True)
        close(got, up + dst.float(), BF16_TOL, f'upsample add {mode} {ss}->{ds}')
        got = K_.upsample_add(src.to(dev), torch.empty_like(dst, device=dev), tri, accumulate=False)
        close(got, up, BF16_TOL, f'upsample {mode}')
        dd = bf(rnd((2, *ds, 16), 3))
        up.backward(dd.float())
        close(K_.upsample_bwd(dd.to(dev), tuple(src.shape), tri), sr.grad, BF16_TOL, f'upsample bwd {mode}')


def test_linear_l2norm(dev):
    from mscl_amd import kernels as K_
    for rows, i, o, relu in ((8, 512, 512, True), (8, 512, 128, False), (2, 128, 128, True), (16, 128, 128, False),
                             (20, 100, 40, True), (32, 300, 70, False), (1, 64, 8, True)):
        x = rnd((rows, i), 1).requires_grad_(True); w = (rnd((o, i), 2) / i ** 0.5).requires_grad_(True); b = rnd((o,), 3).requires_grad_(True)
        y = F.linear(x, w, b); y = F.relu(y) if relu else y
        yd = K_.linear_fwd(x.detach().to(dev), w.detach().to(dev), b.detach().to(dev), relu)
        close(yd, y, F32_TOL, 'linear fwd')
        dy = rnd((rows, o), 4); y.backward(dy)
        dw = torch.zeros(o, i, device=dev); db = torch.zeros(o, device=dev)
        dx = K_.linear_bwd(x.detach().to(dev), w.detach().to(dev), yd, dy.to(dev), dw, db, relu)
        close(dx, x.grad, F32_TOL, 'linear dx'); close(dw, w.grad, F32_TOL, 'linear dw'); close(db, b.grad, F32_TOL, 'linear db')
    x = rnd((8, 128), 5).requires_grad_(True)
    y = F.normalize(x, dim=1); dy = rnd((8, 128), 6); y.backward(dy)
    yd, nr = K_.l2norm_fwd(x.detach().to(dev))
    close(yd, y, 1e-6, 'l2norm fwd')
    close(K_.l2norm_bwd(yd, nr, dy.to(dev)), x.grad, 1e-5, 'l2norm bwd')


@pytest.mark.parametrize('R,K', [(8, 1024), (24, 65536), (16, 4096), (3, 640), (32, 2048), (96, 4096), (45, 640), (24, 2050), (40, 1022)])
def test_nce(R, K, dev):
    """R = 96 is the three stacked row groups of the shipped config's videos_per_gpu = 32 (mscl_r18_cosm_lr2e-2.py:50): rows beyond
    32 run as further row tiles over the same snapshot; 45 = one full + one ragged tile.  K = 2050 / 1022 (even, not a multiple of
    4) are the shapes the fp32-MFMA kernels leave to the vector kernels (a lane of theirs reads two columns, not four); every other
    K runs on the matrix cores (round 6)."""
    from mscl_amd import kernels as K_
    dim, T = 128, 0.07
    queue = F.normalize(rnd((dim, K), 1), dim=0)
    count = torch.randint(0, 9000, (K,), generator=torch.Generator().manual_seed(2))
    q = F.normalize(rnd((R, dim), 3), dim=1).requires_grad_(True)
    kpos = F.normalize(rnd((R, dim), 4), dim=1)
    pos = (q * kpos).sum(1)
    w = queue * (0.99999 ** (1.0 * count))
    logits = torch.cat([pos[:, None], q @ w], 1) / T
    loss_rows = F.cross_entropy(logits, torch.zeros(R, dtype=torch.long), reduction='none')
    rank = (logits[:, 1:] > logits[:, :1]).sum(1)
    lse, lr, rk = K_.nce_forward(queue.to(dev), count.to(dev), q.detach().to(dev), pos.detach().to(dev), 1.0 / T)
    close(lse, torch.logsumexp(logits, 1), 1e-5, 'lse')
    close(lr, loss_rows, 2e-5, 'nce loss rows')
    # the rank is a count of fp32 comparisons: a negative whose logit sits within rounding of the positive's may fall on either side
    # (the MFMA chain, the vector chain and the CPU matmul add the 128 products in different orders), every other one may not
    l64 = torch.cat([pos.double()[:, None], q.double() @ w.double()], 1) / T
    near = ((l64[:, 1:] - l64[:, :1]).abs() <= 4e-6 * l64[:, :1].abs().clamp_min(1.0)).sum(1)
    assert bool(((rk.cpu().long() - rank).abs() <= near).all()), (rk.cpu(), rank, near)
    assert int(near.sum()) <= max(4, R), near      # (the allowance stays a handful of near-ties per pass -- ~1e-4 wide windows among 65536 logits -- not a tolerance on the count)
    scale = rnd((R,), 5).abs() + 0.1
    # gradient of sum_r scale_r * loss_r wrt q through the NEGATIVE logits only
    neg_only = (torch.logsumexp(torch.cat([pos.detach()[:, None], q @ w], 1) / T, 1) * scale).sum()
    gq, = torch.autograd.grad(neg_only, q)
    dq = K_.nce_backward(queue.to(dev), count.to(dev), q.detach().to(dev), lse, scale.to(dev), 1.0 / T)
    close(dq, gq, 1e-4, 'nce dq')
    # full gradient incl. the positive key: d/dq of sum_r scale_r * CE_r with k_pos held constant
    posd = K_.rowdot(q.detach().to(dev), kpos.to(dev))
    close(posd, pos, 1e-5, 'rowdot')
    full = (F.cross_entropy(torch.cat([(q * kpos).sum(1, keepdim=True), q @ w], 1) / T,
                            torch.zeros(R, dtype=torch.long), reduction='none') * scale).sum()
    gfull, = torch.autograd.grad(full, q)
    K_.nce_pos_bwd(kpos.to(dev), posd, lse, scale.to(dev), dq, 1.0 / T)
    close(dq, gfull, 1e-4, 'nce dq incl. positive')
    # the same in ONE call (round 5: the slab-sum launch adds the positive pair's term)
    dq1 = K_.nce_backward(queue.to(dev), count.to(dev), q.detach().to(dev), lse, scale.to(dev), 1.0 / T, pos_pair=(kpos.to(dev), posd))
    close(dq1, gfull, 1e-4, 'nce dq incl. positive, one call')


def test_loss_pack_positive_logits(dev):
    """mscl_loss_pack's `pos` output: <query row, key row> of every row of the three passes, bit-identical to mscl_rowdot on the rows
    the pack lays out (the pack launch replaced three rowdot launches on the loss phase)"""
    from mscl_amd import kernels as K_
    B, D, t, Cf = 8, 128, 4, 64
    for use_aug in (True, False):
        q_rgb, q_fb, q_fa, k_rgb, k_fb, k_fa = (F.normalize(rnd((B, D), 10 + i), dim=1).to(dev) for i in range(6))
        p_fb, p_fa = rnd((B * t, Cf), 20).to(dev), rnd((B * t, Cf), 21).to(dev)
        QA, KA, sA, QC, KC, sC, ones, flow, ws, (posA, posB, posC) = K_.loss_pack(q_rgb, q_fb, q_fa, k_rgb, k_fb, k_fa, p_fb, p_fa, t, use_aug, 0.5)
        n = 3 if use_aug else 2
        assert posA.shape == (n * B,) and posB.shape == (B,) and posC.shape == (n * B,)
        assert torch.equal(posA, K_.rowdot(QA, KA)) and torch.equal(posB, K_.rowdot(q_fb, k_fb)) and torch.equal(posC, K_.rowdot(QC, KC))
        assert torch.equal(QA[B:2 * B], q_fb) and torch.equal(KC[:B], k_fa) and torch.equal(KC[B:2 * B], k_fb)


def test_enqueue_bit_exact(dev):
    from mscl_amd import kernels as K_
    dim, K, n = 128, 64, 8
    queue = rnd((dim, K), 1); count = torch.zeros(K, dtype=torch.long); ptr = torch.zeros(1, dtype=torch.long)
    qd, cd, pd = queue.to(dev), count.to(dev), ptr.to(dev)
    for step in range(11):                       # wraps around K=64 after 8 steps
        keys = rnd((n, dim), 10 + step)
        K_.queue_enqueue(qd, cd, pd, keys.to(dev))
        count += 1; p = int(ptr); queue[:, p:p + n] = keys.T; count[p:p + n] = 1; ptr[0] = (p + n) % K
        assert torch.equal(cd.cpu(), count) and torch.equal(pd.cpu(), ptr) and torch.equal(qd.cpu(), queue)
    with pytest.raises(Exception):
        K_.queue_enqueue(qd, cd, pd, rnd((7, dim), 0).to(dev))      # K % n != 0 (moco.py:432 assert)
    # the benchmark's queue (260 blocks; the one that finishes last moves the pointer): five enqueues back to back, no host sync between
    K = 65536
    queue = rnd((dim, K), 2); count = torch.zeros(K, dtype=torch.long); ptr = torch.tensor([K - 16])
    qd, cd, pd = queue.to(dev), count.to(dev), ptr.to(dev)
    keys = [rnd((n, dim), 30 + i) for i in range(5)]
    kd = [k.to(dev) for k in keys]
    for k in kd:
        K_.queue_enqueue(qd, cd, pd, k)
    for k in keys:
        count += 1; p = int(ptr); queue[:, p:p + n] = k.T; count[p:p + n] = 1; ptr[0] = (p + n) % K
    assert torch.equal(cd.cpu(), count) and torch.equal(pd.cpu(), ptr) and torch.equal(qd.cpu(), queue)


@pytest.mark.parametrize('B,t', [(2, 4), (8, 8), (3, 2)])
def test_lmcl(B, t, dev):
    from mscl_amd import kernels as K_
    C, T = 128, 0.07
    rgb = rnd((B, t, C), 1).requires_grad_(True); flow = rnd((B, 2 * t, C), 2).requires_grad_(True)
    xr = F.normalize(rgb.transpose(1, 2), dim=1); xf = F.normalize(flow.transpose(1, 2), dim=1)
    sim = torch.bmm(xr.transpose(1, 2), xf).flatten(0, 1) / T
    labels = torch.arange(t).repeat(B)
    loss = F.cross_entropy(sim, labels)
    loss.backward()
    pos = sim[torch.arange(B * t), labels]
    rank = (sim > pos[:, None]).sum(1)
    ls, hits, drgb, dflow = K_.lmcl(rgb.detach().to(dev), flow.detach().to(dev), 1.0 / T)
    close(ls / (B * t), loss.reshape(1), 1e-5, 'lmcl loss')
    assert hits.cpu().tolist() == [int((rank == 0).sum()), int((rank < 5).sum())]
    close(drgb, rgb.grad, 1e-4, 'lmcl drgb'); close(dflow, flow.grad, 1e-4, 'lmcl dflow')


def test_ema_sgd(dev):
    from mscl_amd import kernels as K_
    n = 100003
    pk, pq = rnd((n + 1,), 1)[:n].clone(), rnd((n + 1,), 2)[:n].clone()
    m = 0.9953
    pkd = pk.to(dev); pb = torch.empty(n, dtype=torch.bfloat16, device=dev)
    K_.ema_update(pkd, pq.to(dev), pb, m)
    ref = pk * m + pq * (1.0 - m)
    close(pkd, ref, 1e-6, 'ema'); assert torch.equal(pb.cpu(), pkd.cpu().to(torch.bfloat16))
    # clip + SGD, two steps, against torch.optim.SGD + clip_grad_norm_
    p = torch.nn.Parameter(rnd((n,), 3)); opt = torch.optim.SGD([p], lr=0.02, momentum=0.9, weight_decay=1e-4)
    pd = p.detach().clone().to(dev); buf = torch.zeros(n, device=dev); pbf = torch.empty(n, dtype=torch.bfloat16, device=dev)
    for step in range(2):
        g = rnd((n,), 10 + step) * (3.0 if step == 0 else 0.01)     # step 0 clips (norm > 40), step 1 does not
        p.grad = g.clone()
        tn = torch.nn.utils.clip_grad_norm_([p], 40.0, 2.0)
        opt.step()
        ss = torch.zeros(1, device=dev)
        K_.sumsq(g.to(dev), ss)
        close(ss.sqrt(), tn.reshape(1), 1e-5, 'grad norm')
        ss2 = torch.full((1,), 7.0, device=dev)                     # overwritten, and bit-reproducible (replicas must agree)
        K_.sumsq(g.to(dev), ss2)
        assert torch.equal(ss, ss2)
        K_.sgd_step(pd, g.to(dev), buf, pbf, ss, 40.0, 0.02, 0.9, 1e-4, first=(step == 0))
        close(pd, p.detach(), 2e-6, f'sgd step {step}')
    assert torch.equal(pbf.cpu(), pd.cpu().to(torch.bfloat16))


def test_flow_visualize_vs_reference_golden(dev):
    """uv -> colour wheel (ssl_aug.py:87-136).  Byte work: the quantised levels must equal the reference's.  The one
    tolerated difference: the angle goes through atan2f, whose last bit differs between the host's vector library (the
    golden was produced on a CPU) and the device's -- as it does between the reference's own CPU and CUDA runs -- and
    where the interpolated colour sits within ~1e-6 of a quantisation edge that flips floor() by one level.
    Bound: never more than 1 level, on at most 0.5 % of the elements (observed 0.2 % on the edge-case grid)."""
    import os
    import numpy as np
    from mscl_amd import kernels as K_
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'flowvis_g7.npz'))
    for a, b in (('uv', 'levels'), ('uv2', 'levels2')):
        uv = torch.from_numpy(g[a]).to(dev)
        want = torch.from_numpy(g[b]).permute(0, 2, 3, 4, 1).contiguous()          # (B,3,T,H,W) -> (B,T,H,W,3)
        out, lv = K_.flow_visualize(uv, want_levels=True)
        d = (lv.cpu().int() - want.int()).abs()
        assert int(d.max()) <= 1, int(d.max())
        assert float((d > 0).float().mean()) <= 5e-3, float((d > 0).float().mean())
        ref = (lv.float() / 255).to(torch.bfloat16)
        assert torch.equal(out[..., :3], ref) and float(out[..., 3:].abs().max()) == 0.0
        # frame window + horizontal flip of the image for sample 1
        flip = torch.tensor([0, 1], dtype=torch.uint8, device=dev)
        T = uv.shape[2]
        out2, lv2 = K_.flow_visualize(uv, t_off=T // 2, T=T // 2, flip=flip, want_levels=True)
        assert torch.equal(lv2[0], lv[0, T // 2:]) and torch.equal(lv2[1], lv[1, T // 2:].flip(2))


def test_pack_input_flip(dev):
    from mscl_amd import kernels as K_
    x = torch.rand((3, 3, 4, 6, 10), device=dev)
    flip = torch.tensor([1, 0, 1], dtype=torch.uint8, device=dev)
    a = K_.pack_input(x, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225), flip=flip)
    b = K_.pack_input(x, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
    assert torch.equal(a[1], b[1]) and torch.equal(a[0], b[0].flip(2)) and torch.equal(a[2], b[2].flip(2))


def test_flow_fra_visualize_vs_reference_golden(dev):
    """Flow Rotation Augmentation + visualiser fused (transforms_motion.py:103-142 + ssl_aug.py:87-136) against the
    reference pipeline class.  Normalised vectors: float32 results of the reference's mixed float32 / float64 arithmetic,
    compared at 2 ulp (sin / cos / sqrt in double on the device vs the host, then one rounding to float32); colour levels: +-1 level on <= 0.5 % of the elements
    (see test_flow_visualize_vs_reference_golden)."""
    import os
    import numpy as np
    from mscl_amd import kernels as K_
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'fra_g8.npz'))
    uv = torch.from_numpy(g['uv']).permute(0, 4, 1, 2, 3).contiguous().to(dev)            # (B,T,H,W,2) -> (B,2,T,H,W)
    cid = torch.from_numpy(g['cid']).to(torch.int32).to(dev)
    out, lv, nm = K_.flow_fra_visualize(uv, cid, want_debug=True)
    want = torch.from_numpy(g['normed']).float()                                           # what ToTensor hands to the GPU
    err = (nm.cpu() - want).abs()
    assert float(err.max()) <= 2.4e-7, float(err.max())          # values lie in [-1, 1]: two float32 ulps at 1.0
    wl = torch.from_numpy(g['levels']).permute(0, 2, 3, 4, 1).contiguous()               # (B,3,2T,H,W) -> (B,2T,H,W,3)
    d = (lv.cpu().int() - wl.int()).abs()
    assert int(d.max()) <= 1 and float((d > 0).float().mean()) <= 5e-3, (int(d.max()), float((d > 0).float().mean()))
    assert torch.equal(out[..., :3], (lv.float() / 255).to(torch.bfloat16)) and float(out[..., 3:].abs().max()) == 0.0


def test_edge_cases_and_error_codes(dev):
    """Edge cases the reference guards or that sit on kernel boundaries:
      * enqueue exactly up to the end of the queue wraps the pointer to 0 (moco.py:423-440), and a batch that does not
        divide K is refused like the reference's `assert self.K % batch_size == 0` (moco.py:432);
      * a convolution whose position count is not a multiple of any tile (ragged M) and a 1-position map;
      * argument errors come back as negative codes -> MsclError, never as a launch."""
    from mscl_amd import kernels as K_, lib
    from mscl_amd.recognizers import MoCoV2
    Kq, dim, n = 32, 128, 8
    queue = torch.zeros((dim, Kq), device=dev); count = torch.zeros(Kq, dtype=torch.long, device=dev)
    ptr_ = torch.tensor([Kq - n], dtype=torch.long, device=dev)
    keys = torch.nn.functional.normalize(torch.randn(n, dim, device=dev), dim=1)
    K_.queue_enqueue(queue, count, ptr_, keys)
    assert int(ptr_) == 0 and torch.equal(queue[:, Kq - n:], keys.T)
    assert torch.equal(count.cpu(), torch.cat([torch.ones(Kq - n, dtype=torch.long), torch.ones(n, dtype=torch.long)]))
    rec = MoCoV2(backbone=dict(type='resnet_flow.r2d_18'), neck=dict(type='BaseMoCo'), moco_head=dict(type='MoCoHead', basename='_flow',
                 loss_cls=dict(type='CrossEntropyLoss_torch', ignore_index=-1)), dim_in=128, K=20, max_iters=10, mlp=True)
    rec.queue, rec.count, rec.queue_ptr = torch.zeros((128, 20), device=dev), torch.zeros(20, dtype=torch.long, device=dev), \
        torch.zeros(1, dtype=torch.long, device=dev)
    with pytest.raises(AssertionError):
        rec.dequeue_and_enqueue(keys)                        # 20 % 8 != 0
    # ragged / tiny convolutions against the fp32 reference
    for shape, Kc in (((1, 1, 5, 7, 64), 64), ((1, 1, 1, 1, 64), 64), ((3, 2, 9, 5, 16), 32)):
        x = bf(rnd(shape, 21)); w = bf(rnd((Kc, 3, 3, 3, shape[-1]), 22, scale=0.05))
        d = K_.conv_desc(shape, Kc, (3, 3, 3), (1, 1, 1), (1, 1, 1))
        y = K_.conv3d_fwd(x.to(dev), w.to(dev), d)
        close(y, _conv_ref(x.float(), w.float(), (1, 1, 1), (1, 1, 1)), BF16_TOL, f'ragged conv {shape}')
    # error paths
    with pytest.raises(lib.MsclError):
        K_.conv3d_fwd(torch.zeros((1, 1, 4, 4, 12), dtype=torch.bfloat16, device=dev),
                      torch.zeros((8, 3, 3, 3, 12), dtype=torch.bfloat16, device=dev),
                      K_.conv_desc((1, 1, 4, 4, 12), 8, (3, 3, 3), (1, 1, 1), (1, 1, 1)))          # C % 8 != 0
    with pytest.raises(lib.MsclError):
        K_.pack_input(torch.zeros((1, 3, 4, 4, 4)), None, None)                                     # CPU tensor
    with pytest.raises(lib.MsclError):
        lib.call('mscl_sumsq', None, None, 0, None, 0, None)


def test_color_aug_and_blur_match_oracle(dev):
    """csrc/color_aug.hip vs oracle/coloraug.py (the restated kornia arithmetic of ssl_aug_v2.py:31-43): every jitter op
    alone and chained in several orders, grayscale, blur at small / large sigma, samples left untouched."""
    from mscl_amd import kernels as K
    from oracle import coloraug
    g = torch.Generator().manual_seed(17)
    B, T, H, W = 8, 3, 20, 27
    x = torch.rand((B, 3, T, H, W), generator=g)
    x[0, :, 0, :4, :4] = 0.5                          # grey pixels: zero saturation, delta == 0 branch
    x[1, 0, 0, :4, :4] = x[1, 1, 0, :4, :4]           # ties between channel maxima
    P = torch.zeros(B, K.AUG_PARAMS)
    orders = [[0, 1, 2, 3], [3, 2, 1, 0], [2, 0, 3, 1], [1, 3, 0, 2], [0, 1, 2, 3], [3, 0, 1, 2], [2, 3, 0, 1], [0, 2, 1, 3]]
    P[:, 1:5] = torch.tensor(orders, dtype=torch.float32)
    P[:, 0] = torch.tensor([1, 1, 1, 1, 0, 1, 1, 1.])
    P[:, 5] = torch.tensor([0.6, 1.4, 1.0, 1.1, 1.3, 0.9, 0.7, 1.2])
    P[:, 6] = torch.tensor([1.4, 0.6, 1.0, 0.8, 1.3, 1.1, 1.2, 0.9])
    P[:, 7] = torch.tensor([0.6, 1.4, 1.3, 1.0, 0.7, 0.8, 1.1, 1.2])
    P[:, 8] = torch.tensor([-0.1, 0.1, 0.05, -0.03, 0.1, 0.0, 0.08, -0.07]) * 6.283185307179586
    P[:, 9] = torch.tensor([0, 0, 1, 0, 0, 1, 0, 0.])
    P[:, 10] = torch.tensor([0, 0.1, 2.0, 0, 0.7, 0, 1.3, 0])
    for ks in (0, 11, 5):
        got = K.color_aug(x.to(dev), P.to(dev), ks).cpu()
        ref = coloraug.color_aug(x, P, ks)
        assert (got - ref).abs().max().item() < 3e-5, ks
    untouched = K.color_aug(x.to(dev), torch.zeros(B, K.AUG_PARAMS, device=dev), 11).cpu()
    assert torch.equal(untouched, x)
    # argument checks: blur wider than the frame allows, even tap count, aliased scratch
    from mscl_amd import lib
    xs = torch.rand((1, 3, 1, 4, 4), device=dev)
    with pytest.raises(lib.MsclError):
        K.color_aug(xs, torch.zeros(1, K.AUG_PARAMS, device=dev), 11)
    with pytest.raises(lib.MsclError):
        K.color_aug(xs, torch.zeros(1, K.AUG_PARAMS, device=dev), 4)
    with pytest.raises(lib.MsclError):
        K.color_aug(xs, torch.zeros(2, K.AUG_PARAMS, device=dev), 0)
    assert K.color_aug(torch.empty((0, 3, 2, 8, 8), device=dev), torch.zeros(0, K.AUG_PARAMS, device=dev), 3).shape[0] == 0


def test_maxpool_hw(dev):
    """(1,3,3) / (1,2,2) / (0,1,1) max-pool, forward and gather backward, against torch on the same bf16 values -- bit for bit,
    including the tie rule (first tap in scan order): a ReLU output with whole windows at zero."""
    from mscl_amd import kernels as K_
    for shape in ((2, 3, 14, 14, 64), (1, 2, 9, 11, 8), (2, 2, 7, 7, 16)):
        x = F.relu(bf(rnd(shape, 11)).float() - 0.5)          # ~70 % exact zeros
        x = bf(x)
        out, win = K_.maxpool_hw_fwd(x.to(dev))
        xr = x.float().permute(0, 4, 1, 2, 3).requires_grad_(True)
        yr = F.max_pool3d(xr, (1, 3, 3), (1, 2, 2), (0, 1, 1))
        assert torch.equal(out.float().cpu(), yr.detach().permute(0, 2, 3, 4, 1))
        dy = bf(rnd(tuple(out.shape), 12))
        yr.backward(dy.float().permute(0, 4, 1, 2, 3))
        dx = K_.maxpool_hw_bwd(dy.to(dev), win, tuple(x.shape))
        close(dx, xr.grad.permute(0, 2, 3, 4, 1), BF16_TOL, 'maxpool bwd')    # sums of up to 4 bf16 gradients, rounded once


def test_statistics_groups_equal_separate_calls(dev):
    """mscl_conv3d_fwd_groups / mscl_bn_act_fwd_groups / _bwd_groups on a batch of two calls' worth of samples == the two calls made
    one after the other (per-call batch statistics, running statistics updated in call order, dgamma / dbeta summed)."""
    import ctypes
    from mscl_amd import kernels as K_, lib
    N, T, H, W, C, Kc = 4, 3, 10, 10, 16, 32
    x = bf(rnd((N, T, H, W, C), 21)); w = bf(rnd((Kc, 1, 3, 3, C), 22, scale=0.1))
    xg, wg = x.to(dev), w.to(dev)
    gamma, beta = (rnd((Kc,), 23) * 0.2 + 1).to(dev), (rnd((Kc,), 24) * 0.1).to(dev)
    dout = bf(rnd((N, T, H, W, Kc), 25)).to(dev)

    def bnp(stats, rm, rv, nbt, save, groups):
        sp, sv = stats.data_ptr(), save.data_ptr()
        return lib.BnParams(sp, sp + 4 * Kc, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(),
                            sv, sv + 4 * groups * Kc)

    def run(xs, groups, rm, rv, nbt, dgamma, dbeta, dos):
        d = K_.conv_desc(tuple(xs.shape), Kc, (1, 3, 3), (1, 1, 1), (0, 1, 1))
        stats = torch.zeros((groups, K_.STAT_SLOTS, 2, Kc), device=dev)
        y = torch.empty(K_.out_shape(d), dtype=torch.bfloat16, device=dev)
        out = torch.empty_like(y)
        save = torch.empty((2, groups, Kc), device=dev)
        lib.call('mscl_conv3d_fwd_groups', ctypes.byref(d), xs.data_ptr(), wg.data_ptr(), y.data_ptr(), None, None, 0,
                 stats.data_ptr(), stats.data_ptr() + 4 * Kc, groups, None, 0, lib.stream_ptr())
        bp = bnp(stats, rm, rv, nbt, save, groups)
        rows = y.numel() // Kc
        lib.call('mscl_bn_act_fwd_groups', y.data_ptr(), ctypes.byref(bp), None, None, out.data_ptr(), rows, Kc, 1e-5, 0.1, 1, groups,
                 lib.stream_ptr())
        scratch = torch.zeros((groups * K_.STAT_SLOTS * 4 * Kc,), device=dev)
        dy, _ = K_.bn_act_bwd(dos, out, y, gamma, save[0], save[1], dgamma, dbeta, 1, scratch, groups=groups)
        return out, dy
    mk = lambda: (torch.zeros(Kc, device=dev), torch.ones(Kc, device=dev), torch.zeros((), dtype=torch.long, device=dev),
                  torch.zeros(Kc, device=dev), torch.zeros(Kc, device=dev))
    rm2, rv2, nb2, dg2, db2 = mk()
    out2, dy2 = run(xg, 2, rm2, rv2, nb2, dg2, db2, dout)
    rm1, rv1, nb1, dg1, db1 = mk()
    oa, da = run(xg[:2].contiguous(), 1, rm1, rv1, nb1, dg1, db1, dout[:2].contiguous())
    ob, dbb = run(xg[2:].contiguous(), 1, rm1, rv1, nb1, dg1, db1, dout[2:].contiguous())
    close(out2, torch.cat([oa, ob]), 2.0 ** -8, 'grouped BN forward')
    close(dy2, torch.cat([da, dbb]), 2.0 ** -7, 'grouped BN backward')
    assert int(nb2) == int(nb1) == 2
    close(rm2, rm1, 1e-5, 'running mean'); close(rv2, rv1, 1e-5, 'running var')
    close(dg2, dg1, 1e-4, 'dgamma'); close(db2, db1, 1e-4, 'dbeta')


def test_bn_wide_maps_and_fixed_order_statistics(dev):
    """BatchNorm backward on maps of 1024 / 2048 channels (512-channel chunks) against autograd, and mscl_bn_stats (the
    deterministic mode's statistics pass) against fp32 sums, twice: bit-identical."""
    from mscl_amd import kernels as K_, lib
    for C in (1024, 2048):
        rows = 96
        y = bf(rnd((rows, C), 31)); dout = bf(rnd((rows, C), 32))
        gamma = (rnd((C,), 33) * 0.2 + 1)
        yr = y.float().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = torch.zeros(C, requires_grad=True)
        o = F.relu(F.batch_norm(yr, None, None, gr, br, training=True, eps=1e-5))
        o.backward(dout.float())
        mean = y.float().mean(0); inv = 1.0 / torch.sqrt(y.float().var(0, unbiased=False) + 1e-5)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        scratch = torch.zeros((K_.STAT_SLOTS * 4 * C,), device=dev)
        dy, _ = K_.bn_act_bwd(dout.to(dev), bf(o.detach()).to(dev), y.to(dev), gamma.to(dev), mean.to(dev), inv.to(dev), dg, db, 1, scratch)
        close(dy, yr.grad, 2.0 ** -6, f'bn bwd C={C}')
        close(dg, gr.grad, 2e-3, 'dgamma'); close(db, br.grad, 2e-3, 'dbeta')
        yd = y.to(dev)
        for parts_n in (0, lib.call_raw('mscl_det_parts_floats', rows, C, 1, 2)):     # partials in the slots / in caller scratch
            st = [torch.zeros((K_.STAT_SLOTS, 2, C), device=dev) for _ in range(2)]
            parts = torch.empty((parts_n,), device=dev) if parts_n else None
            for t in st:
                lib.call('mscl_bn_stats', yd.data_ptr(), t.data_ptr(), t.data_ptr() + 4 * C, rows, C, 1, lib.ptr(parts), parts_n, lib.stream_ptr())
            assert torch.equal(st[0], st[1])
            assert not st[0][1:].any(), 'the sums belong in slot 0, the other slots stay zero'
            close(st[0][0, 0], y.float().sum(0), 1e-5, 'stats sum'); close(st[0][0, 1], (y.float() ** 2).sum(0), 1e-5, 'stats sumsq')


def test_deterministic_bn_sums_two_levels(dev):
    """deterministic mode's BatchNorm sums on a map large enough for many partial blocks (layer-2 size) and with two statistics
    groups: statistics pass and backward with caller scratch for the partials (mscl_det_parts_floats) and without it (the slots hold
    them) agree with fp32 sums / the atomic path, and repeat bit for bit."""
    from mscl_amd import kernels as K_, lib
    C, rows = 128, 2 * 6272
    y = bf(rnd((rows, C), 51)); dout = bf(rnd((rows, C), 52)); gamma = rnd((C,), 53) * 0.2 + 1; beta = rnd((C,), 54) * 0.1
    yd, dd = y.to(dev), dout.to(dev)
    for G in (1, 2):
        yg = y.float().view(G, rows // G, C)
        need = lib.call_raw('mscl_det_parts_floats', rows, C, G, 2)
        assert need == G * min(128, -(-(rows // G) // (256 // (C // 8) * 8))) * 2 * C
        got = []
        for parts_n in (0, need, need):
            st = torch.zeros((G, K_.STAT_SLOTS, 2, C), device=dev)
            parts = torch.full((parts_n,), float('nan'), device=dev) if parts_n else None
            lib.call('mscl_bn_stats', yd.data_ptr(), st.data_ptr(), st.data_ptr() + 4 * C, rows, C, G, lib.ptr(parts), parts_n, lib.stream_ptr())
            assert not st[:, 1:].any()
            close(st[:, 0, 0], yg.sum(1), 1e-5, f'sum G={G}'); close(st[:, 0, 1], (yg ** 2).sum(1), 1e-5, f'sumsq G={G}')
            got.append(st)
        assert torch.equal(got[1], got[2])
        # backward: atomic path as the yardstick, then deterministic mode with and without scratch
        mean = yg.mean(1); inv = 1.0 / torch.sqrt(yg.var(1, unbiased=False) + 1e-5)
        out = bf(F.relu((yg - mean[:, None]) * inv[:, None] * gamma + beta)).view(rows, C).to(dev)
        sm = (mean if G > 1 else mean[0]).contiguous().to(dev); si = (inv if G > 1 else inv[0]).contiguous().to(dev)

        def run():
            dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            scr = torch.zeros((G * K_.STAT_SLOTS * 4 * C,), device=dev)
            dy, _ = K_.bn_act_bwd(dd, out, yd, gamma.to(dev), sm, si, dg, db, 1, scr, groups=G)
            return dy, dg, db
        ref = run()
        lib.set_deterministic(True)
        try:
            a, b = run(), run()
            lib.DET = False                  # no scratch for the partials: the slots hold them
            c = run()
        finally:
            lib.set_deterministic(False)
        for u, v in zip(a, b):
            assert torch.equal(u, v)
        for r in (a, c):
            close(r[0], ref[0], 2.0 ** -6, f'dy G={G}'); close(r[1], ref[1], 1e-4, 'dgamma'); close(r[2], ref[2], 1e-4, 'dbeta')


def test_linear_more_than_32_rows(dev):
    """the LMCL flow transform of mscl_r50 and the batched flow projection head: 64 rows = two launches of the 32-row kernels"""
    from mscl_amd import kernels as K_
    rows, i, o = 64, 256, 128
    x = rnd((rows, i), 41).requires_grad_(True); w = (rnd((o, i), 42) / i ** 0.5).requires_grad_(True); b = rnd((o,), 43).requires_grad_(True)
    y = F.linear(x, w, b)
    yd = K_.linear_fwd(x.detach().to(dev), w.detach().to(dev), b.detach().to(dev), False)
    close(yd, y, F32_TOL, 'linear fwd 64 rows')
    dy = rnd((rows, o), 44); y.backward(dy)
    dw = torch.zeros(o, i, device=dev); db = torch.zeros(o, device=dev)
    dx = K_.linear_bwd(x.detach().to(dev), w.detach().to(dev), yd, dy.to(dev), dw, db, False)
    close(dx, x.grad, F32_TOL, 'dx'); close(dw, w.grad, F32_TOL, 'dw'); close(db, b.grad, F32_TOL, 'db')


def test_nce_virtual_enqueue_equals_real_enqueue(dev):
    """mscl_nce_fwd_virt / _bwd_virt: the InfoNCE pass on the snapshot AFTER an enqueue, read from the buffers BEFORE it, against the
    same pass after the real mscl_queue_enqueue -- bit for bit (same arithmetic on the same values), at a pointer in the middle
    of the queue and at its last slot group."""
    from mscl_amd import kernels as K_, lib
    dim, Kq, R, n = 128, 4096, 24, 8
    lib.set_deterministic(True)            # one add per dq element: the two runs are comparable bit for bit
    try:
        _virt_vs_real(K_, dev, dim, Kq, R, n)
    finally:
        lib.set_deterministic(False)


def _virt_vs_real(K_, dev, dim, Kq, R, n):
    for p0 in (1024, Kq - n):
        queue = F.normalize(rnd((dim, Kq), 51), dim=0).to(dev)
        count = torch.randint(0, 5000, (Kq,), generator=torch.Generator().manual_seed(52)).to(dev)
        qptr = torch.tensor([p0], dtype=torch.long, device=dev)
        q = F.normalize(rnd((R, dim), 53), dim=1).to(dev); kpos = F.normalize(rnd((R, dim), 54), dim=1).to(dev)
        keys = F.normalize(rnd((n, dim), 55), dim=1).to(dev)
        scale = torch.full((R,), 1.0 / R, device=dev)
        pos = K_.rowdot(q, kpos)
        lse_v, loss_v, rank_v = K_.nce_forward(queue, count, q, pos, 1 / 0.07, virt=(keys, qptr))
        dq_v = K_.nce_backward(queue, count, q, lse_v, scale, 1 / 0.07, virt=(keys, qptr))
        K_.queue_enqueue(queue, count, qptr, keys)
        assert int(qptr) == (p0 + n) % Kq
        lse_r, loss_r, rank_r = K_.nce_forward(queue, count, q, pos, 1 / 0.07)
        dq_r = K_.nce_backward(queue, count, q, lse_r, scale, 1 / 0.07)
        assert torch.equal(lse_v, lse_r) and torch.equal(loss_v, loss_r) and torch.equal(rank_v, rank_r)
        assert torch.equal(dq_v, dq_r)


def test_conv_wgrad_group(dev):
    """mscl_conv3d_wgrad_group (round 5): the weight and bias gradients of several small layers in one launch each -- 3x3x3, 1x3x3,
    1x1x1 and strided layers of different maps, one module applied to two maps (shared dw: atomics), accumulation into non-zero
    buffers -- against CPU fp32 autograd and against the per-layer entry point; window-resident / big-tile shapes are refused."""
    import ctypes
    from mscl_amd import kernels as K_, lib
    specs = [  # N,T,H,W, C, K, kernel, stride, pad, shares dw with
        (2, 2, 7, 7, 512, 512, (3, 3, 3), (1, 1, 1), (1, 1, 1), None),
        (2, 4, 14, 14, 256, 512, (3, 3, 3), (2, 2, 2), (1, 1, 1), None),
        (2, 4, 14, 14, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), None),
        (2, 2, 7, 7, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), 2),        # the same module on a second pyramid level
        (2, 4, 14, 14, 256, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), None),
        (2, 4, 14, 14, 128, 256, (1, 1, 1), (2, 2, 2), (0, 0, 0), None),
        (1, 2, 7, 7, 128, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1), None),
        (2, 2, 7, 7, 256, 1024, (1, 1, 1), (1, 1, 1), (0, 0, 0), None),    # a biased layer wider than 512: column sums in two chunks
    ]
    n = len(specs)
    descs = (lib.ConvDesc * n)()
    xs, dys, dws, dbs, refs, keep = [], [], [], [], [], []
    for i, (N, T, H, W, C, Kc, kern, st, pad, share) in enumerate(specs):
        x = bf(rnd((N, T, H, W, C), 100 + i)); w = rnd((Kc, *kern, C), 200 + i).requires_grad_(True)
        d = K_.conv_desc(x.shape, Kc, kern, st, pad)
        assert lib.call_raw('mscl_conv3d_wgrad_groupable', ctypes.byref(d)) == 1, specs[i]
        y = _conv_ref(x.float(), w, st, pad)
        dy = bf(rnd(tuple(y.shape), 300 + i))
        y.backward(dy.float())
        ctypes.memmove(ctypes.byref(descs[i]), ctypes.byref(d), ctypes.sizeof(lib.ConvDesc))
        xs.append(x.to(dev)); dys.append(dy.to(dev))
        if share is None:
            dws.append(torch.full((Kc, *kern, C), 0.5, device=dev)); dbs.append(torch.full((Kc,), -1.0, device=dev))
            refs.append([w.grad.clone(), dy.float().sum(dim=(0, 1, 2, 3))])
        else:
            dws.append(dws[share]); dbs.append(dbs[share])
            refs[share][0] += w.grad; refs[share][1] += dy.float().sum(dim=(0, 1, 2, 3))
            refs.append(None)
    arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    n0 = lib.call_raw('mscl_debug_wgrad_group_launches')
    lib.call('mscl_conv3d_wgrad_group', n, descs, arr(xs), arr(dys), arr(dws), arr(dbs), lib.stream_ptr())
    assert lib.call_raw('mscl_debug_wgrad_group_launches') == n0 + 1
    for i, r in enumerate(refs):
        if r is not None:
            close(dws[i] - 0.5, r[0], F32_TOL, f'grouped wgrad {i}')
            close(dbs[i] + 1.0, r[1], F32_TOL, f'grouped dbias {i}')
    # the per-layer entry point on the same operands (fp32 sums in another order)
    for i, (N, T, H, W, C, Kc, kern, st, pad, share) in enumerate(specs):
        if share is None and i != 2:
            dw1 = torch.zeros((Kc, *kern, C), device=dev)
            K_.conv3d_wgrad(xs[i], dys[i], K_.conv_desc(xs[i].shape, Kc, kern, st, pad), dw1)
            close(dws[i] - 0.5, dw1, F32_TOL, f'grouped vs single {i}')
    # shapes the window-resident kernels / the 128 x 128 tile take are not groupable, and the call refuses them
    for shp in [((8, 16, 56, 56, 64), 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)), ((8, 8, 28, 28, 128), 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
                ((8, 16, 56, 56, 64), 128, (3, 3, 3), (2, 2, 2), (1, 1, 1)), ((2, 4, 14, 14, 16), 16, (1, 3, 3), (1, 1, 1), (0, 1, 1))]:
        d = K_.conv_desc(*shp)
        assert lib.call_raw('mscl_conv3d_wgrad_groupable', ctypes.byref(d)) == 0, shp
    d1 = (lib.ConvDesc * 1)()
    ctypes.memmove(ctypes.byref(d1[0]), ctypes.byref(K_.conv_desc((8, 16, 56, 56, 64), 64, (3, 3, 3), (1, 1, 1), (1, 1, 1))), ctypes.sizeof(lib.ConvDesc))
    one = (ctypes.c_void_p * 1)(xs[0].data_ptr())
    assert lib.call_raw('mscl_conv3d_wgrad_group', 1, d1, one, one, one, None, lib.stream_ptr()) == -2


def test_upsample_beside_conv_streams(dev):
    """The stand-alone reproducer of round 5's "dropped corner" (tools/diag/flake_repro.hip: conv -> trilinear up-sampling on one
    stream, bit-compared with its first result, while the key and flow trunks' convs run on two more streams in one captured HIP
    graph; HIP runtime + C ABI only) against the PRODUCT library: 60 000 comparisons, none may differ.  With packed fp32
    instructions in the up-sampling kernel the same run shows ~230 differing comparisons (profiles/r06_flake.md: a `v_pk_mul_f32`
    with a cross-half op_sel returns zero in lanes 48-63 beside MFMA kernels of another hardware queue; three boxes); the library
    is built without those instructions (csrc/build.sh checks its code objects).
    Informational second run: the VALU probe of the same tool in the compiled kernel's instruction form -- it shows whether THIS
    box exhibits the hardware behaviour at all (printed, not asserted)."""
    import subprocess
    from mscl_amd import lib
    exe = os.path.join(ROOT, 'tools', 'diag', 'flake_repro')
    if not os.path.exists(exe):                   # __graft_entry__.build() builds it; a tree that arrived without built files builds it here
        subprocess.run(['bash', os.path.join(ROOT, 'tools', 'diag', 'build.sh')], check=True, capture_output=True, timeout=600)
    assert os.path.exists(exe), 'tools/diag/flake_repro is missing and tools/diag/build.sh did not produce it'
    r = subprocess.run([exe, '--lib', lib.LIB_PATH, '--replays', '3000'], capture_output=True, text=True, timeout=300)
    tail = [ln for ln in r.stdout.splitlines() if ln.startswith('SUMMARY') or 'reference launch' in ln]
    print('\n'.join(tail))
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert any('comparisons 60000 differing 0 ' in ln for ln in tail), tail
    p = subprocess.run([exe, '--lib', lib.LIB_PATH, '--replays', '2000', '--probe', '0', '--side', 'convs', '--streams', 'B'],
                       capture_output=True, text=True, timeout=300)
    print('\n'.join(ln for ln in p.stdout.splitlines() if ln.startswith('PROBE')))


# conv_dgrad_s2.hip (round 6): the window-resident input gradient of a 3x3x3 / stride-2 / pad-1 entry conv, forced onto small maps
# (MSCL_DGRAD_S2=2; by default only maps that give nearly every CU a 256-cell tile take it): one tile per plane with a ragged end,
# odd extents on every axis (the last odd output of an axis does not exist; T' + 1 runs past the clip), two output-channel tiles and
# four dy-channel chunks, several tiles per plane; launch counter, addend, and a repeat-launch race screen on the DMA rings.
S2_CASES = [
    # name, N,T,H,W, C (dx), K (dy)
    ('s2_64_128', 2, 8, 24, 24, 64, 128),
    ('s2_odd', 1, 7, 23, 21, 64, 128),
    ('s2_128_256', 1, 4, 20, 20, 128, 256),
    ('s2_tiles', 1, 4, 44, 40, 64, 128),               # 22 x 21 padded cells = 462: two tiles per plane, the second ragged
]


@pytest.mark.parametrize('case', S2_CASES, ids=[c[0] for c in S2_CASES])
def test_conv_dgrad_s2(case, dev):
    from mscl_amd import kernels as K_, lib
    name, N, T, H, W, C, K = case
    kern, st, pad = (3, 3, 3), (2, 2, 2), (1, 1, 1)
    x = rnd((N, T, H, W, C), 31).requires_grad_(True); w = bf(rnd((K, *kern, C), 32, scale=(2.0 / (C * 27)) ** 0.5))
    d = K_.conv_desc((N, T, H, W, C), K, kern, st, pad)
    yr = _conv_ref(x, w.float(), st, pad)
    dy = bf(rnd(tuple(yr.shape), 33))
    yr.backward(dy.float())
    wT = torch.empty((C, *kern, K), dtype=torch.bfloat16, device=dev)
    K_.weight_transpose(w.to(dev), wT, K, 27, C)
    os.environ['MSCL_DGRAD_S2'] = '2'
    lib.call_raw('mscl_tuning_reload')
    try:
        n0 = lib.call_raw('mscl_debug_dgrad_s2_launches')
        dx = K_.conv3d_dgrad(dy.to(dev), wT, d)
        assert lib.call_raw('mscl_debug_dgrad_s2_launches') == n0 + 1, 'the window-resident kernel did not take the launch'
        close(dx, x.grad, BF16_TOL, 's2 dgrad')
        add = bf(rnd((N, T, H, W, C), 34))
        close(K_.conv3d_dgrad(dy.to(dev), wT, d, addend=add.to(dev)), x.grad + add.float(), BF16_TOL, 's2 dgrad + addend')
        for _ in range(10):
            assert torch.equal(K_.conv3d_dgrad(dy.to(dev), wT, d), dx)
        # and the same arithmetic as the implicit-GEMM parity classes up to the order of the fp32 sums
        os.environ['MSCL_DGRAD_S2'] = '0'
        lib.call_raw('mscl_tuning_reload')
        close(K_.conv3d_dgrad(dy.to(dev), wT, d), dx, BF16_TOL, 's2 vs parity classes')
    finally:
        os.environ.pop('MSCL_DGRAD_S2', None)
        lib.call_raw('mscl_tuning_reload')


# In-place input gradient (round 6): dx += conv_transpose(dy, w) with addend == dx.  For a strided 1x1x1 shortcut the positions no
# tap reaches are not touched at all (bit-identical afterwards); the positions that are reached hold base + gradient.  Also a 3x3x3
# stride-2 conv in place (every position is reached: the plain sum).
INPLACE_CASES = [
    # name, N,T,H,W, C (dx), K (dy), kernel, stride, pad
    ('sc_64_128_s222', 2, 8, 24, 24, 64, 128, (1, 1, 1), (2, 2, 2), (0, 0, 0)),
    ('sc_odd_s222', 1, 5, 7, 9, 64, 128, (1, 1, 1), (2, 2, 2), (0, 0, 0)),
    ('sc_256_512_s122', 1, 4, 14, 14, 256, 512, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
    ('sc_flow_16_32_s122', 2, 4, 28, 28, 16, 32, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
    ('k333_s222', 1, 4, 12, 12, 64, 128, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    # 1024 dy channels on a small map: 32 K steps and few tiles, the shape on which split-K would engage -- it must not once classes are
    # dropped (its finalize pass walks every position of dx)
    ('sc_512_1024_s122_small', 1, 2, 8, 8, 512, 1024, (1, 1, 1), (1, 2, 2), (0, 0, 0)),
    ('sc_512_2048_s222_small', 1, 2, 6, 6, 512, 2048, (1, 1, 1), (2, 2, 2), (0, 0, 0)),
]


@pytest.mark.parametrize('case', INPLACE_CASES, ids=[c[0] for c in INPLACE_CASES])
def test_conv_dgrad_in_place(case, dev):
    from mscl_amd import kernels as K_
    name, N, T, H, W, C, K, kern, st, pad = case
    taps = kern[0] * kern[1] * kern[2]
    x = rnd((N, T, H, W, C), 41).requires_grad_(True); w = bf(rnd((K, *kern, C), 42, scale=(2.0 / (C * taps)) ** 0.5))
    d = K_.conv_desc((N, T, H, W, C), K, kern, st, pad)
    yr = _conv_ref(x, w.float(), st, pad)
    dy = bf(rnd(tuple(yr.shape), 43))
    yr.backward(dy.float())
    wT = torch.empty((C, *kern, K), dtype=torch.bfloat16, device=dev)
    K_.weight_transpose(w.to(dev), wT, K, taps, C)
    base = bf(rnd((N, T, H, W, C), 44))
    out = base.to(dev).clone()
    r = K_.conv3d_dgrad(dy.to(dev), wT, d, out=out)
    assert r.data_ptr() == out.data_ptr()
    close(out, x.grad + base.float(), BF16_TOL, 'in-place dgrad')
    # and the same as the two-map form
    close(out, K_.conv3d_dgrad(dy.to(dev), wT, d, addend=base.to(dev)), BF16_TOL, 'in place vs addend')
    if kern == (1, 1, 1):
        reached = torch.zeros((T, H, W), dtype=torch.bool)
        reached[::st[0], ::st[1], ::st[2]] = True
        assert torch.equal(out.cpu()[:, ~reached], base[:, ~reached]), 'a position no tap reaches was rewritten'
        assert not torch.equal(out.cpu()[:, reached], base[:, reached])


@pytest.mark.parametrize('outer,inner,C', [(8, 98, 512), (8, 6272, 128), (3, 1, 2048), (2, 2049, 64), (2, 130, 16), (1, 1025, 256)])
def test_pool_fwd_shapes(outer, inner, C, dev):
    """mean over the middle axis on the shapes of the step (layer-4 map, the 28 x 28 pyramid level), the Bottleneck trunks' 2048
    channels, a narrow map, and row counts one past a trip of the 1024-thread layout; twice (a fixed summation order)"""
    from mscl_amd import kernels as K_
    x = bf(rnd((outer, inner, C), 51))
    p = K_.pool_fwd(x.to(dev), outer, inner)
    close(p, x.double().mean(1).float(), 2e-6 * max(1.0, inner ** 0.5), 'pool fwd')
    assert torch.equal(K_.pool_fwd(x.to(dev), outer, inner), p)


# conv_wgrad_stem.hip (round 6): the window-resident weight gradient of the W-paired RGB stems on the same cases as the forward kernel:
# launch counter, CPU fp32 autograd of F.conv3d on the 3-channel clip (through the fold of the paired staging buffer), accumulation
# into a non-zero dw, the same bits on every run (slot-ordered slab sums), and the general kernel's result.
@pytest.mark.parametrize('case', STEM_CASES, ids=[c[0] for c in STEM_CASES])
def test_conv_wgrad_stem_window_resident(case, dev):
    from mscl_amd import kernels as K_
    from mscl_amd import lib
    name, N, T, H, W, kT, sT, pT, forced = case
    Co = 64
    x3 = bf(rnd((N, T, H, W, 3), 11)); w = bf(rnd((Co, kT, 7, 7, 3), 12, scale=(2.0 / (3 * 49 * kT)) ** 0.5))
    x8 = torch.zeros((N, T, H, W, 8), dtype=torch.bfloat16); x8[..., :3] = x3
    wr = w.float().requires_grad_(True)
    yr = _conv_ref(x3.float(), wr, (sT, 2, 2), (pT, 3, 3))
    dy = bf(rnd(tuple(yr.shape), 13))
    yr.backward(dy.float())
    xp = K_.pair_w(x8.to(dev))
    d = K_.conv_desc(tuple(xp.shape), Co, (kT, 7, 4), (sT, 2, 1), (pT, 3, 1))
    if forced:
        lib.tune(MSCL_WGRAD_STEM=1)
    try:
        n0 = lib.call_raw('mscl_debug_wgrad_stem_launches')
        dw8 = torch.zeros((Co, kT, 7, 4, 8), dtype=torch.float32, device=dev)
        K_.conv3d_wgrad(xp, dy.to(dev), d, dw8, None)
        assert lib.call_raw('mscl_debug_wgrad_stem_launches') == n0 + 1, 'the window-resident stem weight gradient did not take the launch'
        g = torch.zeros((Co, kT, 7, 7, 3), dtype=torch.float32, device=dev)
        K_.pair_w_grad_fold(dw8, g)
        close(g, wr.grad, F32_TOL, f'stem wgrad {name}')
        assert float(dw8[..., 6:].abs().max()) == 0.0          # channels 6, 7 of a pair are zero in the clip
        first = dw8.clone()
        K_.conv3d_wgrad(xp, dy.to(dev), d, dw8, None)            # accumulates
        assert torch.equal(dw8, 2 * first)
        for _ in range(3):
            again = torch.zeros_like(dw8)
            K_.conv3d_wgrad(xp, dy.to(dev), d, again, None)
            assert torch.equal(again, first)
        lib.tune(MSCL_WGRAD_STEM=0)
        ref = torch.zeros_like(dw8)
        K_.conv3d_wgrad(xp, dy.to(dev), d, ref, None)
        assert lib.call_raw('mscl_debug_wgrad_stem_launches') == n0 + 5
        close(first, ref, F32_TOL, 'window-resident stem weight gradient vs general kernel')
    finally:
        lib.tune(MSCL_WGRAD_STEM=None)
