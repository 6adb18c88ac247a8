"""Tensor-level wrappers over the C ABI: allocate outputs with torch, pass raw device pointers.

Activations: torch.bfloat16, shape (N, T, H, W, C) contiguous (NDHWC).  Conv kernels: bf16
[Cout][kT][kH][kW][Cin] (forward) and [Cin][kT][kH][kW][Cout] (input gradient).  No torch compute ops
are used on the data path; torch only owns the memory and the stream.
"""
import ctypes

import torch

from . import lib
from .lib import BnParams, ConvDesc, call, ptr, stream_ptr


def _triple(v):
    return (v, v, v) if isinstance(v, int) else tuple(v)


def conv_desc(x_shape, K, kernel, stride, pad):
    N, T, H, W, C = x_shape
    k, s, p = _triple(kernel), _triple(stride), _triple(pad)
    To = (T + 2 * p[0] - k[0]) // s[0] + 1
    Ho = (H + 2 * p[1] - k[1]) // s[1] + 1
    Wo = (W + 2 * p[2] - k[2]) // s[2] + 1
    return ConvDesc(N, T, H, W, C, To, Ho, Wo, K, k[0], k[1], k[2], s[0], s[1], s[2], p[0], p[1], p[2])


def out_shape(d):
    return (d.N, d.To, d.Ho, d.Wo, d.K)


def _splitk_floats(rows, chans, cap=16):
    """floats of fp32 scratch for split-K when the layer has too few position tiles to fill 256 CUs (else 0).
    cap: the most slabs the launch may use -- the library splits K over as many blocks as the scratch holds copies of the output
    (csrc/conv_pp.hip, conv_igemm.hip), so the caller's scratch size IS the split policy: 16 = latency-optimal (the RGB query
    chain, the step's critical chain), less for the chains that run beside it and only add to the sum of kernel time
    (nn.Conv3dHip.split_cap, recognizers.MSCLWithAug.set_side_split); 1 = no split."""
    tiles = ((rows + 127) // 128) * ((chans + 127) // 128 if chans >= 128 else 1)
    if tiles > 256 or chans < 64 or cap <= 1:
        return 0
    return min(cap, 16, (512 + tiles - 1) // tiles) * rows * chans


def _splitk_ws(rows, chans, device, cap=16):
    n = _splitk_floats(rows, chans, cap)
    return torch.empty((n,), dtype=torch.float32, device=device) if n else None


def fwd_ws_floats(d, stat_groups=0, cap=16):
    """floats of fp32 scratch a forward conv wants: split-K slabs, and in deterministic mode the partials of the statistics pass
    (stat_groups > 0; the two uses follow one another on the stream, the larger size serves both)"""
    rows = d.N * d.To * d.Ho * d.Wo
    return max(_splitk_floats(rows, d.K, cap), lib.det_parts_floats(rows, d.K, stat_groups, 2) if stat_groups else 0)


def fwd_ws(d, device, stat_groups=0, cap=16):
    n = fwd_ws_floats(d, stat_groups, cap)
    return torch.empty((n,), dtype=torch.float32, device=device) if n else None


STAT_SLOTS = 16         # = MSCL_STAT_SLOTS (include/mscl_hip.h): BN statistics buffers are [slots][2][C]
STAT_ACTIVE = 4         # = MSCL_STAT_ACTIVE (csrc/common.h): the slots the atomic producers use and the consumers add outside deterministic mode


def new_stats(C, device, groups=1):
    """zeroed [groups][slots][2][C] statistics buffer (groups: mscl_conv3d_fwd_groups); element [0, 0] / [0, 1] of group 0 is
    what the conv's stats arguments and the BatchNorm parameter block point at"""
    buf = ZEROS.take(groups * STAT_SLOTS * 2 * C, device)
    return buf.view(STAT_SLOTS, 2, C) if groups == 1 else buf.view(groups, STAT_SLOTS, 2, C)


PROFILE = None           # bench.py: dict(events=[]) -> (mode, desc fields, event pair) around every conv launch (eager steps)
_DESC_FIELDS = ('N', 'T', 'H', 'W', 'C', 'To', 'Ho', 'Wo', 'K', 'kT', 'kH', 'kW', 'sT', 'sH', 'sW')


def prof_begin():
    if PROFILE is None:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0


def prof_end(e0, mode, d):
    if e0 is None:
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    PROFILE['events'].append((mode, tuple(getattr(d, f) for f in _DESC_FIELDS), e0, e1))


def conv3d_fwd(x, w, d, bias=None, addend=None, relu=False, stats=None, split_cap=16):
    """y = conv(x, w) (+bias) (+addend) (relu).  stats = (sum, sumsq) fp32 K-vectors, pre-zeroed."""
    y = torch.empty(out_shape(d), dtype=torch.bfloat16, device=x.device)
    s0, s1 = stats if stats is not None else (None, None)
    ws = fwd_ws(d, x.device, 1 if stats is not None else 0, split_cap)
    e0 = prof_begin()
    call('mscl_conv3d_fwd', ctypes.byref(d), ptr(x), ptr(w), ptr(y), ptr(bias), ptr(addend), int(relu),
         ptr(s0), ptr(s1), ptr(ws), ws.numel() if ws is not None else 0, stream_ptr())
    prof_end(e0, 'fwd', d)
    return y


def conv3d_dgrad(dy, wT, d, addend=None, split_cap=16, out=None):
    """out: an existing (N, T, H, W, C) bf16 map to ACCUMULATE into (dx = out += ...; the library's in-place form, include/mscl_hip.h)"""
    if out is not None:
        assert addend is None and out.is_contiguous() and tuple(out.shape) == (d.N, d.T, d.H, d.W, d.C) and out.dtype == torch.bfloat16
        assert max(d.sT, d.sH, d.sW) > 1, 'in place is defined for strided convs only (include/mscl_hip.h)'
        dx = addend = out
    else:
        dx = torch.empty((d.N, d.T, d.H, d.W, d.C), dtype=torch.bfloat16, device=dy.device)
    ws = _splitk_ws(d.N * d.T * d.H * d.W, d.C, dy.device, split_cap)
    e0 = prof_begin()
    call('mscl_conv3d_dgrad', ctypes.byref(d), ptr(dy), ptr(wT), ptr(dx), ptr(addend), ptr(ws), ws.numel() if ws is not None else 0, stream_ptr())
    prof_end(e0, 'dgrad', d)
    return dx


def wgrad_ws_floats(d, with_bias):
    """floats of fp32 scratch the weight gradient of this layer wants (a function of the descriptor and of the library's mode:
    callers on the hot path cache it per (module, shape, lib.DET_GEN))"""
    # slabs of the window-resident / shared-tap kernels, or deterministic mode's per-split slabs (+ bias partials)
    return lib.call_raw('mscl_conv3d_wgrad_ws', ctypes.byref(d), int(with_bias))


def conv3d_wgrad(x, dy, d, dw, dbias=None, ws_floats=None):
    """dw (fp32 [K][taps][C], accumulated), dbias (fp32 [K], accumulated)."""
    n = wgrad_ws_floats(d, dbias is not None) if ws_floats is None else ws_floats
    ws = torch.empty((n,), dtype=torch.float32, device=x.device) if n > 0 else None
    e0 = prof_begin()
    call('mscl_conv3d_wgrad', ctypes.byref(d), ptr(x), ptr(dy), ptr(dw), ptr(dbias), ptr(ws),
         ws.numel() if ws is not None else 0, stream_ptr())
    prof_end(e0, 'wgrad', d)


def weight_transpose(w, wT, Cout, taps, Cin):
    call('mscl_weight_transpose', ptr(w), ptr(wT), Cout, taps, Cin, stream_ptr())


def build_transpose_table(entries, device):
    """entries: [(w bf16 tensor, wT bf16 tensor, Cout, taps, Cin)] -> (device table, n, total_blocks)"""
    import struct
    raw, first = b'', 0
    for w, wT, co, taps, ci in entries:
        raw += struct.pack('<QQiiii', w.data_ptr(), wT.data_ptr(), co, taps, ci, first)
        # block ownership must mirror weight_transpose_batched_kernel: 64x64 tiles per tap, else 256 elements
        first += (co // 64) * (ci // 64) * taps if (co % 64 == 0 and ci % 64 == 0) else (co * taps * ci + 255) // 256
    table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
    return table, len(entries), first


def weight_transpose_batched(table, n, total_blocks):
    call('mscl_weight_transpose_batched', ptr(table), n, total_blocks, stream_ptr())


class ZeroPool:
    """Pre-zeroed fp32 scratch handed out in slices and re-zeroed with ONE memset per step (BatchNorm
    statistics / reduction scratch used to cost ~190 tiny fill launches per step)."""

    def __init__(self):
        self.buf, self.used, self.high = None, 0, 0

    def reset(self, device, size=4 << 20):
        if self.buf is None or self.buf.device != torch.device(device):
            self.buf = torch.zeros(size, dtype=torch.float32, device=device)
            self.high = 0
        else:
            # Re-zero what has ever been handed out, not just the previous step's share: the consumers differ between
            # launch modes (branches replayed from sub-graphs bring their own pools), and a reset that is being captured
            # is replayed for every later step, so it clears the whole buffer (16 MB, ~5 us).
            self.high = max(self.high, self.used)
            if torch.cuda.is_current_stream_capturing():
                self.buf.zero_()
            elif self.high:
                self.buf[:self.high].zero_()
        self.used = 0

    def take(self, n, device):
        n64 = (n + 63) // 64 * 64
        if self.buf is None or self.buf.device != torch.device(device) or self.used + n64 > self.buf.numel():
            return torch.zeros(n, dtype=torch.float32, device=device)     # outside a step, or pool exhausted
        out = self.buf[self.used:self.used + n]
        self.used += n64
        return out


ZEROS = ZeroPool()


def _bnp(stats, gamma, beta, rm, rv, nbt, smean, sinv):
    return BnParams(ptr(stats[0]), ptr(stats[1]), ptr(gamma), ptr(beta), ptr(rm), ptr(rv), ptr(nbt), ptr(smean), ptr(sinv))


def bn_act_fwd(y, bn, residual=None, res_bn=None, relu=True, eps=1e-5, momentum=0.1):
    """bn / res_bn: BnParams.  Returns out (bf16, same shape as y)."""
    out = torch.empty_like(y)
    C = y.shape[-1]
    rows = y.numel() // C
    call('mscl_bn_act_fwd', ptr(y), ctypes.byref(bn), ptr(residual),
         ctypes.byref(res_bn) if res_bn is not None else None, ptr(out), rows, C, eps, momentum, int(relu), stream_ptr())
    return out


def bn_act_bwd(dout, out, y, gamma, smean, sinv, dgamma, dbeta, relu, scratch, res=None, want_identity_dres=False, beta=None,
               groups=1):
    """res = None | dict(y=, gamma=, mean=, invstd=, dgamma=, dbeta=).  Returns (dy, dres|None).
    beta (no residual, relu): ReLU mask recomputed from y instead of read from `out`.
    groups: BatchNorm statistics groups (smean / sinv [groups][C], scratch [groups][STAT_SLOTS][4C])."""
    C = y.shape[-1]
    rows = y.numel() // C
    dy = torch.empty_like(y)
    dres = torch.empty_like(y) if (res is not None or want_identity_dres) else None
    r = res or {}
    pn = lib.det_parts_floats(rows, C, groups, 4) if relu != 2 else 0       # deterministic mode: scratch for the per-block partials
    parts = torch.empty((pn,), dtype=torch.float32, device=y.device) if pn else None
    call('mscl_bn_act_bwd_groups', ptr(dout), ptr(out), ptr(y), ptr(gamma), ptr(beta), ptr(smean), ptr(sinv), ptr(dgamma), ptr(dbeta),
         ptr(r.get('y')), ptr(r.get('gamma')), ptr(r.get('mean')), ptr(r.get('invstd')), ptr(r.get('dgamma')),
         ptr(r.get('dbeta')), ptr(dy), ptr(dres), int(want_identity_dres and res is None), ptr(scratch), rows, C,
         int(relu), groups, ptr(parts), pn, stream_ptr())
    return dy, dres


class IndirectInput:
    """A (B,C,T,H,W) contiguous fp32 clip whose device address is read from a device word (`slot`, one int64) at kernel time:
    what a whole-step HIP graph takes as its inputs, so that a new batch costs a pointer upload instead of a copy into static
    buffers (graph.py).  Quacks like the tensor for the few attributes the step touches before mscl_pack_input reads it."""

    def __init__(self, slot, shape, device):
        self.slot, self.shape, self.device = slot, tuple(shape), device

    def contiguous(self):
        return self

    def is_contiguous(self):
        return True


def pack_input(x, mean=None, std=None, t_off=0, T=None, flip=None, out=None):
    """frames [t_off, t_off+T) of (B,C<=3,Ttot,H,W) fp32 -> (B,T,H,W,8) bf16, optional (x-mean)/std, optional
    per-sample horizontal flip (uint8 mask of B entries on the device)."""
    B, C, Ttot, H, W = x.shape
    T = Ttot - t_off if T is None else T
    if not x.is_contiguous():
        raise lib.MsclError('pack_input needs a contiguous NCTHW tensor')
    if out is None:            # `out`: a (B,T,H,W,8) slice of a larger batch (base || rotated flow clips in one trunk pass)
        out = torch.empty((B, T, H, W, 8), dtype=torch.bfloat16, device=x.device)
    elif tuple(out.shape) != (B, T, H, W, 8) or not out.is_contiguous():
        raise lib.MsclError('pack_input: `out` must be a contiguous (B,T,H,W,8) bf16 tensor')
    m = (ctypes.c_float * 3)(*mean) if mean is not None else None
    s = (ctypes.c_float * 3)(*std) if std is not None else None
    if isinstance(x, IndirectInput):
        call('mscl_pack_input_ind', ptr(x.slot), ptr(out), B, C, T, H, W, Ttot, t_off, m, s, ptr(flip), stream_ptr())
    else:
        call('mscl_pack_input', ptr(x), ptr(out), B, C, T, H, W, Ttot, t_off, m, s, ptr(flip), stream_ptr())
    return out


def pair_w(x):
    """(B,T,H,W,8) packed 3-channel clip -> (B,T,H,(W+1)/2+1,8): pixel pairs along W as channels (RGB stem, see mscl_pair_w)"""
    B, T, H, W, C = x.shape
    if C != 8:
        raise lib.MsclError(f'pair_w needs (B,T,H,W,8), got {tuple(x.shape)}')
    out = torch.empty((B, T, H, (W + 1) // 2 + 1, 8), dtype=torch.bfloat16, device=x.device)
    call('mscl_pair_w', ptr(x), ptr(out), B * T * H, W, stream_ptr())
    return out


def pair_w_weight(w, w8):
    """w physical (Cout,kT,kH,7,3) -> w8 (Cout,kT,kH,4,8): w8[..., j, 3p + c] = w[..., 2j + p, c]"""
    w8[..., :, 0:3].copy_(w[..., 0::2, :])
    w8[..., 0:3, 3:6].copy_(w[..., 1::2, :])


def pair_w_grad_fold(dw8, g):
    """inverse of pair_w_weight for the gradient staging buffer: g (Cout,kT,kH,7,3) += the 21 live slots of dw8"""
    g[..., 0::2, :].add_(dw8[..., :, 0:3])
    g[..., 1::2, :].add_(dw8[..., 0:3, 3:6])


AUG_PARAMS = 16          # floats per sample in the colour-augmentation parameter rows (include/mscl_hip.h)


def color_aug(x, params, blur_ksize=0):
    """colour jitter / grayscale (and, with blur_ksize > 0, the Gaussian blur) of a (B,3,T,H,W) fp32 clip in [0,1],
    given per-sample parameter rows (B,16) on the device (ssl_aug_v2.py:31-43); returns a new fp32 clip."""
    B, C, T, H, W = x.shape
    if C != 3 or not x.is_contiguous() or x.dtype != torch.float32:
        raise lib.MsclError('color_aug needs a contiguous fp32 (B,3,T,H,W) clip')
    if tuple(params.shape) != (B, AUG_PARAMS) or params.dtype != torch.float32 or not params.is_contiguous():
        raise lib.MsclError(f'color_aug needs ({B},{AUG_PARAMS}) fp32 parameter rows')
    out = torch.empty_like(x)
    call('mscl_color_aug', ptr(x), ptr(out), ptr(params), B, T, H, W, stream_ptr())
    if blur_ksize:
        tmp = torch.empty_like(x)
        call('mscl_gauss_blur', ptr(out), ptr(tmp), ptr(out), ptr(params), blur_ksize, B, 3 * T, H, W, stream_ptr())
    return out


def flow_visualize(uv, t_off=0, T=None, flip=None, want_levels=False, out=None):
    """(B,2,Ttot,H,W) fp32 optical flow -> colour-wheel image (B,T,H,W,8) bf16 (ssl_aug.py:87-136);
    with want_levels also the quantised bytes (B,T,H,W,3)."""
    B, C, Ttot, H, W = uv.shape
    T = Ttot - t_off if T is None else T
    if C != 2 or not uv.is_contiguous():
        raise lib.MsclError('flow_visualize needs a contiguous (B,2,T,H,W) tensor')
    if out is None:
        out = torch.empty((B, T, H, W, 8), dtype=torch.bfloat16, device=uv.device)
    elif tuple(out.shape) != (B, T, H, W, 8) or not out.is_contiguous():
        raise lib.MsclError('flow_visualize: `out` must be a contiguous (B,T,H,W,8) bf16 tensor')
    lv = torch.empty((B, T, H, W, 3), dtype=torch.uint8, device=uv.device) if want_levels else None
    call('mscl_flow_visualize', ptr(uv), ptr(out), ptr(lv), B, T, H, W, Ttot, t_off, ptr(flip), stream_ptr())
    return (out, lv) if want_levels else out


def flow_fra_visualize(uv, cid, ratios=(0.2, 1.8), num_chunks=8, flip=None, want_debug=False):
    """raw flow (B,2,T,H,W) fp32 + chunk ids (B,) int32 -> colour images of the base and the rotated copy, (B,2T,H,W,8) bf16
    (Flow Rotation Augmentation, transforms_motion.py:103-142, + FlowVisualizer).  want_debug also returns
    (levels (B,2T,H,W,3) uint8, normed (B,2T,H,W,2) fp32)."""
    B, C, T, H, W = uv.shape
    if C != 2 or not uv.is_contiguous() or cid.dtype != torch.int32:
        raise lib.MsclError('flow_fra_visualize needs a contiguous (B,2,T,H,W) tensor and int32 chunk ids')
    out = torch.empty((B, 2 * T, H, W, 8), dtype=torch.bfloat16, device=uv.device)
    lv = torch.empty((B, 2 * T, H, W, 3), dtype=torch.uint8, device=uv.device) if want_debug else None
    nm = torch.empty((B, 2 * T, H, W, 2), dtype=torch.float32, device=uv.device) if want_debug else None
    scratch = torch.empty((2 * B * T,), dtype=torch.float64, device=uv.device)
    call('mscl_flow_fra_visualize', ptr(uv), ptr(cid), float(ratios[0]), float(ratios[1]), int(num_chunks), ptr(out), ptr(lv),
         ptr(nm), ptr(scratch), B, T, H, W, ptr(flip), stream_ptr())
    return (out, lv, nm) if want_debug else out


def crop_resize(src, boxes, out_hw):
    """paired-view crop + cv2-style bilinear resize (+ / 255 for uint8 frames) in one pass.
    src: (B,T,Hs,Ws,C) uint8 (C = 3) or fp32 (C <= 16), dense in its last four axes (a slice along T of a larger upload is
    fine); boxes: (B,4) int32 {x1,y1,x2,y2} on the device; returns (B,C,T,Ho,Wo) fp32."""
    B, T, Hs, Ws, C = src.shape
    if src.stride()[1:] != (Hs * Ws * C, Ws * C, C, 1) or boxes.dtype != torch.int32 or tuple(boxes.shape) != (B, 4):
        raise lib.MsclError('crop_resize needs frames dense in (T,H,W,C) and (B,4) int32 boxes')
    Ho, Wo = out_hw
    out = torch.empty((B, C, T, Ho, Wo), dtype=torch.float32, device=src.device)
    bs = src.stride(0)
    if src.dtype == torch.uint8:
        if C != 3:
            raise lib.MsclError('uint8 frames must be RGB')
        call('mscl_crop_resize_u8', ptr(src), ptr(boxes), ptr(out), B, T, Hs, Ws, Ho, Wo, bs, stream_ptr())
    elif src.dtype == torch.float32:
        call('mscl_crop_resize_f32', ptr(src), ptr(boxes), ptr(out), B, T, Hs, Ws, C, Ho, Wo, bs, stream_ptr())
    else:
        raise lib.MsclError(f'crop_resize: unsupported dtype {src.dtype}')
    return out


def add_relu(a, b=None, c=None, relu=False):
    out = torch.empty_like(a)
    call('mscl_add_relu', ptr(a), ptr(b), ptr(c), ptr(out), a.numel(), int(relu), stream_ptr())
    return out


def relu_bwd(dout, out):
    din = torch.empty_like(dout)
    call('mscl_relu_bwd', ptr(dout), ptr(out), ptr(din), dout.numel(), stream_ptr())
    return din


def upsample_add(src, dst, trilinear, accumulate):
    N, Ts, Hs, Ws, C = src.shape
    _, Td, Hd, Wd, _ = dst.shape
    call('mscl_upsample_add', ptr(src), ptr(dst), N, Ts, Hs, Ws, Td, Hd, Wd, C, int(trilinear), int(accumulate), stream_ptr())
    return dst


def upsample_bwd(ddst, src_shape, trilinear):
    N, Ts, Hs, Ws, C = src_shape
    _, Td, Hd, Wd, _ = ddst.shape
    dsrc = torch.empty(src_shape, dtype=torch.bfloat16, device=ddst.device)
    call('mscl_upsample_bwd', ptr(ddst), ptr(dsrc), N, Ts, Hs, Ws, Td, Hd, Wd, C, int(trilinear), stream_ptr())
    return dsrc


def maxpool_hw_fwd(x):
    """(1,3,3) / (1,2,2) / (0,1,1) max-pool of an NDHWC bf16 map -> (out, win); win feeds maxpool_hw_bwd"""
    N, T, H, W, C = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((N, T, Ho, Wo, C), dtype=torch.bfloat16, device=x.device)
    win = torch.empty((N, T, Ho, Wo, C // 8), dtype=torch.int32, device=x.device)
    call('mscl_maxpool_hw_fwd', ptr(x), ptr(out), ptr(win), N * T, H, W, C, stream_ptr())
    return out, win


def maxpool_hw_bwd(dout, win, x_shape):
    N, T, H, W, C = x_shape
    dx = torch.empty(x_shape, dtype=torch.bfloat16, device=dout.device)
    call('mscl_maxpool_hw_bwd', ptr(dout), ptr(win), ptr(dx), N * T, H, W, C, stream_ptr())
    return dx


def pool_fwd(x, outer, inner):
    C = x.shape[-1]
    out = torch.empty((outer, C), dtype=torch.float32, device=x.device)
    call('mscl_pool_fwd', ptr(x), ptr(out), outer, inner, C, stream_ptr())
    return out


def pool_bwd(dout, shape, outer, inner, into=None):
    C = shape[-1]
    dx = into if into is not None else torch.empty(shape, dtype=torch.bfloat16, device=dout.device)
    call('mscl_pool_bwd', ptr(dout), ptr(dx), outer, inner, C, int(into is not None), stream_ptr())
    return dx


LIN_MAX_ROWS = 32          # rows per launch of the small fp32 linear kernels (LIN_MAX_ROWS in csrc/elementwise.hip)


def linear_fwd(x, w, b, relu):
    rows, in_f = x.shape
    out_f = w.shape[0]
    y = torch.empty((rows, out_f), dtype=torch.float32, device=x.device)
    for r0 in range(0, rows, LIN_MAX_ROWS):            # more rows than one launch takes (the LMCL flow transform: B * 2t rows)
        n = min(LIN_MAX_ROWS, rows - r0)
        call('mscl_linear_fwd', ptr(x) + 4 * r0 * in_f, ptr(w), ptr(b), ptr(y) + 4 * r0 * out_f, n, in_f, out_f, int(relu), stream_ptr())
    return y


def linear_bwd(x, w, y, dy, dw, db, relu, need_dx=True):
    rows, in_f = x.shape
    out_f = w.shape[0]
    dx = torch.empty_like(x) if need_dx else None
    for r0 in range(0, rows, LIN_MAX_ROWS):            # dw / db accumulate (+=) over the row chunks
        n = min(LIN_MAX_ROWS, rows - r0)
        call('mscl_linear_bwd', ptr(x) + 4 * r0 * in_f, ptr(w), ptr(y) + 4 * r0 * out_f, ptr(dy) + 4 * r0 * out_f,
             (ptr(dx) + 4 * r0 * in_f) if need_dx else None, ptr(dw), ptr(db), n, in_f, out_f, int(relu), stream_ptr())
    return dx


def l2norm_fwd(x):
    y = torch.empty_like(x)
    norms = torch.empty((x.shape[0],), dtype=torch.float32, device=x.device)
    call('mscl_l2norm_fwd', ptr(x), ptr(y), ptr(norms), x.shape[0], x.shape[1], stream_ptr())
    return y, norms


def l2norm_bwd(y, norms, dy):
    dx = torch.empty_like(y)
    call('mscl_l2norm_bwd', ptr(y), ptr(norms), ptr(dy), ptr(dx), y.shape[0], y.shape[1], stream_ptr())
    return dx


NCE_COLS = 128           # queue columns per block of the InfoNCE passes (NCE_BCOLS in csrc/contrast.hip): nblk = ceil(K / 128)


def nce_forward(queue, count, q, pos, inv_T, virt=None):
    """Returns (lse, loss_rows, rank) for R query rows against the aged queue snapshot.
    virt = (new_keys (n, dim) fp32, queue_ptr int64[1]): the snapshot AFTER those keys' enqueue, read from the buffers as they stand
    before it (mscl_nce_fwd_virt)."""
    R, dim = q.shape
    K = queue.shape[1]
    nblk = (K + NCE_COLS - 1) // NCE_COLS
    dev = q.device
    part = torch.empty((nblk, R, 3), dtype=torch.float32, device=dev)
    lse = torch.empty((R,), dtype=torch.float32, device=dev)
    loss = torch.empty((R,), dtype=torch.float32, device=dev)
    rank = torch.empty((R,), dtype=torch.int32, device=dev)
    vk, vp = virt if virt is not None else (None, None)
    call('mscl_nce_fwd_virt', ptr(queue), ptr(count), ptr(q), ptr(pos), ptr(part), R, dim, K, inv_T,
         ptr(vk), vk.shape[0] if vk is not None else 0, ptr(vp), stream_ptr())
    call('mscl_nce_finish', ptr(part), ptr(pos), ptr(lse), ptr(loss), ptr(rank), R, nblk, inv_T, stream_ptr())
    return lse, loss, rank


def nce_backward(queue, count, q, lse, row_scale, inv_T, virt=None, pos_pair=None):
    """dq (R, dim) = inv_T * row_scale[r] * sum_k softmax_k * W[:, k] (negatives only).
    pos_pair = (kpos (R, dim), pos (R,)): the positive pair's term of nce_pos_bwd is added in the same launches."""
    R, dim = q.shape
    dq = ZEROS.take(R * dim, q.device).view(R, dim)          # (a slice of the step's pre-zeroed pool: no fill launch in the loss phase)
    Kq = queue.shape[1]
    ws = torch.empty(((Kq + 127) // 128) * ((min(R, 32) + 7) // 8 * 8) * dim, dtype=torch.float32, device=q.device)
    vk, vp = virt if virt is not None else (None, None)
    kp, ps = pos_pair if pos_pair is not None else (None, None)
    call('mscl_nce_bwd_virt', ptr(queue), ptr(count), ptr(q), ptr(lse), ptr(row_scale), ptr(dq), ptr(ws), ws.numel(), R, dim, Kq,
         inv_T, ptr(vk), vk.shape[0] if vk is not None else 0, ptr(vp), ptr(kp), ptr(ps), stream_ptr())
    return dq


def rowdot(a, b):
    out = torch.empty((a.shape[0],), dtype=torch.float32, device=a.device)
    call('mscl_rowdot', ptr(a), ptr(b), ptr(out), a.shape[0], a.shape[1], stream_ptr())
    return out


def loss_pack(q_rgb, q_fb, q_fa, k_rgb, k_fb, k_fa, p_fb, p_fa, t, use_aug, w_intra):
    """one launch for the loss phase's row layout (mscl_loss_pack): returns views QA, KA, sA, QC, KC, sC, ones, flow of one buffer,
    that buffer, and the positive logits (posA, posB, posC) of the three passes' rows"""
    B, D = q_rgb.shape
    Cf = p_fb.shape[1]
    n = 3 if use_aug else 2
    sizes = [n * B * D] * 4 + [n * B, n * B, B, B * 2 * t * Cf, n * B, B, n * B]
    ws = torch.empty((sum(sizes),), dtype=torch.float32, device=q_rgb.device)
    call('mscl_loss_pack', ptr(q_rgb), ptr(q_fb), ptr(q_fa), ptr(k_rgb), ptr(k_fb), ptr(k_fa), ptr(p_fb), ptr(p_fa), ptr(ws),
         ptr(ws[sum(sizes[:8]):]), B, D, t, Cf, int(use_aug), float(w_intra), stream_ptr())
    parts, o = [], 0
    for sz in sizes:
        parts.append(ws[o:o + sz]); o += sz
    QA, KA, QC, KC, sA, sC, ones, flow, posA, posB, posC = parts
    return (QA.view(n * B, D), KA.view(n * B, D), sA, QC.view(n * B, D), KC.view(n * B, D), sC, ones, flow.view(B, 2 * t, Cf), ws,
            (posA, posB, posC))


def loss_unpack(dA, dB, dC, dpr, dpf, B, D, t, C, Cf, use_aug):
    """one launch for the loss node's input gradients (mscl_loss_unpack): flat buffer + views dq_rgb, dq_fb, dq_fa, dp_rgb, dp_fb, dp_fa"""
    sizes = [B * D] * 3 + [B * t * C, B * t * Cf, B * t * Cf]
    out = torch.empty((sum(sizes),), dtype=torch.float32, device=dA.device)
    call('mscl_loss_unpack', ptr(dA), ptr(dB), ptr(dC), ptr(dpr), ptr(dpf), ptr(out), B, D, t, C, Cf, int(use_aug), stream_ptr())
    return out, sizes


def nce_pos_bwd(kpos, pos, lse, row_scale, dq, inv_T):
    call('mscl_nce_pos_bwd', ptr(kpos), ptr(pos), ptr(lse), ptr(row_scale), ptr(dq), dq.shape[0], dq.shape[1], inv_T, stream_ptr())


def queue_enqueue(queue, count, qptr, keys):
    n, dim = keys.shape
    call('mscl_queue_enqueue', ptr(queue), ptr(count), ptr(qptr), ptr(keys), n, dim, queue.shape[1], stream_ptr())


def lmcl(rgb, flow, inv_T):
    """rgb (B,t,C), flow (B,2t,C) fp32 pooled features -> loss_sum[1], hits[2], drgb, dflow."""
    B, t, C = rgb.shape
    dev = rgb.device
    loss_sum = ZEROS.take(1, dev)
    hits = ZEROS.take(2, dev).view(torch.int32)
    drgb, dflow = torch.empty_like(rgb), torch.empty_like(flow)
    call('mscl_lmcl', ptr(rgb), ptr(flow), ptr(loss_sum), ptr(hits), ptr(drgb), ptr(dflow), B, t, C, inv_T, stream_ptr())
    return loss_sum, hits, drgb, dflow


def step_logs(rankA, lossA, rankB, lossB, rankC, lossC, lmcl_sum, lmcl_hits, B, n_groups, w_intra, n_rows):
    """the step's log vector (MSCLWithAug key order, last = total loss) in one launch"""
    logs = torch.empty((23 if n_groups == 3 else 17,), dtype=torch.float32, device=lossA.device)
    call('mscl_step_logs', ptr(rankA), ptr(lossA), ptr(rankB), ptr(lossB), ptr(rankC), ptr(lossC), ptr(lmcl_sum), ptr(lmcl_hits),
         B, n_groups, n_groups, float(w_intra), float(n_rows), ptr(logs), stream_ptr())
    return logs


def ema_update(pk, pq, pk_bf16, m):
    call('mscl_ema_update', ptr(pk), ptr(pq), ptr(pk_bf16), pk.numel(), float(m), stream_ptr())


def ema_update_dev(pk, pq, pk_bf16, m_dev):
    call('mscl_ema_update_dev', ptr(pk), ptr(pq), ptr(pk_bf16), pk.numel(), ptr(m_dev), stream_ptr())


def sgd_step_dev(p, g, buf, p_bf16, sumsq_t, max_norm, lr_dev, momentum, wd):
    call('mscl_sgd_step_dev', ptr(p), ptr(g), ptr(buf), ptr(p_bf16), p.numel(), ptr(sumsq_t), float(max_norm), ptr(lr_dev),
         float(momentum), float(wd), stream_ptr())


_SUMSQ_PARTIALS = {}


def sumsq(g, out):
    """out[0] = sum(g^2), bit-reproducible (two-phase, no atomics)"""
    part = _SUMSQ_PARTIALS.get(g.device)
    if part is None:
        part = _SUMSQ_PARTIALS[g.device] = torch.empty(1024, dtype=torch.float32, device=g.device)
    call('mscl_sumsq', ptr(g), ptr(out), g.numel(), ptr(part), part.numel(), stream_ptr())


def sgd_step(p, g, buf, p_bf16, sumsq_t, max_norm, lr, momentum, wd, first):
    call('mscl_sgd_step', ptr(p), ptr(g), ptr(buf), ptr(p_bf16), p.numel(), ptr(sumsq_t), float(max_norm), float(lr),
         float(momentum), float(wd), int(first), stream_ptr())


def cast_bf16(src, dst):
    call('mscl_cast_bf16', ptr(src), ptr(dst), src.numel(), stream_ptr())
