"""Probe for run-to-run differences in deterministic mode: the step of tests/test_model_gpu.py::
test_graphed_step_equals_eager_bitwise_in_deterministic_mode (B = 2, T = 8, 32^2, K = 64; five optimizer steps from the same weights on
the same batches) run REPS times in each of three launch modes -- eager (with the key / query sub-graphs), whole-step graph reading
its inputs by address, whole-step graph with static inputs -- keeping, after every step, the loss, the gradient arena, the parameter
arena, the key arena and both queues.  Every run is compared with the first eager run; a difference is reported with the step it
first appears at, the buffer, how many elements differ and which parameters they belong to.
usage: python tools/flake_det.py [REPS [MODES]]   (MODES: comma-separated subset of eager,graph,graph_static; the first run is the reference)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_model_gpu as tm                                                        # noqa: E402
from mscl_amd import ClipSGD, lib                                                  # noqa: E402
from mscl_amd.graph import GraphedStep                                             # noqa: E402
from mscl_amd.synthetic import synthetic_batch                                     # noqa: E402


LOGS = [None]
LMCL = {}


def _tap_lmcl():
    """keep hold of the tensors of the last LMCL call (under a whole-step capture: the graph's static tensors, rewritten by every
    replay) so that a difference in `loss_pos` can be traced to the kernel's inputs or its outputs"""
    from mscl_amd import kernels as K
    orig = K.lmcl

    def lmcl(rgb, flow, inv_T):
        out = orig(rgb, flow, inv_T)
        LMCL.update(lm_rgb=rgb, lm_flow=flow, lm_sum=out[0], lm_hits=out[1], lm_drgb=out[2], lm_dflow=out[3])
        return out
    K.lmcl = lmcl
    # the RGB query neck's chain (trunk stage maps -> FPN outputs -> each PConv3D's outputs), held alive for the same purpose
    from mscl_amd import necks
    tpn_fwd, pconv_fwd, fpn_fwd = necks.TPNSingleHip.forward, necks.PConv3DHip.forward, necks.FPNHip.forward
    depth = [0]

    def tpn(self, feats, levels=None):
        if torch.is_grad_enabled():
            depth[0] = 0
            for i, f in enumerate(feats[-3:]):
                LMCL[f'n0_feat{i}'] = f
        return tpn_fwd(self, feats, levels=levels)

    def fpn(self, feats, levels=None):
        outs = fpn_fwd(self, feats, levels=levels)
        if torch.is_grad_enabled():
            for i, o in enumerate(outs):
                if o is not None:
                    LMCL[f'n1_fpn{i}'] = o
        return outs

    def pconv(self, xs, levels=None):
        outs = pconv_fwd(self, xs, levels=levels)
        if torch.is_grad_enabled():
            for i, o in enumerate(outs):
                if o is not None:
                    LMCL[f'n{2 + depth[0]}_pconv{i}'] = o
            depth[0] += 1
        return outs
    necks.TPNSingleHip.forward, necks.PConv3DHip.forward, necks.FPNHip.forward = tpn, pconv, fpn
    # and what lies between a PConv3D's inputs and outputs: every conv_bias / upsample result of the neck, in call order
    cb, up = necks.conv_bias, necks.upsample
    seq = [0]

    def conv_bias(conv, x, addend=None, relu=False):
        y = cb(conv, x, addend=addend, relu=relu)
        if torch.is_grad_enabled() and x.requires_grad:
            LMCL[f'm{seq[0]:02d}_conv_{tuple(y.shape)[1:4]}'] = y
            seq[0] += 1
        return y

    def upsample(src, size, trilinear):
        y = up(src, size, trilinear)
        if torch.is_grad_enabled() and src.requires_grad:
            LMCL[f'm{seq[0]:02d}_up_{tuple(y.shape)[1:4]}'] = y
            seq[0] += 1
        return y
    necks.conv_bias, necks.upsample = conv_bias, upsample
    if os.environ.get('FLAKE_SENTINEL'):
        # every conv output starts as 7.0 everywhere: a consumer that reads a row BEFORE its producer wrote it then shows a corner
        # replaced by 7, where a load that returns nothing shows a corner dropped
        import ctypes
        from mscl_amd.kernels import call, ptr, stream_ptr

        def conv3d_fwd(x, w, d, bias=None, addend=None, relu=False, stats=None):
            y = torch.full(K.out_shape(d), 7.0, dtype=torch.bfloat16, device=x.device)
            s0, s1 = stats if stats is not None else (None, None)
            ws = K.fwd_ws(d, x.device, 1 if stats is not None else 0)
            call('mscl_conv3d_fwd', ctypes.byref(d), ptr(x), ptr(w), ptr(y), ptr(bias), ptr(addend), int(relu),
                 ptr(s0), ptr(s1), ptr(ws), ws.numel() if ws is not None else 0, stream_ptr())
            return y
        K.conv3d_fwd = conv3d_fwd
    tpn0 = necks.TPNSingleHip.forward

    def tpn_reset(self, feats, levels=None):
        if torch.is_grad_enabled():
            seq[0] = 0
        return tpn0(self, feats, levels=levels)
    necks.TPNSingleHip.forward = tpn_reset
    if os.environ.get('FLAKE_DIAG'):
        _route_through_diag()


DIAG = {}


def _route_through_diag():
    """FLAKE_DIAG=1: every TRILINEAR up-sampling launch of the step goes through the self-checking twin of the kernel
    (tools/diag/upsample_diag.hip, built by tools/diag/build.sh): both sides of every stage compared per lane, disagreements
    recorded on the device with the wave's hardware ids; diag_report() prints the records that arrived since the last call"""
    import ctypes
    from mscl_amd import kernels as K
    from mscl_amd.kernels import ptr, stream_ptr
    lib_ = ctypes.CDLL(os.path.join(ROOT, 'tools', 'diag', 'libups_diag.so'))
    lib_.ups_diag_upsample_add.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 10 + [ctypes.c_void_p]
    lib_.ups_diag_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib_.ups_diag_use_b(int(os.environ.get('FLAKE_DIAG_USE_B', '0')))
    DIAG.update(lib=lib_, seen=0, words=lib_.ups_diag_rec_words())
    orig = K.upsample_add

    def upsample_add(src, dst, trilinear, accumulate):
        if not trilinear:
            return orig(src, dst, trilinear, accumulate)
        N, Ts, Hs, Ws, C = src.shape
        _, Td, Hd, Wd, _ = dst.shape
        rc = lib_.ups_diag_upsample_add(ptr(src), ptr(dst), N, Ts, Hs, Ws, Td, Hd, Wd, C, 1, int(accumulate), stream_ptr())
        if rc:
            raise RuntimeError(f'ups_diag_upsample_add -> {rc}')
        return dst
    K.upsample_add = upsample_add


def diag_report(tag=''):
    if not DIAG:
        return
    import ctypes
    lib_, W = DIAG['lib'], DIAG['words']
    cnt = (ctypes.c_uint * 8)()
    rec = (ctypes.c_uint * (2048 * W))()
    n = lib_.ups_diag_read(cnt, rec, 2048)
    print(f'[diag{tag}] records {cnt[0]} (kept {n}), diag launches {cnt[1]}', flush=True)
    import struct

    def f32(u):
        return struct.unpack('<f', struct.pack('<I', u))[0]
    for i in range(DIAG['seen'], max(n, 0)):
        R = rec[i * W:(i + 1) * W]
        hw = R[6]
        print(f'  rec {i}: off-mismatch {R[0] & 255:08b} wt-mismatch {(R[0] >> 8) & 255:08b} load-mismatch {(R[0] >> 16) & 255:08b} '
              f'zeroA {(R[0] >> 24) & 255:08b} zeroB {R[1] & 255:08b} result-mismatch {(R[1] >> 8) & 255:08b} | launch {R[2]} block {R[3]} '
              f'thread {R[4]} (lane {R[4] & 63}) e {R[5]} | HW_ID {hw:#010x} (wave {hw & 15} simd {(hw >> 4) & 3} cu {(hw >> 8) & 15} sh {(hw >> 12) & 1} '
              f'se {(hw >> 13) & 7}) XCC {R[7] & 15} t {R[8] | (R[9] << 32)} | corner {R[10]} off {R[11]}/{R[12]} wt {f32(R[13]):.6f}/{f32(R[14]):.6f} '
              f'A {R[15]:08x} {R[16]:08x} {R[17]:08x} {R[18]:08x} B {R[19]:08x} {R[20]:08x} {R[21]:08x} {R[22]:08x} C(after pause) {R[23]:08x} '
              f'{R[24]:08x} {R[25]:08x} {R[26]:08x} | fA0 {f32(R[27]):.6g} fB0 {f32(R[28]):.6g} | total {R[29]} dims {R[30] & 255}x{(R[30] >> 8) & 255}x{R[30] >> 16} C {R[31]} '
              f'| wtA {[round(f32(R[32 + j]), 5) for j in range(8)]} wtB {[round(f32(R[40 + j]), 5) for j in range(8)]}', flush=True)
    DIAG['seen'] = max(n, DIAG['seen'])


KEYS = ('lm_rgb', 'lm_flow', 'lm_sum', 'lm_hits', 'lm_drgb', 'lm_dflow', 'loss', 'G', 'Q', 'KX', 'Qb', 'Kb', 'queue', 'queue_flow', 'count', 'count_flow', 'buffers')


def run(mode, dev):
    B, T, H, Kq = 2, 8, 32, 64
    model, cfg = tm.build(T, Kq, dev)
    opt = ClipSGD.from_cfg(model, cfg.optimizer, cfg.optimizer_config)
    batches = [synthetic_batch(B, T, H, H, 0, s, device=dev) for s in range(3)]
    snaps = []

    def snap(loss):
        torch.cuda.synchronize()
        a = model.arena
        bufs = torch.cat([t.detach().float().flatten() for n, t in sorted(model.named_buffers()) if 'queue' not in n])
        snaps.append(dict(loss=float(loss), logs=LOGS[0], G=a.G.clone(), Q=a.Q.clone(), KX=a.KX.clone(), Qb=a.Qb.clone(), Kb=a.Kb.clone(),
                          queue=model.recognizer.queue.clone(), queue_flow=model.recognizer_flow.queue.clone(),
                          count=model.recognizer.count.clone(), count_flow=model.recognizer_flow.count.clone(), buffers=bufs))
        snaps[-1].update({k: v.clone() for k, v in LMCL.items()})
    if mode == 'eager':
        for s in (0, 0, 0, 1, 2):
            out = model.train_step(batches[s], sync_logs=False)
            opt.zero_grad(); out['loss'].backward(); opt.step()
            LOGS[0] = {k: float(v) for k, v in out['log_vars'].items()} if 'log_vars' in out else None
            snap(out['loss'].detach())
        snaps = snaps[2:]
    else:
        gs = GraphedStep(model, opt, batches[0], warmup=2, indirect=None if mode == 'graph' else False)
        for s in range(3):
            r = gs.step(batches[s])
            LOGS[0] = [float(v) for v in r[1]] if len(r) > 1 and r[1] is not None else None
            snap(r[0])
    return snaps, model


def names(model, idx):
    out = []
    for sl in model.arena.slots:
        n = int(((idx >= sl.off) & (idx < sl.off + sl.numel)).sum())
        if n:
            out.append(f'{sl.name}:{n}')
    return out[:12]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    global MODES
    MODES = tuple(sys.argv[2].split(',')) if len(sys.argv) > 2 else ('eager', 'graph', 'graph_static')
    dev = torch.device('cuda', 0)
    lib.set_deterministic(True)
    _tap_lmcl()
    ref = None
    bad = 0
    for rep in range(reps):
        for mode in MODES:
            snaps, model = run(mode, dev)
            if ref is None:
                ref = snaps
                continue
            hit = None
            for step, (a, b) in enumerate(zip(ref, snaps)):
                for key in sorted(k for k in a if k[0] in 'mn' and k[1].isdigit()) + list(KEYS):
                    same = a[key] == b[key] if key == 'loss' else torch.equal(a[key], b[key])
                    if not same:
                        hit = (step, key)
                        break
                if hit:
                    break
            if hit:
                bad += 1
                step, key = hit
                msg = f'rep {rep} mode {mode}: first difference at step {step} in {key}'
                if key == 'loss':
                    msg += f' ({ref[step]["loss"]!r} vs {snaps[step]["loss"]!r})'
                else:
                    d = (ref[step][key] != snaps[step][key]).flatten()
                    idx = d.nonzero().flatten()
                    msg += f': {int(d.sum())} of {d.numel()} elements, first index {int(idx[0])}, last {int(idx[-1])}'
                    if key in ('G', 'Q', 'KX'):
                        msg += ' ' + ' '.join(names(model, idx))
                    md = float((ref[step][key].flatten()[idx] - snaps[step][key].flatten()[idx]).abs().max())
                    msg += f' max |diff| {md:.3g}'
                # what else differs at that step
                other = [k for k in sorted(ref[step]) if k not in ('loss', 'logs') and not torch.equal(ref[step][k], snaps[step][k])]
                print(msg + f' | differing buffers at that step: {other}', flush=True)
                if '_up_' in key and step > 0:
                    # is the wrong row what the SAME launch produced one replay earlier, or what it would produce from the previous
                    # step's source (a stale read)?
                    from mscl_amd import kernels as K
                    keys_sorted = sorted(k for k in snaps[step] if k[0] == 'm' and k[1].isdigit())
                    src_key = keys_sorted[keys_sorted.index(key) - 1]
                    got, want = snaps[step][key], ref[step][key]
                    C = got.shape[-1]
                    rows = sorted({int(i) // C for i in idx})
                    prev_out = snaps[step - 1][key]
                    dst = torch.empty_like(got)
                    K.upsample_add(snaps[step - 1][src_key], dst, True, accumulate=False)
                    cur_src, prv_src = snaps[step][src_key].float(), snaps[step - 1][src_key].float()
                    Ns, Ts_, Hs_, Ws_, _ = cur_src.shape
                    _, Td_, Hd_, Wd_, _ = got.shape

                    def lin(dd, n_in, n_out):
                        import numpy as np
                        sc = np.float32(n_in) / np.float32(n_out)
                        sv = max(np.float32(0), (np.float32(dd) + np.float32(0.5)) * sc - np.float32(0.5))
                        i0 = min(int(sv), n_in - 1); i1 = min(i0 + 1, n_in - 1)
                        return i0, i1, float(sv - np.float32(i0))
                    for r in rows:
                        n_, rem = divmod(r, Td_ * Hd_ * Wd_); t_, rem = divmod(rem, Hd_ * Wd_); h_, w_ = divmod(rem, Wd_)
                        (t0, t1, a_), (h0, h1, b_), (w0, w1, c_) = lin(t_, Ts_, Td_), lin(h_, Hs_, Hd_), lin(w_, Ws_, Wd_)
                        gr, rt = got.reshape(-1, C)[r].float(), want.reshape(-1, C)[r].float()
                        best = []
                        for k in range(8):
                            tt, hh, ww = (t1 if k & 4 else t0), (h1 if k & 2 else h0), (w1 if k & 1 else w0)
                            wt = (a_ if k & 4 else 1 - a_) * (b_ if k & 2 else 1 - b_) * (c_ if k & 1 else 1 - c_)
                            if wt == 0:
                                continue
                            cs, ps = cur_src[n_, tt, hh, ww], prv_src[n_, tt, hh, ww]
                            best.append((float((gr - (rt - wt * cs)).abs().max()), f'corner {k} (src {tt},{hh},{ww}; weight {wt:.3f}) DROPPED'))
                            best.append((float((gr - (rt + wt * (ps - cs))).abs().max()), f'corner {k} (src {tt},{hh},{ww}; weight {wt:.3f}) from the PREVIOUS step'))
                            best.append((float((gr - (rt + wt * (7.0 - cs))).abs().max()), f'corner {k} (src {tt},{hh},{ww}; weight {wt:.3f}) read as the SENTINEL 7.0'))
                        best.sort()
                        print(f'   row {r} = (n {n_}, t {t_}, h {h_}, w {w_}): best single-corner explanations: ' + '; '.join(f'{m} (residual {e:.3g})' for e, m in best[:3]), flush=True)
                    for r in rows:
                        g = got.reshape(-1, C)[r].float()
                        print(f'   row {r}: equals previous replay\'s output row: {bool(torch.equal(g, prev_out.reshape(-1, C)[r].float()))}; '
                              f'equals upsample(previous step\'s source) row: {bool(torch.equal(g, dst.reshape(-1, C)[r].float()))}; '
                              f'max |got - that| {float((g - dst.reshape(-1, C)[r].float()).abs().max()):.3g}, max |got - right| '
                              f'{float((g - want.reshape(-1, C)[r].float()).abs().max()):.3g}', flush=True)
        print(f'rep {rep} done, {bad} differing run(s) so far', flush=True)
        if DIAG and (rep % 20 == 19 or rep == reps - 1):
            diag_report(f' rep {rep}')
    lib.set_deterministic(False)


if __name__ == '__main__':
    main()
