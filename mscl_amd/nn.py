"""HIP-backed building blocks: parameter-holding nn.Modules with the reference's state-dict names, and
torch.autograd.Functions that call the gfx950 kernels through the C ABI.

PyTorch's role here is plumbing only: it owns device memory, the stream and the autograd *graph*; all
arithmetic on activations, parameters and gradients is done by libmscl_hip.so.  Parameter gradients
never travel through autograd: kernels accumulate them straight into the flat gradient arena
(mscl_amd/arena.py), so a fused block is one graph node.
"""
import ctypes


import torch
import torch.nn as nn

from . import kernels as K
from . import lib
from . import parallel
from .lib import MsclError


def _triple(v):
    return (v, v, v) if isinstance(v, int) else tuple(int(i) for i in v)


def _need_gpu(t):
    if not t.is_cuda:
        raise MsclError('mscl_amd modules compute on the GPU only: move the model with .materialize("cuda") '
                        'and feed CUDA tensors (there is no CPU fallback; the CPU restatement lives in oracle/ for tests)')


class Conv3dHip(nn.Module):
    """Parameter holder + kernel front-end for nn.Conv3d (state-dict: `weight` (Cout,Cin,kT,kH,kW), `bias`)."""

    def __init__(self, cin, cout, kernel, stride=1, padding=0, bias=False, pair_w=False):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.kernel_size, self.stride, self.padding = _triple(kernel), _triple(stride), _triple(padding)
        self.weight = nn.Parameter(torch.empty(cout, cin, *self.kernel_size))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        self.cin_eff = cin if cin % 8 == 0 else 8
        self.taps = self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        # W-paired execution of a 3-channel (kT,kH,7) / stride-2 / pad-3 stem (kernels.pair_w): the kernels see a
        # (kT,kH,4) / stride-1 / pad-1 convolution over pixel pairs; state dict and results are unchanged
        # the window-resident layer-1 kernels cover this shape (their launcher still checks the plane size)
        self.halo_shape = (cin == 64 and cout == 64 and self.kernel_size == (3, 3, 3) and self.stride == (1, 1, 1)
                           and self.padding == (1, 1, 1))
        self.pair_w = bool(pair_w)
        if self.pair_w and not (cin == 3 and self.kernel_size[2] == 7 and self.stride[2] == 2 and self.padding[2] == 3):
            raise ValueError('pair_w covers the 3-channel 7-wide stride-2 pad-3 stem only')
        self.k_exec = (self.kernel_size[0], self.kernel_size[1], 4) if self.pair_w else self.kernel_size
        self.s_exec = (self.stride[0], self.stride[1], 1) if self.pair_w else self.stride
        self.p_exec = (self.padding[0], self.padding[1], 1) if self.pair_w else self.padding
        self._rt = None           # runtime views, set by materialize()
        self._plans = {}          # per input shape: cached descriptor / workspace size (cba_fwd)
        self.split_cap = 16       # most split-K slabs a launch of this conv may use (kernels._splitk_floats): lower on side chains
        self._descs = {}

    def extra_repr(self):
        return f'{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, padding={self.padding}, bias={self.bias is not None}'

    def desc(self, x_shape):
        d = self._descs.get(tuple(x_shape))
        if d is None:
            d = K.conv_desc(tuple(x_shape), self.out_channels, self.k_exec, self.s_exec, self.p_exec)
            self._descs[tuple(x_shape)] = d
        return d

    def fwd(self, x, addend=None, relu=False, stats=None):
        rt = self._rt
        if rt is None:
            raise MsclError('model not materialized on a GPU: call model.materialize(device) first')
        return K.conv3d_fwd(x, rt['w'], self.desc(x.shape), bias=rt['bias'], addend=addend, relu=relu, stats=stats, split_cap=self.split_cap)

    def wT(self):
        """the transposed bf16 kernel the input-gradient kernels read, refreshed first if an optimizer step has left it behind the
        masters (TransposeState: the refresh is deferred off the step's serial tail, so EVERY reader comes through here)"""
        st = self._rt.get('wt_state')
        if st is not None:
            if st.stale:
                st.refresh()
            elif st.event is not None:
                st.sync_reader()
        return self._rt['wT']

    def dgrad(self, dy, x_shape, addend=None, out=None):
        return K.conv3d_dgrad(dy, self.wT(), self.desc(x_shape), addend=addend, split_cap=self.split_cap, out=out)

    def strided_pointwise(self):
        """a 1x1x1 conv with a stride: its input gradient reaches one position in prod(stride) -- cheaper added INTO an existing
        map (dgrad(out=...)) than written as a map of mostly zeros"""
        return tuple(self.kernel_size) == (1, 1, 1) and max(self.stride) > 1

    def wgrad(self, x, dy):
        """dw += ...: plain adds where one block owns an element (include/mscl_hip.h, mscl_conv3d_wgrad INVARIANT): every
        application of one conv module runs its weight gradient on ONE stream (the step keeps each trunk / neck on one chain)"""
        rt = self._rt
        d = self.desc(x.shape)
        key = ('w', tuple(x.shape), lib.DET_GEN)
        n = self._plans.get(key)
        if n is None:             # (one ctypes query per (shape, mode) instead of per launch: the eager step is host-bound)
            n = self._plans[key] = K.wgrad_ws_floats(d, rt['dbias'] is not None)
        K.conv3d_wgrad(x, dy, d, rt['dw'], rt['dbias'], ws_floats=n)
        rt['slot_w'].touched = True
        if rt['slot_b'] is not None:
            rt['slot_b'].touched = True


class TransposeState:
    """The transposed kernels [Cin][taps][Cout] of a model's convs (one batched launch refreshes all of them) and whether they lag
    the masters.  An optimizer step marks them stale instead of refreshing in its serial tail; the step refreshes them at its
    head on a side stream (recognizers.MSCLWithAug._device_step), and any OTHER backward entry -- encode_q + backward, a
    trunk-only loop, a custom step function -- refreshes them lazily at its first input gradient (Conv3dHip.wT), on the stream
    that gradient runs on."""

    def __init__(self, table):
        self.table, self.stale = table, False
        self.event, self.stream, self.waited = None, None, set()

    def refresh(self):
        """one launch on the CURRENT stream.  The copies are shared by every stream that runs input gradients, so the refresh
        leaves an event behind and a reader on another stream waits for it once (sync_reader) -- a custom multi-stream step
        whose first input gradient triggered the lazy refresh on one stream must not read half-written copies on another.
        (Under graph capture the step's own fork / join orders the refresh; no event is kept.)"""
        K.weight_transpose_batched(*self.table)
        self.stale = False
        self.stream = lib.stream_ptr()
        self.waited = {self.stream}
        if torch.cuda.is_current_stream_capturing():
            self.event = None
        else:
            self.event = torch.cuda.Event()
            self.event.record()

    def sync_reader(self):
        sp = lib.stream_ptr()
        if sp not in self.waited:
            # A capturing stream may not wait for an event recorded outside the capture (stream-capture isolation), and need not:
            # whoever began the capture on it had this stream wait for its parent first (torch.cuda.graph: capture stream <- the
            # caller's current stream), so what the refresh wrote is ordered before the captured launches.
            if not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream().wait_event(self.event)
            self.waited.add(sp)


class BatchNorm3dHip(nn.Module):
    """Parameter/buffer holder for nn.BatchNorm3d (training-mode batch statistics only)."""

    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = c, eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer('running_mean', torch.zeros(c))
        self.register_buffer('running_var', torch.ones(c))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        self._rt = None
        self._bnp = None          # cached parameter block of the fused forward (cba_fwd)

    def extra_repr(self):
        return f'{self.num_features}, eps={self.eps}, momentum={self.momentum}'


# BatchNorm statistics groups of the trunk pass being recorded (VideoResNetHip.forward(bn_groups=...)): a batch that holds
# the inputs of G consecutive calls of the reference's module -- the base and the rotated flow clips of one step
# (recognizers/mscl.py:239-240) -- runs every conv / weight-gradient / input-gradient kernel ONCE while BatchNorm keeps one set
# of batch statistics per call.  The backward reads the group count off the saved statistics' shape.
BN_GROUPS = [1]


def cba_fwd(conv, bn, x, residual, relu):
    """conv -> BN(batch stats fused into the conv epilogue) -> (+residual) -> (ReLU).
    Returns (y raw conv output, out, save[2,C] = mean/invstd; [2,G,C] with G = BN_GROUPS[0] > 1 statistics groups).

    The eager step is host-bound (~900 launches), so this path avoids per-call Python work: descriptor, split-K
    workspace size and the constant half of the BatchNorm parameter block are cached per (module, input shape)."""
    if not bn.training:
        raise MsclError('BatchNorm3dHip implements training-mode statistics only (both MoCo encoders run in train(), SURVEY App. E-6)')
    rt = conv._rt
    if rt is None:
        raise MsclError('model not materialized on a GPU: call model.materialize(device) first')
    shape = tuple(x.shape)
    pkey = (shape, lib.DET_GEN, conv.split_cap)        # (the scratch size below depends on the library's deterministic mode and on the split policy)
    plan = conv._plans.get(pkey)
    if plan is None:
        d = conv.desc(shape)
        plan = conv._plans[pkey] = (d, ctypes.byref(d), K.out_shape(d), K.fwd_ws_floats(d, 2 if d.N % 2 == 0 else 1, conv.split_cap),
                                     (d.N, d.T, d.H, d.W, d.C, d.K, d.kT))
    d, dref, oshape, ws_n, sig = plan
    C = conv.out_channels
    dev = x.device
    G = BN_GROUPS[0]
    stats = K.new_stats(C, dev, G)                 # [G][slots][2][C]; the pointers address group 0, slot 0
    buf = torch.empty((2,) + oshape, dtype=torch.bfloat16, device=dev)
    y, out = buf[0], buf[1]
    save = torch.empty((2, C) if G == 1 else (2, G, C), dtype=torch.float32, device=dev)
    ws = torch.empty((ws_n,), dtype=torch.float32, device=dev) if ws_n else None
    st = lib.stream_ptr()
    e0 = K.prof_begin()
    s_ptr = stats.data_ptr()
    lib.call('mscl_conv3d_fwd_groups', dref, x.data_ptr(), rt['w'].data_ptr(), y.data_ptr(),
             rt['bias'].data_ptr() if rt['bias'] is not None else None, None, 0, s_ptr, s_ptr + 4 * C, G,
             ws.data_ptr() if ws is not None else None, ws_n, st)
    K.prof_end(e0, 'fwd', d)
    bp = bn._bnp
    if bp is None:
        brt = bn._rt
        bp = bn._bnp = lib.BnParams(None, None, brt['gamma'].data_ptr(), brt['beta'].data_ptr(), bn.running_mean.data_ptr(),
                                    bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr(), None, None)
        bn._bnp_ref = ctypes.byref(bp)
    sv = save.data_ptr()
    bp.sum, bp.sumsq, bp.save_mean, bp.save_invstd = s_ptr, s_ptr + 4 * C, sv, sv + 4 * G * C
    lib.call('mscl_bn_act_fwd_groups', y.data_ptr(), bn._bnp_ref, residual.data_ptr() if residual is not None else None, None,
             out.data_ptr(), y.numel() // C, C, bn.eps, bn.momentum, int(relu), G, st)
    return y, out, save


def cba_eval(conv, bn, x, residual, relu):
    """cba_fwd in evaluation mode (module.eval()): BatchNorm normalises with its running statistics, nothing is saved or
    updated.  ref: nn.BatchNorm3d inference branch as used by Recognizer3D._do_test (recognizers/recognizer3d.py:33-96)."""
    rt = conv._rt
    if rt is None:
        raise MsclError('model not materialized on a GPU: call model.materialize(device) first')
    y = K.conv3d_fwd(x, rt['w'], conv.desc(x.shape), bias=rt['bias'])
    out = torch.empty_like(y)
    C = conv.out_channels
    brt = bn._rt
    bp = lib.BnParams(None, None, brt['gamma'].data_ptr(), brt['beta'].data_ptr(), bn.running_mean.data_ptr(),
                      bn.running_var.data_ptr(), None, None, None)
    lib.call('mscl_bn_act_fwd', y.data_ptr(), ctypes.byref(bp), residual.data_ptr() if residual is not None else None, None,
             out.data_ptr(), y.numel() // C, C, bn.eps, bn.momentum, int(relu), lib.stream_ptr())
    return out


def cba_bwd(conv, bn, dout, out, y, save, x, relu, need_dx, want_dres=False, dx_addend=None, dx_into=None):
    """backward of cba_fwd: BN(+ReLU) input gradient, conv weight gradient (into the arena), conv input
    gradient (optionally fused with `dx_addend`, or accumulated into the existing map `dx_into`).  Returns (dx|None, dres|None).
    (Rounds 1-3 carried an opt-in fused form -- the layer-1 input-gradient kernel reducing the consuming BatchNorm's sums in its
    epilogue -- that broke even at best; removed in round 4, see conv_halo.hip.)"""
    rt = bn._rt
    C = conv.out_channels
    G = save.shape[1] if save.dim() == 3 else 1                # statistics groups of the forward pass
    scratch = K.ZEROS.take(G * K.STAT_SLOTS * 4 * C, dout.device)
    dy, dres = K.bn_act_bwd(dout, out, y, rt['gamma'], save[0], save[1], rt['dgamma'], rt['dbeta'], relu, scratch,
                            want_identity_dres=want_dres, beta=rt['beta'] if (relu and not want_dres and y.numel() >= MASK_FROM_Y_MIN) else None,
                            groups=G)
    rt['slot_g'].touched = True
    rt['slot_b'].touched = True
    _wgrad(conv, x, dy)
    dx = conv.dgrad(dy, x.shape, addend=dx_addend, out=dx_into) if need_dx else None
    return dx, dres


# Weight gradients are leaves of the backward dependency chain (only the input gradient feeds the next layer), so
PAIR_STEM = True            # RGB stem on W-paired input (kernels.pair_w): K 1176 -> 672 (round 1: forward 126 -> 81 us; the plain stem stays testable via Conv3dHip(pair_w=False))
MASK_FROM_Y_MIN = 1 << 24   # BN backward recomputes the ReLU mask from y on maps this large (72 vs 78 us on layer 1; smaller maps lose)


class _BackwardEnd:
    """Hooks that run once when the running backward pass ends (the deferred weight gradients), through the autograd engine's
    callback queue.  A hook that raises does not keep the others from running -- the first error is raised after the last.
    The engine callback is queued on every add (it is per pass and dropped with a pass that dies; a flag kept here would go stale
    and the next pass's hooks would never run): the first one to fire runs the hooks, the rest find none."""
    hooks = []

    @staticmethod
    def add(fn):
        """RuntimeError when no backward pass is running (the engine refuses the callback)"""
        torch.autograd.Variable._execution_engine.queue_callback(_BackwardEnd.run)
        if fn not in _BackwardEnd.hooks:
            _BackwardEnd.hooks.append(fn)

    @staticmethod
    def run():
        hooks, err = list(_BackwardEnd.hooks), None
        del _BackwardEnd.hooks[:]
        for fn in hooks:
            try:
                fn()
            except Exception as e:          # noqa: BLE001 -- re-raised below
                err = err or e
        if err is not None:
            raise err


class WGradQueue:
    """Deferred weight gradients of the small layers, launched as ONE grouped kernel (mscl_conv3d_wgrad_group, csrc/conv_wgrad.hip).

    A weight gradient is a leaf of the backward chain.  On the small maps (layers 3-4, their entries and shortcuts, the pyramid
    levels) it is a launch-latency-bound kernel -- 8-41 us at 6 % MFMA busy, ~30 of them and as many bias column sums per step,
    most on the RGB query chain with the chip nearly idle -- and streams cannot hide it (every fork edge inside the captured step
    costs more than it hides: profiles/r05_ab_sweeps.md).  So the launches are deferred instead: _wgrad() queues (conv, x, dy), which
    keeps the operands alive, and the queue of a stream goes out on that stream as one grouped launch (+ one for the bias
    gradients) when it is full, when a gradient bucket is about to be reduced, and -- through the autograd engine's
    end-of-backward callback -- before .backward() returns.  One queue per stream: the flow trunk's backward runs on the flow stream."""

    def __init__(self):
        self.queues = {}            # raw stream handle -> (torch stream, [(conv, x, dy, desc)])

    def add(self, conv, x, dy, d):
        sp = lib.stream_ptr()
        q = self.queues.get(sp)
        if q is None:
            q = self.queues[sp] = (torch.cuda.current_stream(), [])
        q[1].append((conv, x, dy, d))
        try:
            _BackwardEnd.add(self.flush_all)
        except RuntimeError:                # not inside a backward pass (a kernel-level caller): nothing would flush the queue later
            self._flush(q[1])
            return
        if len(q[1]) >= lib.WGRAD_GROUP_MAX:
            self._flush(q[1])

    def flush(self):
        """the current stream's queue, now (before a gradient bucket's all-reduce)"""
        q = self.queues.get(lib.stream_ptr())
        if q is not None and q[1]:
            self._flush(q[1])

    def clear(self):
        """drop everything queued and the pending end-of-backward hook, launching nothing.  For the paths on which a backward pass
        did NOT reach its end-of-backward callbacks (the engine skips them when backward raises: an aborted capture, an
        out-of-memory error the caller catches): the queued (conv, x, dy) items would otherwise be launched by the NEXT backward's
        callback, reading operands of a pass that no longer exists (under an aborted capture: the released graph pool)."""
        n = sum(len(items) for _, items in self.queues.values())
        for _, items in self.queues.values():
            del items[:]
        self.queues.clear()
        del _BackwardEnd.hooks[:]
        return n

    def pending(self):
        return sum(len(items) for _, items in self.queues.values())

    def flush_all(self):
        """every stream's queue, each on its own stream (the end of a backward pass; the step's sync_streams joins the streams)"""
        for st, items in self.queues.values():
            if items:
                with torch.cuda.stream(st):
                    self._flush(items)

    @staticmethod
    def _flush(items):
        n = len(items)
        descs = (lib.ConvDesc * n)()
        xs, dys, dws, dbs = ((ctypes.c_void_p * n)() for _ in range(4))
        gf = []
        for i, (conv, x, dy, d) in enumerate(items):
            ctypes.memmove(ctypes.byref(descs[i]), ctypes.byref(d), ctypes.sizeof(lib.ConvDesc))
            rt = conv._rt
            xs[i], dys[i], dws[i] = x.data_ptr(), dy.data_ptr(), rt['dw'].data_ptr()
            dbs[i] = rt['dbias'].data_ptr() if rt['dbias'] is not None else None
            rt['slot_w'].touched = True
            if rt['slot_b'] is not None:
                rt['slot_b'].touched = True
            gf.append(2.0 * d.N * d.To * d.Ho * d.Wo * d.K * d.kT * d.kH * d.kW * d.C)
        e0 = K.prof_begin()
        lib.call('mscl_conv3d_wgrad_group', n, descs, xs, dys, dws, dbs, lib.stream_ptr())
        if e0 is not None:          # bench.py's stage table: the group's time split over its layers by their FLOPs
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            tot = sum(gf)
            for (conv, x, dy, d), f in zip(items, gf):
                K.PROFILE['events'].append(('wgrad', tuple(getattr(d, k) for k in K._DESC_FIELDS), e0, e1, f / tot))
        del items[:]                # (drops the references that kept x / dy alive)


WGRADS = WGradQueue()
GROUP_MAX_ROWS = 16384              # output positions up to which a layer's weight gradient is deferred into a grouped launch
SHORTCUT_INTO_DX = [True]          # False: a strided 1x1x1 shortcut's input gradient is a map of its own, added by the entry conv's (A/B, tools/ab_step.py)
GROUP_WGRADS = [True]               # False: every weight gradient is launched where it arises (A/B, tools/ab_step.py)


def _wgrad(conv, x, dy):
    """(Round 4 measured the narrowest form of all: only the 10-us slab sums of the window-resident weight gradients -- a leaf of
    the chain, HBM-bound -- launched on the idle key stream behind an event, with persistent per-layer workspaces and the join
    in sync_streams: 981-987 vs 1090-1095 clip-pairs/s in three alternating pairs.  Whatever forks off the RGB query chain during
    backward costs more than it hides; the mechanism was removed again.)
    (weight gradients on a side stream, off the dgrad / BatchNorm-backward chain, were measured in rounds 1 and 2: -6 ... -13 %
    on the step -- two MFMA-heavy kernels side by side do not pay -- and removed in round 3.  Round 3 re-measured the narrowest
    form: only the layer-1 / stem weight gradients of the RGB query trunk, issued AFTER their layer's input gradient on the idle
    key stream so that they would run beside the next BatchNorm-backward passes: 984-988 vs 1043-1048 clip-pairs/s; with layer 2 and
    the flow trunk's as well 922-937.
    Round 5 measured the opposite selection: only the weight gradients of the SMALL maps (<= 16384 / <= 8192 output positions:
    layers 3-4, their entries and shortcuts, the pyramid levels, the flow trunk's last layer -- latency-bound kernels at 6 % MFMA
    busy, ~0.5 ms of the RGB query chain) on ONE extra stream, in issue order, joined once before the optimizer: 1040 / 1042 vs 1129
    clip-pairs/s (two captured graphs replayed alternately in one process, tools/ab_step.py).  Not the kernels but the fork costs:
    every cross-stream edge inside the captured step is a barrier packet and a signal between hardware queues, ~30 of them per step
    here, and each delays the chain it leaves.  Concurrency for small launches has to come from ONE launch (a grouped kernel), not
    from more streams: WGradQueue.)"""
    if GROUP_WGRADS[0]:
        key = ('wg', tuple(x.shape), lib.DET_GEN, GROUP_MAX_ROWS)
        ok = conv._plans.get(key)
        d = conv.desc(x.shape)
        if ok is None:
            # small maps only: a large map's weight gradient fills the chip by itself, and deferring it would keep its dy alive
            ok = conv._plans[key] = (d.N * d.To * d.Ho * d.Wo <= GROUP_MAX_ROWS and conv._rt.get('dw8_flush') is None
                                     and bool(lib.call_raw('mscl_conv3d_wgrad_groupable', ctypes.byref(d))))
        if ok:
            WGRADS.add(conv, x, dy, d)
            return
    conv.wgrad(x, dy)


HOLD_BUCKETS = [False]          # True while a branch's backward is being captured into a sub-graph (recognizers.QueryGraph)


def _bucket_done(mod):
    """data-parallel hook: this module's backward completes a gradient bucket -> start its all-reduce now"""
    buckets = getattr(mod, '_grad_buckets', ())
    if buckets and (HOLD_BUCKETS[0] or not parallel.single()):
        WGRADS.flush()          # the bucket's deferred weight gradients go out before its all-reduce (and inside a sub-graph capture)
    if HOLD_BUCKETS[0]:
        return                  # nothing executes during capture; the replaying node fires the trigger itself
    for red, i in buckets:
        red.bucket_done(i)


class BucketCounter:
    """Order-independent trigger for a gradient bucket whose modules are applied several times per step in an order autograd
    does not promise (the FPN / SEPC convs shared across pyramid levels, the projection MLP): every forward application made
    under autograd counts up, every backward counts down, and the bucket's all-reduce starts when the count returns to zero --
    i.e. when the last of this step's applications has written its weight gradient.  An application whose output nothing reads
    never runs backward: the count then stays positive and GradReducer.finish() sends the bucket (and resets the count)."""

    def __init__(self, reducer, idx):
        self.red, self.idx, self.pending = reducer, idx, 0
        reducer.counters.append(self)

    def fwd(self):
        if torch.is_grad_enabled():
            self.pending += 1
            return True
        return False

    def bwd(self):
        self.pending -= 1
        if self.pending == 0 and not HOLD_BUCKETS[0]:
            if not parallel.single():
                WGRADS.flush()
            self.red.bucket_done(self.idx)


def stem_input(conv, x):
    """the packed clip as the stem's kernels read it: pixel pairs along W for a pair_w stem (one 3-us pass)"""
    return K.pair_w(x) if conv.pair_w else x


def _cb(mod):
    """(conv, bn) of a conv+BN unit under either naming: nn.Sequential (torchvision / fastonly: '0', '1') or mmcv's
    ConvModule ('conv', 'bn')"""
    return (mod[0], mod[1]) if isinstance(mod, nn.Sequential) else (mod.conv, mod.bn)


class _StemFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, stem):
        conv, bn = _cb(stem)
        x = stem_input(conv, x)
        y, out, save = cba_fwd(conv, bn, x, None, True)
        ctx.stem = stem
        ctx.save_for_backward(x, y, out, save)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, y, out, save = ctx.saved_tensors
        conv, bn = _cb(ctx.stem)
        cba_bwd(conv, bn, dout.contiguous(), out, y, save, x, True, need_dx=False)
        fl = conv._rt.get('dw8_flush')
        if fl is not None:                      # 3-channel stem: fold the 8-channel staging gradient into the arena now,
            dw8, gview, cin = fl                # so the bucket all-reduce launched below sees it
            if conv.pair_w:
                K.pair_w_grad_fold(dw8, gview)
            else:
                gview.add_(dw8[..., :cin])
            dw8.zero_()
        _bucket_done(ctx.stem)
        return None, None, None


class _BlockFn(torch.autograd.Function):
    """One BasicBlock as a single graph node (ref: r3d.py:95-127 / fastonly.py:104-136):
    out = relu(bn2(conv2(relu(bn1(conv1 x)))) + shortcut(x))."""

    @staticmethod
    def forward(ctx, x, block):
        c1, b1 = block.conv1[0], block.conv1[1]
        c2, b2 = block.conv2[0], block.conv2[1]
        y1, a1, s1 = cba_fwd(c1, b1, x, None, True)
        if block.downsample is not None:
            yd, ad, sd = cba_fwd(block.downsample[0], block.downsample[1], x, None, False)
            res = ad
        else:
            yd = sd = None
            res = x
        y2, out, s2 = cba_fwd(c2, b2, a1, res, True)
        ctx.block = block
        ctx.has_ds = yd is not None
        if ctx.has_ds:
            ctx.save_for_backward(x, y1, a1, s1, y2, out, s2, yd, sd)
        else:
            ctx.save_for_backward(x, y1, a1, s1, y2, out, s2)
        return out

    @staticmethod
    def backward(ctx, dout):
        block = ctx.block
        if ctx.has_ds:
            x, y1, a1, s1, y2, out, s2, yd, sd = ctx.saved_tensors
        else:
            x, y1, a1, s1, y2, out, s2 = ctx.saved_tensors
        c1, b1 = block.conv1[0], block.conv1[1]
        c2, b2 = block.conv2[0], block.conv2[1]
        da1, dz = cba_bwd(c2, b2, dout.contiguous(), out, y2, s2, a1, True, need_dx=True, want_dres=True)
        if ctx.has_ds and SHORTCUT_INTO_DX[0] and block.downsample[0].strided_pointwise():
            # the strided 1x1x1 shortcut reaches one input position in eight (four): the entry conv's gradient first, the shortcut's
            # added into it at those positions -- not a map of mostly zeros written here and read back as the entry's addend
            dx, _ = cba_bwd(c1, b1, da1, a1, y1, s1, x, True, need_dx=True)
            cba_bwd(block.downsample[0], block.downsample[1], dz, None, yd, sd, x, False, need_dx=True, dx_into=dx)
        else:
            if ctx.has_ds:
                shortcut_grad, _ = cba_bwd(block.downsample[0], block.downsample[1], dz, None, yd, sd, x, False, need_dx=True)
            else:
                shortcut_grad = dz
            dx, _ = cba_bwd(c1, b1, da1, a1, y1, s1, x, True, need_dx=True, dx_addend=shortcut_grad)
        _bucket_done(block)
        return dx, None


class BasicBlockHip(nn.Module):
    def __init__(self, cin, cout, kernel, stride, pad, shortcut_stride=None):
        super().__init__()
        self.conv1 = nn.Sequential(Conv3dHip(cin, cout, kernel, stride, pad), BatchNorm3dHip(cout), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(Conv3dHip(cout, cout, kernel, 1, pad), BatchNorm3dHip(cout))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if shortcut_stride is not None:
            self.downsample = nn.Sequential(Conv3dHip(cin, cout, 1, shortcut_stride, 0), BatchNorm3dHip(cout))

    def forward(self, x):
        return _BlockFn.apply(x, self)


class VideoResNetHip(nn.Module):
    """R3D-18 ('rgb') / r2d_18 ('flow') trunks on HIP kernels, returning the four stage maps (NDHWC bf16).
    ref: torchvision r3d_18 == mmaction/models/backbones/r3d.py:176-184,216-296;
         flow: mmaction/models/backbones/fastonly.py:185-193,238-326,399-408;
         multi-level forward: mmaction/models/recognizers/moco.py:12-24.
    Input: packed clip (N,T,H,W,8) bf16 from kernels.pack_input."""

    def __init__(self, kind):
        super().__init__()
        self.kind = kind
        if kind == 'rgb':
            base, kernel, pad = 64, (3, 3, 3), (1, 1, 1)
            self.stem = nn.Sequential(Conv3dHip(3, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3), pair_w=PAIR_STEM), BatchNorm3dHip(64),
                                      nn.ReLU(inplace=True))
            st = lambda s: (s, s, s)
        elif kind == 'flow':
            base, kernel, pad = 16, (1, 3, 3), (0, 1, 1)
            self.stem = nn.Sequential(Conv3dHip(3, 16, (1, 7, 7), (2, 2, 2), (0, 3, 3)), BatchNorm3dHip(16), nn.ReLU(inplace=True))
            st = lambda s: (1, s, s)
        else:
            raise ValueError(kind)
        cin = base
        for li, mult in enumerate((1, 2, 4, 8), start=1):
            cout, s = base * mult, (1 if li == 1 else 2)
            first = BasicBlockHip(cin, cout, kernel, st(s), pad, shortcut_stride=st(s) if (s != 1 or cin != cout) else None)
            setattr(self, f'layer{li}', nn.Sequential(first, BasicBlockHip(cout, cout, kernel, 1, pad)))
            cin = cout
        self.avgpool = nn.AdaptiveAvgPool3d((1, 1, 1))
        self.fc = nn.Identity()           # base_moco.py:90-91,100-101 disables the classifier
        self._anchor = None
        self.reset_parameters()

    def reset_parameters(self):
        """ref: r3d.py:298-310 / fastonly.py:312-326: kaiming_normal(fan_out, relu) convs, BN 1/0."""
        for m in self.modules():
            if isinstance(m, Conv3dHip):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, BatchNorm3dHip):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    @torch.no_grad()
    def forward_eval(self, x):
        """the same trunk with BatchNorm in evaluation mode (running statistics); no autograd graph"""
        x = cba_eval(self.stem[0], self.stem[1], stem_input(self.stem[0], x), None, True)
        outs = []
        for li in range(1, 5):
            for blk in getattr(self, f'layer{li}'):
                a1 = cba_eval(blk.conv1[0], blk.conv1[1], x, None, True)
                res = x if blk.downsample is None else cba_eval(blk.downsample[0], blk.downsample[1], x, None, False)
                x = cba_eval(blk.conv2[0], blk.conv2[1], a1, res, True)
            outs.append(x)
        return outs

    def forward(self, x, bn_groups=1):
        """bn_groups > 1: the batch holds the inputs of that many consecutive calls of the reference's module (equal shares
        along N, in call order); every BatchNorm keeps one set of batch statistics per call and updates its running
        statistics once per call, in order -- the results of the separate calls, with each conv kernel launched once."""
        _need_gpu(x)
        if not self.training:
            return self.forward_eval(x)        # running statistics: the groups of a batch are normalised alike
        if x.shape[0] % bn_groups:
            raise ValueError(f'batch of {x.shape[0]} does not split into {bn_groups} BatchNorm groups')
        if self._anchor is None or self._anchor.device != x.device:
            self._anchor = torch.zeros(1, device=x.device, requires_grad=True)
        BN_GROUPS[0] = bn_groups
        try:
            x = _StemFn.apply(x, self._anchor, self.stem)
            outs = []
            for li in range(1, 5):
                for blk in getattr(self, f'layer{li}'):
                    x = blk(x)
                outs.append(x)
        finally:
            BN_GROUPS[0] = 1
        return outs


# ------------------------------------------------------------------------------------------ Bottleneck trunks (mscl_r50)
class ConvModuleBN(nn.Module):
    """mmcv ConvModule with norm_cfg=BN3d: bias-free conv under `.conv`, BatchNorm3d under `.bn` (the ReLU, when present, has no
    parameters).  Call sites: backbones/resnet3d.py:262-296,448-459, resnet3d_slowfast.py:157-166."""

    def __init__(self, cin, cout, kernel, stride, pad, pair_w=False):
        super().__init__()
        self.conv = Conv3dHip(cin, cout, kernel, stride, pad, pair_w=pair_w)
        self.bn = BatchNorm3dHip(cout)


class _MaxPoolFn(torch.autograd.Function):
    """nn.MaxPool3d((1,3,3), (1,2,2), (0,1,1)) on an NDHWC map (resnet3d.py:461-467, fastonly.py:229-230)"""

    @staticmethod
    def forward(ctx, x):
        out, win = K.maxpool_hw_fwd(x)
        ctx.shape = tuple(x.shape)
        ctx.save_for_backward(win)
        return out

    @staticmethod
    def backward(ctx, dout):
        return K.maxpool_hw_bwd(dout.contiguous(), ctx.saved_tensors[0], ctx.shape)


class _BottleneckFn(torch.autograd.Function):
    """One Bottleneck as a single graph node:
    out = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1 x))))))) + shortcut(x)), the stride on conv2
    (ref: backbones/resnet3d.py:298-330 Bottleneck3d.forward, backbones/fastonly.py:165-183)."""

    @staticmethod
    def forward(ctx, x, block):
        (c1, b1), (c2, b2), (c3, b3) = _cb(block.conv1), _cb(block.conv2), _cb(block.conv3)
        y1, a1, s1 = cba_fwd(c1, b1, x, None, True)
        y2, a2, s2 = cba_fwd(c2, b2, a1, None, True)
        if block.downsample is not None:
            cd, bd = _cb(block.downsample)
            yd, ad, sd = cba_fwd(cd, bd, x, None, False)
            res = ad
        else:
            yd = sd = None
            res = x
        y3, out, s3 = cba_fwd(c3, b3, a2, res, True)
        ctx.block = block
        ctx.has_ds = yd is not None
        saved = (x, y1, a1, s1, y2, a2, s2, y3, out, s3)
        ctx.save_for_backward(*(saved + ((yd, sd) if ctx.has_ds else ())))
        return out

    @staticmethod
    def backward(ctx, dout):
        block = ctx.block
        t = ctx.saved_tensors
        x, y1, a1, s1, y2, a2, s2, y3, out, s3 = t[:10]
        (c1, b1), (c2, b2), (c3, b3) = _cb(block.conv1), _cb(block.conv2), _cb(block.conv3)
        da2, dz = cba_bwd(c3, b3, dout.contiguous(), out, y3, s3, a2, True, need_dx=True, want_dres=True)
        cd, bd = _cb(block.downsample) if ctx.has_ds else (None, None)
        late = ctx.has_ds and SHORTCUT_INTO_DX[0] and cd.strided_pointwise()        # (see _BlockFn.backward: the strided shortcut's gradient goes INTO dx)
        if ctx.has_ds and not late:
            shortcut_grad, _ = cba_bwd(cd, bd, dz, None, t[10], t[11], x, False, need_dx=True)
        else:
            shortcut_grad = None if late else dz
        da1, _ = cba_bwd(c2, b2, da2, a2, y2, s2, a1, True, need_dx=True)
        dx, _ = cba_bwd(c1, b1, da1, a1, y1, s1, x, True, need_dx=True, dx_addend=shortcut_grad)
        if late:
            cba_bwd(cd, bd, dz, None, t[10], t[11], x, False, need_dx=True, dx_into=dx)
        _bucket_done(block)
        return dx, None


class BottleneckHip(nn.Module):
    """flavour 'mmcv': mmaction Bottleneck3d (backbones/resnet3d.py:162-330): conv1 3x1x1 when the block is inflated ('3x1x1'
    style) else 1x1x1, conv2 1x3x3 with the spatial stride ('pytorch' style), conv3 1x1x1; sub-modules named conv / bn.
    flavour 'tv': the flow trunk's Bottleneck (backbones/fastonly.py:137-183): conv1 1x1x1, conv2 Conv3DNoTemporal
    (fastonly.py:61-80), sub-modules nn.Sequential (names 0 / 1)."""

    def __init__(self, cin, planes, stride, downsample, flavour, inflate=False):
        super().__init__()
        cout = 4 * planes
        s3 = (1, stride, stride)
        if flavour == 'mmcv':
            k1, p1 = ((3, 1, 1), (1, 0, 0)) if inflate else ((1, 1, 1), (0, 0, 0))
            unit = lambda ci, co, k, s, p, relu: ConvModuleBN(ci, co, k, s, p)
        else:
            k1, p1 = (1, 1, 1), (0, 0, 0)

            def unit(ci, co, k, s, p, relu):
                mods = [Conv3dHip(ci, co, k, s, p), BatchNorm3dHip(co)]
                return nn.Sequential(*(mods + ([nn.ReLU(inplace=True)] if relu else [])))
        self.conv1 = unit(cin, planes, k1, 1, p1, True)
        self.conv2 = unit(planes, planes, (1, 3, 3), s3, (0, 1, 1), True)
        self.conv3 = unit(planes, cout, 1, 1, 0, False)
        self.downsample = unit(cin, cout, 1, s3, 0, False) if downsample else None
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return _BottleneckFn.apply(x, self)


def _bottleneck_stage(cin, planes, blocks, stride, flavour, inflate):
    units = [BottleneckHip(cin, planes, stride, stride != 1 or cin != 4 * planes, flavour, inflate)]
    units += [BottleneckHip(4 * planes, planes, 1, False, flavour, inflate) for _ in range(blocks - 1)]
    return nn.Sequential(*units)


class _BottleneckTrunk(nn.Module):
    """shared forward of the two Bottleneck trunks: stem conv+BN+ReLU, max-pool (1,3,3)/(1,2,2), four stages; returns the four
    stage maps (out_indices (0,1,2,3) / the patched multi-level forward of recognizers/moco.py:12-24)"""

    def _stem(self):
        raise NotImplementedError

    @torch.no_grad()
    def forward_eval(self, x):
        """the same trunk with BatchNorm in evaluation mode (running statistics), no autograd graph: what the reference's
        evaluation pass runs after model.eval() (core/evaluation/eval_hooks.py:471-487; mscl_r50 ships evaluation=dict(interval=5))"""
        conv, bn = _cb(self._stem())
        x, _ = K.maxpool_hw_fwd(cba_eval(conv, bn, stem_input(conv, x), None, True))
        outs = []
        for li in range(1, 5):
            for blk in getattr(self, f'layer{li}'):
                (c1, b1), (c2, b2), (c3, b3) = _cb(blk.conv1), _cb(blk.conv2), _cb(blk.conv3)
                a2 = cba_eval(c2, b2, cba_eval(c1, b1, x, None, True), None, True)
                res = x if blk.downsample is None else cba_eval(*_cb(blk.downsample), x, None, False)
                x = cba_eval(c3, b3, a2, res, True)
            outs.append(x)
        return outs

    def forward(self, x, bn_groups=1):
        _need_gpu(x)
        if not self.training:
            return self.forward_eval(x)        # running statistics: the groups of a batch are normalised alike
        if x.shape[0] % bn_groups:
            raise ValueError(f'batch of {x.shape[0]} does not split into {bn_groups} BatchNorm groups')
        if self._anchor is None or self._anchor.device != x.device:
            self._anchor = torch.zeros(1, device=x.device, requires_grad=True)
        BN_GROUPS[0] = bn_groups               # see VideoResNetHip.forward
        try:
            x = _MaxPoolFn.apply(_StemFn.apply(x, self._anchor, self._stem()))
            outs = []
            for li in range(1, 5):
                for blk in getattr(self, f'layer{li}'):
                    x = blk(x)
                outs.append(x)
        finally:
            BN_GROUPS[0] = 1
        return outs


class ResNet3dSlowOnlyHip(_BottleneckTrunk):
    """ResNet3dSlowOnly depth 50 as configs/recognition/moco/mscl_r50_cosm_lr3e-2.py:16-26 builds it.
    ref: backbones/resnet3d_slowonly.py:15-52 (inflate (0,0,1,1), no pool2, lateral off), backbones/resnet3d.py:448-467 (stem
    conv (5,7,7) / (2,2,2) / (2,3,3) + BN + ReLU, max-pool (1,3,3) / (1,2,2) / (0,1,1)), :407-415 (depth 50 = Bottleneck3d x
    (3,4,6,3)), resnet3d_slowfast.py:89-204 (make_res_layer), resnet3d.py:795-831 (init), :845-860 (forward).
    Input: packed clip (N,T,H,W,8) bf16 from kernels.pack_input."""

    def __init__(self, depth=50, pretrained=None, pretrained2d=False, lateral=False, num_stages=4, conv1_kernel=(5, 7, 7),
                 conv1_stride_t=2, pool1_stride_t=1, spatial_strides=(1, 2, 2, 2), out_indices=(0, 1, 2, 3),
                 inflate=(0, 0, 1, 1), zero_init_residual=True, **kwargs):
        super().__init__()
        if depth != 50 or lateral or pretrained is not None or pretrained2d or num_stages != 4 or kwargs:
            raise NotImplementedError('ResNet3dSlowOnly is built for the mscl_r50 configuration (depth 50, from scratch, no lateral)')
        if tuple(spatial_strides) != (1, 2, 2, 2) or tuple(out_indices) != (0, 1, 2, 3) or pool1_stride_t != 1:
            raise NotImplementedError('spatial_strides (1,2,2,2), out_indices (0,1,2,3), pool1_stride_t 1 (mscl_r50_cosm_lr3e-2.py:16-26)')
        k = _triple(conv1_kernel)
        self.conv1 = ConvModuleBN(3, 64, k, (conv1_stride_t, 2, 2), tuple((i - 1) // 2 for i in k), pair_w=PAIR_STEM and k[2] == 7)
        self.maxpool = nn.MaxPool3d((1, 3, 3), (pool1_stride_t, 2, 2), (0, 1, 1))
        self.zero_init_residual = zero_init_residual
        cin = 64
        for li, (planes, blocks, stride, inf) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3), spatial_strides, inflate), 1):
            setattr(self, f'layer{li}', _bottleneck_stage(cin, planes, blocks, stride, 'mmcv', bool(inf)))
            cin = 4 * planes
        self._anchor = None
        self.init_weights()

    def _stem(self):
        return self.conv1

    @property
    def stem(self):              # the name the gradient-bucket plan and the key/query sub-graphs use for the first unit
        return self.conv1

    def init_weights(self, pretrained=None):
        """ref: resnet3d.py:819-831: kaiming_init convs (normal, fan_out, relu), BN 1 / 0, conv3.bn weight 0"""
        for m in self.modules():
            if isinstance(m, Conv3dHip):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, BatchNorm3dHip):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, BottleneckHip):
                    nn.init.constant_(m.conv3.bn.weight, 0)


class FlowR2D50Hip(_BottleneckTrunk):
    """resnet_flow.r2d_50: fastonly.py:431-441 (Bottleneck x (3,4,6,3), Conv3DNoTemporal), :238-262 (inplanes 8), :222-235
    (BottleneckStem: conv (1,7,7) / (2,2,2) / (0,3,3) to 8 channels + BN + ReLU + max-pool), :291-326 (_make_layer, init)."""

    def __init__(self):
        super().__init__()
        self.stem = nn.Sequential(Conv3dHip(3, 8, (1, 7, 7), (2, 2, 2), (0, 3, 3)), BatchNorm3dHip(8), nn.ReLU(inplace=True),
                                  nn.MaxPool3d((1, 3, 3), (1, 2, 2), (0, 1, 1)))
        cin = 8
        for li, (planes, blocks, stride) in enumerate(((8, 3, 1), (16, 4, 2), (32, 6, 2), (64, 3, 2)), 1):
            setattr(self, f'layer{li}', _bottleneck_stage(cin, planes, blocks, stride, 'tv', False))
            cin = 4 * planes
        self.avgpool = nn.AdaptiveAvgPool3d((1, 1, 1))
        self.fc = nn.Identity()
        self._anchor = None
        for m in self.modules():
            if isinstance(m, Conv3dHip):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, BatchNorm3dHip):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _stem(self):
        return self.stem


# ------------------------------------------------------------------------------------------ neck pieces
class _ConvBiasFn(torch.autograd.Function):
    """out = relu?(conv(x) + bias + addend)"""

    @staticmethod
    def forward(ctx, x, addend, conv, relu, counted):
        out = conv.fwd(x, addend=addend, relu=relu)
        ctx.conv, ctx.relu, ctx.has_add, ctx.counted = conv, relu, addend is not None, counted
        ctx.save_for_backward(x, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, out = ctx.saved_tensors
        dz = K.relu_bwd(dout.contiguous(), out) if ctx.relu else dout.contiguous()
        _wgrad(ctx.conv, x, dz)
        if ctx.counted:
            ctx.conv._bucket_counter.bwd()           # data-parallel: the last application of the bucket starts its all-reduce
        dx = ctx.conv.dgrad(dz, x.shape) if ctx.needs_input_grad[0] else None
        return dx, (dz if ctx.has_add else None), None, None, None


def conv_bias(conv, x, addend=None, relu=False):
    c = getattr(conv, '_bucket_counter', None)
    counted = c is not None and (x.requires_grad or (addend is not None and addend.requires_grad)) and c.fwd()
    return _ConvBiasFn.apply(x, addend, conv, relu, counted)


class _UpsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, size, trilinear):
        dst = torch.empty((src.shape[0], *size, src.shape[-1]), dtype=torch.bfloat16, device=src.device)
        K.upsample_add(src, dst, trilinear, accumulate=False)
        ctx.src_shape, ctx.trilinear = tuple(src.shape), trilinear
        return dst

    @staticmethod
    def backward(ctx, ddst):
        return K.upsample_bwd(ddst.contiguous(), ctx.src_shape, ctx.trilinear), None, None


def upsample(src, size, trilinear):
    return _UpsampleFn.apply(src, tuple(size), trilinear)


class _PoolFn(torch.autograd.Function):
    """mean over the middle axis of (outer, inner, C) bf16 -> (outer, C) fp32"""

    @staticmethod
    def forward(ctx, x, outer, inner):
        ctx.shape, ctx.outer, ctx.inner = tuple(x.shape), outer, inner
        return K.pool_fwd(x, outer, inner)

    @staticmethod
    def backward(ctx, dout):
        return K.pool_bwd(dout.contiguous(), ctx.shape, ctx.outer, ctx.inner), None, None


def pool(x, outer, inner):
    return _PoolFn.apply(x, outer, inner)


class _MlpHeadFn(torch.autograd.Function):
    """q = normalize(Linear(ReLU(Linear(emb)))).  ref: recognizers/moco.py:367-372,528-529."""

    @staticmethod
    def forward(ctx, emb, mlp, counted):
        l1, l2 = mlp[0], mlp[2]
        h = K.linear_fwd(emb, l1._rt['w'], l1._rt['b'], True)
        z = K.linear_fwd(h, l2._rt['w'], l2._rt['b'], False)
        q, norms = K.l2norm_fwd(z)
        ctx.mlp, ctx.counted = mlp, counted
        ctx.save_for_backward(emb, h, z, q, norms)
        return q

    @staticmethod
    def backward(ctx, dq):
        emb, h, z, q, norms = ctx.saved_tensors
        l1, l2 = ctx.mlp[0], ctx.mlp[2]
        dz = K.l2norm_bwd(q, norms, dq.contiguous())
        dh = K.linear_bwd(h, l2._rt['w'], z, dz, l2._rt['dw'], l2._rt['db'], False)
        demb = K.linear_bwd(emb, l1._rt['w'], h, dh, l1._rt['dw'], l1._rt['db'], True)
        for l in (l1, l2):
            l._rt['slot_w'].touched = True
            l._rt['slot_b'].touched = True
        if ctx.counted:
            ctx.mlp._bucket_counter.bwd()
        return demb, None, None


def mlp_head(mlp, emb):
    c = getattr(mlp, '_bucket_counter', None)
    counted = c is not None and emb.requires_grad and c.fwd()
    return _MlpHeadFn.apply(emb, mlp, counted)


class LinearHip(nn.Module):
    """Parameter holder for nn.Linear (state-dict `weight` (out,in), `bias`)."""

    def __init__(self, in_f, out_f):
        super().__init__()
        self.in_features, self.out_features = in_f, out_f
        self.weight = nn.Parameter(torch.empty(out_f, in_f))
        self.bias = nn.Parameter(torch.empty(out_f))
        nn.Linear.reset_parameters(self)
        self._rt = None


class Conv1dK1Hip(nn.Module):
    """Parameter holder for nn.Conv1d(cin, cout, 1) (state-dict `weight` (out, in, 1), `bias`): a per-frame linear map, run by
    the fp32 linear kernels.  ref: heads/local_cl_head.py:30-33 (trans_flow), default nn.Conv1d initialisation (the head's
    init_weights is a no-op, local_cl_head.py:38-39)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        ref = nn.Conv1d(cin, cout, 1)
        self.weight = nn.Parameter(ref.weight.detach().clone())
        self.bias = nn.Parameter(ref.bias.detach().clone())
        self._rt = None
