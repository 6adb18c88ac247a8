"""MoCoV2 and MSCLWithAug on the MI355X kernels, behind the reference's registry surface.

ref: mmaction/models/recognizers/moco.py:324-547 (MoCoV2), recognizers/mscl.py:158-277 (MSCLWithAug),
recognizers/base.py:274-308 (_parse_losses), recognizers/base_moco.py:77-106 (backbone dispatch).

Step schedule (one data-parallel replica; equivalent to the reference's, re-ordered so that every
queue snapshot is streamed exactly once forward and once backward -- SURVEY.md Appendix E-2/3):
  1. pack clips NCTHW fp32 -> NDHWC bf16 (RGB normalised in the same pass)
  2. RGB:  EMA(key enc) ; q = f_q(im_q) ; k = f_k(im_k)                     [no_grad for k]
  3. flow base, then flow rotated: EMA ; q ; k   (EMA twice, separate BN statistics, App. E-5)
  4. loss phase (one autograd node, gradients w.r.t. q / pooled maps computed eagerly):
        pass A  queue_rgb (pre-enqueue):  rows q_rgb | q_flow_base | q_flow_rot  vs k_rgb
        pass B  queue_flow (pre-enqueue): rows q_flow_base                      vs k_flow_base
        enqueue flow keys (all-gathered), pass C queue_flow (post): q_flow_rot vs k_flow_rot,
        q_rgb vs k_flow_base, q_rgb vs k_flow_rot ; enqueue rgb keys ; LMCL
  5. backward through necks / trunks (fused block nodes), gradients accumulate in the flat arena.
"""
import math
import os
from collections import OrderedDict

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import kernels as K
from . import nn as nn_hip
from . import parallel
from .arena import ParamArena
from .lib import MsclError
from .nn import (BatchNorm3dHip, Conv1dK1Hip, Conv3dHip, FlowR2D50Hip, LinearHip, ResNet3dSlowOnlyHip, VideoResNetHip, mlp_head,
                 pool)
from .registry import BACKBONES, RECOGNIZERS, build_head, build_neck, build_ssl_aug
from .staging import StagingRing

LOG_KEYS = ('top1_acc', 'top5_acc', 'loss_cls', 'top1_acc_flow', 'top5_acc_flow', 'loss_cls_flow', 'loss_cls_flow_aug',
            'top1_acc_mx', 'top5_acc_mx', 'loss_cls_mx', 'top1_acc_mx_r', 'top5_acc_mx_r', 'loss_cls_mx_r',
            'top1_acc_mx_aug', 'top5_acc_mx_aug', 'loss_cls_mx_aug', 'top1_acc_mx_r_aug', 'top5_acc_mx_r_aug',
            'loss_cls_mx_r_aug', 'loss_pos', 'top1_acc_pos', 'top5_acc_pos', 'loss')
LOG_KEYS_NO_AUG_MX = tuple(k for k in LOG_KEYS if not k.endswith('_mx_aug') and not k.endswith('_mx_r_aug'))


def momentum_at(iters, max_iters, m_base):
    """ref: moco.py:413-415."""
    factor = min(iters / max_iters, 1)
    return 1 - 0.5 * (1 - m_base) * (math.cos(math.pi * factor) + 1)


def build_backbone_by_name(cfg):
    """ref: base_moco.py:77-106: 'torchvision.<name>' and 'resnet_flow.<name>' bypass the registry."""
    cfg = dict(cfg)
    typ = cfg.pop('type')
    if typ == 'torchvision.r3d_18':
        return VideoResNetHip('rgb', **cfg)
    if typ == 'resnet_flow.r2d_18':
        return VideoResNetHip('flow', **cfg)
    if typ == 'resnet_flow.r2d_50':
        return FlowR2D50Hip(**cfg)
    if typ in BACKBONES:                       # registry names (base_moco.py:104-106): ResNet3dSlowOnly of mscl_r50
        return BACKBONES.build(dict(cfg, type=typ))
    raise NotImplementedError(f'backbone {typ} is outside the MSCL hot path (mscl_r18 / mscl_r50)')


BACKBONES.register_module(name='ResNet3dSlowOnly', module=ResNet3dSlowOnlyHip)


def bind_module(m, ar, key, rec):
    """runtime views of one parameter-holding module into the arenas (`key`: the key-encoder twin); conv stems with fewer than
    8 input channels register their padded-shadow refresh with `rec`"""
    P, Pb = ('KX', 'Kb') if key else ('Q', 'Qb')
    if isinstance(m, Conv3dHip):
        sw = m.weight._mscl_slot
        sb = m.bias._mscl_slot if m.bias is not None else None
        rt = dict(slot_w=sw, slot_b=sb, bias=ar.view(P, sb) if sb is not None else None,
                  dbias=None if key or sb is None else ar.view('G', sb), wT=None, dw=None)
        dev = ar.device
        if m.cin_eff == m.in_channels:
            rt['w'] = ar.packed(Pb, sw)
            if not key:
                rt['dw'] = ar.packed('G', sw)
        else:       # 3-channel stems: zero-padded 8-channel shadow (and padded gradient staging)
            w8 = torch.zeros((m.out_channels, *m.k_exec, 8), dtype=torch.bfloat16, device=dev)
            rt['w'] = w8
            src = ar.packed(P, sw)

            def refresh(w8=w8, src=src, cin=m.in_channels, pair=m.pair_w):
                if pair:
                    K.pair_w_weight(src, w8)
                else:
                    w8[..., :cin].copy_(src)
            (rec._k_refresh if key else rec._q_refresh).append(refresh)
            if not key:
                dw8 = torch.zeros((m.out_channels, *m.k_exec, 8), dtype=torch.float32, device=dev)
                rt['dw'] = dw8
                rt['dw8_flush'] = (dw8, ar.packed('G', sw), m.in_channels)
        if not key and m.cin_eff == m.in_channels:
            wT = torch.empty((m.in_channels, *m.kernel_size, m.out_channels), dtype=torch.bfloat16, device=dev)
            rt['wT'] = wT          # refreshed by ONE batched transpose launch (refresh_after_optimizer)
        m._rt = rt
    elif isinstance(m, BatchNorm3dHip):
        sg, sb = m.weight._mscl_slot, m.bias._mscl_slot
        m._rt = dict(gamma=ar.view(P, sg), beta=ar.view(P, sb), slot_g=sg, slot_b=sb,
                     dgamma=None if key else ar.view('G', sg), dbeta=None if key else ar.view('G', sb))
    elif isinstance(m, LinearHip):
        sw, sb = m.weight._mscl_slot, m.bias._mscl_slot
        m._rt = dict(w=ar.view(P, sw), b=ar.view(P, sb), slot_w=sw, slot_b=sb,
                     dw=None if key else ar.view('G', sw), db=None if key else ar.view('G', sb))
    elif isinstance(m, Conv1dK1Hip):       # (out, in, 1) contiguous == the (out, in) matrix of the linear kernels
        sw, sb = m.weight._mscl_slot, m.bias._mscl_slot
        shp = (m.out_channels, m.in_channels)
        m._rt = dict(w=ar.view(P, sw).view(shp), b=ar.view(P, sb), slot_w=sw, slot_b=sb,
                     dw=ar.view('G', sw).view(shp), db=ar.view('G', sb))


@RECOGNIZERS.register_module()
class MoCoV2(nn.Module):
    def __init__(self, backbone, neck, moco_head, im_key='imgs', dim_in=512, dim=128, K=65536, m_base=0.994,
                 t_decay=0.99999, max_iters=1, T=0.07, mlp=False, aux_info=[], aug=dict(type='IdentityAug'),
                 train_cfg=None, test_cfg=None):
        super().__init__()
        if not mlp:
            raise NotImplementedError('mlp=False projection is not used by the MSCL configs')
        if abs(t_decay - 0.99999) > 0:
            raise NotImplementedError('the reference hard-codes the queue age decay 0.99999 (moco.py:484)')
        self.K, self.m_base, self.m, self.T = K, m_base, m_base, T
        self.iters, self.max_iters, self.batch_size = 0, max_iters, 0
        self.im_key, self.t_decay, self.aux_info = im_key, t_decay, aux_info
        self.backbone_from = 'torchvision'
        self.encoder_q = build_backbone_by_name(backbone)
        self.encoder_k = build_backbone_by_name(backbone)
        self.neck_q, self.neck_k = build_neck(neck), build_neck(neck)
        self.moco_head = build_head(moco_head)
        self.mlp_q = nn.Sequential(LinearHip(dim_in, dim_in), nn.ReLU(), LinearHip(dim_in, dim))
        self.mlp_k = nn.Sequential(LinearHip(dim_in, dim_in), nn.ReLU(), LinearHip(dim_in, dim))
        for pq, pk in self.qk_pairs():
            pk.data.copy_(pq.data)
            pk.requires_grad = False
        self.register_buffer('queue', F.normalize(torch.randn(dim, K), dim=0))
        self.register_buffer('queue_ptr', torch.zeros(1, dtype=torch.long))
        self.register_buffer('count', torch.zeros(K, dtype=torch.long))
        self._weight = None
        self.aug_gpu = build_ssl_aug(aug)
        self._arena, self._range = None, None

    def q_modules(self):
        return (self.encoder_q, self.neck_q, self.mlp_q)

    def k_modules(self):
        return (self.encoder_k, self.neck_k, self.mlp_k)

    def qk_pairs(self):
        for mq, mk in zip(self.q_modules(), self.k_modules()):
            yield from zip(mq.parameters(), mk.parameters())

    @property
    def weight(self):
        raise MsclError('the aged queue snapshot is never materialised on the HIP path (it is streamed by the '
                        'contrastive kernels); use queue / count')

    @torch.no_grad()
    def momentum_update(self, m_dev):
        """ref: moco.py:408-421 -- one kernel over this recognizer's contiguous arena range.  The momentum
        value is read from device memory (`m_dev`, written by MSCLWithAug._pre_step_host through a pinned
        staging buffer) so that a captured HIP graph follows the cosine schedule."""
        a, b = self._range
        ar = self._arena
        K.ema_update_dev(ar.KX[a:b], ar.Q[a:b], ar.Kb[a:b], m_dev)
        for fn in self._k_refresh:
            fn()

    def encode_q(self, x, levels=None, bn_groups=1):
        """levels: the q_mlvl levels the caller reads (None = all, as moco.py:517-529 returns them).
        bn_groups: x holds the inputs of that many consecutive calls (nn.VideoResNetHip.forward)."""
        emb_maps, _ = self.neck_q(self.encoder_q(x) if bn_groups == 1 else self.encoder_q(x, bn_groups=bn_groups), levels=levels)
        emb, maps = emb_maps
        return mlp_head(self.mlp_q, emb), maps

    @torch.no_grad()
    def encode_k(self, x, levels=()):
        """key forward (moco.py:535-545).  k_mlvl has no reader on the MSCL path once the dead unshuffle all-gathers are
        dropped, so by default the key neck computes the embedding only."""
        feats = self.encoder_k(x)
        emb_maps, _ = self.neck_k(feats, levels=levels)
        emb, maps = emb_maps
        return mlp_head(self.mlp_k, emb), maps

    # ------------------------------------------------------------------ stand-alone step (single-recognizer MoCo training)
    # Inside MSCLWithAug the two recognizers share that model's arenas and step; a MoCoV2 built on its own (registry type
    # 'MoCoV2' at the top of a config) owns an arena and runs recognizers/moco.py:442-515 itself.
    def materialize(self, device='cuda'):
        device = torch.device(device)
        if device.type != 'cuda':
            raise MsclError('materialize() needs a GPU device: the HIP path has no CPU fallback')
        from .lib import load
        load()
        ar = ParamArena(device)
        plan = []
        ar.begin_group('rec')
        for mq, mk in zip(self.q_modules(), self.k_modules()):
            for (nq, pq), (nk, pk) in zip(mq.named_parameters(), mk.named_parameters()):
                assert nq == nk and pq.shape == pk.shape
                plan.append((ar.add(nq, pq.shape), pq, pk))
        ar.end_group('rec')
        ar.allocate()
        for slot, pq, pk in plan:
            vq, vk = ar.view('Q', slot), ar.view('KX', slot)
            vq.copy_(pq.data.to(device)); vk.copy_(pk.data.to(device))
            pq.data, pk.data = vq, vk
            pq.grad = ar.view('G', slot)
            pq._mscl_slot = pk._mscl_slot = slot
        for mod in self.modules():
            for bname, buf in list(mod._buffers.items()):
                if buf is not None:
                    mod._buffers[bname] = buf.to(device)
        self.arena = ar
        self._arena, self._range = ar, tuple(ar.ranges['rec'])
        self._k_refresh, self._q_refresh = [], []
        for mods, key in ((self.q_modules(), False), (self.k_modules(), True)):
            for top in mods:
                for m in top.modules():
                    bind_module(m, ar, key, self)
        entries = [(m._rt['w'], m._rt['wT'], m.out_channels, m.taps, m.in_channels) for top in self.q_modules()
                   for m in top.modules() if isinstance(m, Conv3dHip) and m._rt.get('wT') is not None]
        self._tr_table = K.build_transpose_table(entries, device)
        self.reducer = None                       # ClipSGD all-reduces the whole gradient arena after backward
        self._m_ring = StagingRing((1,), torch.float32, device)
        self._m_cpu = torch.zeros(1, dtype=torch.float32)
        self._step = 0
        self.sync_shadows()
        return self

    @torch.no_grad()
    def sync_shadows(self):
        ar = self._standalone()
        K.cast_bf16(ar.Q, ar.Qb)
        K.cast_bf16(ar.KX, ar.Kb)
        for fn in self._q_refresh + self._k_refresh:
            fn()
        K.weight_transpose_batched(*self._tr_table)

    @torch.no_grad()
    def refresh_after_optimizer(self):
        for fn in self._q_refresh:
            fn()
        K.weight_transpose_batched(*self._tr_table)

    def sync_streams(self):
        pass                                      # one stream

    def _standalone(self):
        ar = getattr(self, 'arena', None)
        if ar is None:
            raise MsclError('stand-alone MoCoV2: call materialize(device) first (inside MSCLWithAug the outer model owns the step)')
        return ar

    def zero_grad(self, set_to_none=False):
        self._standalone().G.zero_()

    def forward_train(self, im_q, im_k, aux_info=None, return_features=False, update_queue=True):
        """ref: moco.py:473-515.  im_q / im_k: (B,3,T,H,W) fp32 device clips.  Returns the losses dict (top1_acc, top5_acc,
        loss_cls with the head's suffix); with return_features also dict(q, k)."""
        self._standalone()
        B = im_q.shape[0]
        K.ZEROS.reset(im_q.device)
        bg = B * parallel.world_size()
        self.batch_size = bg
        self.m = momentum_at(self.iters, self.max_iters, self.m_base)
        self._m_cpu[0] = self.m
        self._m_ring.push(self._m_cpu)
        aug = self.aug_gpu
        self.momentum_update(self._m_ring.dev)                                   # moco.py:534
        with torch.no_grad():
            x_k = parallel.shuffle_select(im_k, self._step, 0)                   # shuffle-BN (moco.py:146-172)
            k = self.encode_k(aug.pack_rgb(x_k))[0]
            k = parallel.unshuffle_select(k, self._step, 0)                      # moco.py:174-191
        q, _ = self.encode_q(aug.pack_rgb(im_q), levels=())                      # q_mlvl has no reader in MoCoHead.loss
        loss, rank = _MoCoLossFn.apply(q, k, self, update_queue)
        if self.training:
            self.iters += bg
        self._step += 1
        sfx = self.moco_head.basename
        rk = rank.float()
        losses = OrderedDict()
        losses['top1_acc' + sfx] = (rk < 1).float().mean()
        losses['top5_acc' + sfx] = (rk < 5).float().mean()
        losses['loss_cls' + sfx] = loss
        self._dbg = dict(q=q.detach(), k=k)
        if return_features:
            return losses, dict(q=q, k=k)
        return losses

    def train_step(self, data_batch, optimizer=None, sync_logs=True, **kwargs):
        """ref: moco.py:442-460 + recognizers/base.py:274-308: {'loss', 'log_vars', 'num_samples'}"""
        im_q, im_k = data_batch[self.im_key][0], data_batch[self.im_key][1]
        losses = self.forward_train(im_q, im_k, {})
        loss = sum(v for k2, v in losses.items() if 'loss' in k2)
        vals = torch.stack([v.detach().float() for v in losses.values()] + [loss.detach().float()])
        if not parallel.single():
            dist.all_reduce(vals)
            vals = vals / parallel.world_size()
        keys = list(losses.keys()) + ['loss']
        log_vars = OrderedDict(zip(keys, vals.tolist())) if sync_logs else OrderedDict(zip(keys, vals.unbind()))
        return dict(loss=loss, log_vars=log_vars, num_samples=im_q.shape[0])

    @torch.no_grad()
    def dequeue_and_enqueue(self, keys, gathered=None):
        """ref: moco.py:423-440 (keys are all-gathered first; bookkeeping is bit-exact int64 on device).
        `gathered`: the keys of all replicas in global sample order, when the caller already holds them."""
        keys = gathered if gathered is not None else parallel.all_gather_cat(keys)
        if self.K % keys.shape[0] != 0:
            raise AssertionError('K % batch_size == 0 (moco.py:432)')
        K.queue_enqueue(self.queue, self.count, self.queue_ptr, keys.contiguous())


class _MoCoLossFn(torch.autograd.Function):
    """InfoNCE of ONE recognizer against its own queue (moco.py:481-498 + heads/moco_head.py:38-77): the loss phase of a
    stand-alone MoCoV2 step.  Returns (loss_cls, rank per row); the query gradient is computed here, as in _MSCLLossFn."""

    @staticmethod
    def forward(ctx, q, k, rec, update_queue):
        B = q.shape[0]
        inv_T = 1.0 / rec.T
        ones = torch.full((B,), 1.0 / B, device=q.device)
        pos = K.rowdot(q, k)
        lse, loss_rows, rank = K.nce_forward(rec.queue, rec.count, q, pos, inv_T)
        dq = K.nce_backward(rec.queue, rec.count, q, lse, ones, inv_T, pos_pair=(k.contiguous(), pos))
        if update_queue:
            rec.dequeue_and_enqueue(k)
        ctx.save_for_backward(dq)
        ctx.mark_non_differentiable(rank)
        return loss_rows.mean(), rank

    @staticmethod
    def backward(ctx, g, _grank):
        return g * ctx.saved_tensors[0], None, None, None


class KeyGraph:
    """EMA update + key-encoder forward of one call site as a replayable HIP graph.

    The key branches carry no gradient and contain no collective (the shuffle-BN exchanges sit before and after them),
    so they can be captured on their own even when the step as a whole is launched eagerly -- which is how world
    size > 1 runs (RCCL inside one whole-step graph was only validated on a one-GPU box).  That
    takes ~330 of the ~900 kernel launches per step off the host, which is what bounds the eager step.
    The first two calls run eagerly (lazy plans, first-use allocations), the third captures and replays.
    BatchNorm statistics scratch comes from a private pre-zeroed pool that the graph clears itself."""

    def __init__(self, warmup=2):
        self.graph, self.shape, self.calls, self.warmup = None, None, 0, warmup
        self.pool = K.ZeroPool()
        self.failed = False

    def _body(self, rec, x, m_dev):
        rec.momentum_update(m_dev)
        return rec.encode_k(x)[0]

    def input_slot(self, shape):
        """the graph's own input buffer when the next run() of a packed clip of this shape will be a replay: the caller packs
        straight into it and run() skips its copy (a 26-MB device copy per key branch and eager step otherwise); else None"""
        if self.graph is not None and self.shape == tuple(shape) and not torch.cuda.is_current_stream_capturing():
            return self.static_in
        return None

    def run(self, rec, x, m_dev):
        if torch.cuda.is_current_stream_capturing():        # inside a whole-step capture: stay part of that graph
            return self._body(rec, x, m_dev)
        if self.graph is not None and self.shape == tuple(x.shape):
            if x.data_ptr() != self.static_in.data_ptr():
                self.static_in.copy_(x)
            self.graph.replay()
            return self.out
        self.calls += 1
        if self.calls <= self.warmup:
            return self._body(rec, x, m_dev)
        self.graph, self.shape = None, tuple(x.shape)
        self.static_in = x.clone()
        stream = torch.cuda.current_stream()
        parallel.settle_before_capture()
        graph = torch.cuda.CUDAGraph()
        shared = K.ZEROS
        self.pool.reset(x.device, size=2 << 20)
        try:
            K.ZEROS = self.pool
            with torch.cuda.graph(graph, stream=stream if stream != torch.cuda.default_stream() else None,
                                  capture_error_mode='thread_local'):      # other threads (RCCL watchdog) keep running
                self.pool.buf.zero_()
                self.out = self._body(rec, self.static_in, m_dev)
        except Exception as e:      # noqa: BLE001 -- nothing has executed during a capture: the eager launch below is still exact
            K.ZEROS = shared
            self.failed, self.error, self.warmup = True, f'{type(e).__name__}: {e}', 1 << 60
            return self._body(rec, x, m_dev)
        finally:
            K.ZEROS = shared
        self.graph = graph
        graph.replay()
        return self.out


class _ReplayFn(torch.autograd.Function):
    """autograd node of a QueryGraph replay: forward and backward are one graph launch each"""

    @staticmethod
    def forward(ctx, x, anchor, holder):
        if x.data_ptr() != holder.static_x.data_ptr():      # (the caller may have packed straight into the graph's input: input_slot)
            holder.static_x.copy_(x)
        holder.fwd.replay()
        ctx.holder = holder
        return holder.q.detach(), holder.p.detach()

    @staticmethod
    def backward(ctx, dq, dp):
        h = ctx.holder
        h.gq.copy_(dq)
        h.gp.copy_(dp)
        h.bwd.replay()
        nn_hip._bucket_done(h.trigger)          # the flow trunk's bucket (counts both traversals, GradReducer.need)
        return None, None, None


class QueryGraph:
    """One flow query pass (trunk + projection + LMCL pooling) as two replayable HIP graphs, forward and backward.

    The flow trunk is ~160 small launches per pass and direction, run twice per step: a third of what the host issues in an
    eager (world size > 1) step after the key branches (KeyGraph).  Its backward holds no collective except the bucket
    trigger at the very end, which the replaying autograd node fires itself.  Forward and backward share one private
    memory pool, so the activations the backward reads are the ones the forward replay wrote."""

    def __init__(self, warmup=2):
        self.fwd, self.shape, self.calls, self.warmup = None, None, 0, warmup
        self.fpool, self.bpool = K.ZeroPool(), K.ZeroPool()
        self.failed = False

    def capture(self, body, x, trigger):
        dev = x.device
        self.static_x = x.clone()
        self.trigger = trigger
        stream = torch.cuda.current_stream()
        cap = dict(stream=stream if stream != torch.cuda.default_stream() else None, capture_error_mode='thread_local')
        mem = torch.cuda.graph_pool_handle()
        parallel.settle_before_capture()
        fwd, bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        shared = K.ZEROS
        self.fpool.reset(dev, size=1 << 20)
        self.bpool.reset(dev, size=1 << 20)
        nn_hip.HOLD_BUCKETS[0] = True
        try:
            K.ZEROS = self.fpool
            with torch.cuda.graph(fwd, pool=mem, **cap):
                self.fpool.buf.zero_()
                with torch.enable_grad():
                    q, p, self.map_shape = body(self.static_x)
            self.gq, self.gp = torch.zeros_like(q), torch.zeros_like(p)
            K.ZEROS = self.bpool
            with torch.cuda.graph(bwd, pool=mem, **cap):
                self.bpool.buf.zero_()
                torch.autograd.backward([q, p], [self.gq, self.gp])
        finally:
            K.ZEROS = shared
            nn_hip.HOLD_BUCKETS[0] = False
        self.q, self.p, self.fwd, self.bwd, self.shape = q.detach(), p.detach(), fwd, bwd, tuple(x.shape)

    def input_slot(self, shape):
        """as KeyGraph.input_slot: the forward graph's input buffer when the next run() of this shape will be a replay"""
        if (self.fwd is not None and self.shape == tuple(shape) and not self.failed and torch.is_grad_enabled()
                and not torch.cuda.is_current_stream_capturing()):
            return self.static_x
        return None

    def run(self, body, x, trigger, anchor):
        """body(x) -> (q, pooled map, map shape) is the eager formulation; returns the same triple"""
        usable = not (self.failed or torch.cuda.is_current_stream_capturing() or not torch.is_grad_enabled())
        if usable and self.fwd is not None and self.shape == tuple(x.shape):
            q, p = _ReplayFn.apply(x, anchor, self)
            return q, p, self.map_shape
        self.calls += 1
        if not usable or self.calls <= self.warmup:
            return body(x)
        try:
            self.capture(body, x, trigger)
        except Exception as e:      # noqa: BLE001 -- nothing has executed during a capture: the eager launch below is still exact
            self.failed, self.error = True, f'{type(e).__name__}: {e}'
            # ... but the aborted backward capture may have QUEUED deferred weight gradients whose operands live in the graph's
            # private pool, and the engine skipped the end-of-backward callback that would have launched them: drop them, or the
            # next real backward launches them on released memory (nn.WGradQueue.clear)
            nn_hip.WGRADS.clear()
            return body(x)
        q, p = _ReplayFn.apply(x, anchor, self)
        return q, p, self.map_shape


class _Split2Fn(torch.autograd.Function):
    """x -> (x[:n], x[n:]) along axis 0, whose backward hands the two gradients back as ONE tensor without arithmetic when they
    are neighbours in one buffer (the loss node lays its input gradients out that way, kernels.loss_unpack): autograd's own
    slicing would answer each half with a zero-filled full tensor and add the two -- six small launches on the step's serial
    path per split (profiles/r04_glue_launches.txt: 4 slice_backward + their adds)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n, ctx.rows = n, x.shape[0]
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, ga, gb):
        n, rows = ctx.n, ctx.rows
        if (ga is not None and gb is not None and ga.is_contiguous() and gb.is_contiguous() and ga.dtype == gb.dtype
                and ga.untyped_storage().data_ptr() == gb.untyped_storage().data_ptr()
                and ga.storage_offset() + ga.numel() == gb.storage_offset()):
            return torch.as_strided(ga, (rows,) + tuple(ga.shape[1:]), ga.stride(), ga.storage_offset()), None
        if ga is None and gb is None:
            return None, None
        ref = ga if ga is not None else gb
        ga = ga if ga is not None else ref.new_zeros((n,) + tuple(ref.shape[1:]))
        gb = gb if gb is not None else ref.new_zeros((rows - n,) + tuple(ref.shape[1:]))
        return torch.cat([ga, gb], 0), None


class _MSCLLossFn(torch.autograd.Function):
    """Loss phase as ONE graph node: the 7 InfoNCE terms over 3 queue snapshots + LMCL."""

    @staticmethod
    def forward(ctx, q_rgb, q_fb, q_fa, p_rgb, p_fb, p_fa, k_rgb, k_fb, k_fa, model):
        rec, recf = model.recognizer, model.recognizer_flow
        B, dim = q_rgb.shape
        inv_T, inv_Tx = 1.0 / rec.T, 1.0 / model.moco_mx_head.T
        if abs(inv_T - inv_Tx) > 1e-12 or abs(inv_T - 1.0 / recf.T) > 1e-12:
            raise NotImplementedError('all contrastive temperatures are equal in the MSCL configs (0.07)')
        w_intra, w_inter = model.weight_aug_flow
        use_aug_mx = w_inter > 0
        if not model.same_kn:
            raise NotImplementedError('same_kn=False is not used by the MSCL configs')
        dev = q_rgb.device
        t = p_rgb.shape[0] // B
        C, Cf = p_rgb.shape[1], p_fb.shape[1]
        # every row layout of this phase in ONE launch (was ~10 torch.cat / repeat / full kernels): kernels.loss_pack
        # (round 5: the pack launch also takes the positive logits of every row, the pass that sums the per-block query gradients
        # adds the positive pair's term, the enqueue moves its own pointer -- four launches per InfoNCE pass instead of six and one
        # per enqueue instead of two on this serial stretch)
        QA, KA, sA, QC, KC, sC, ones, flow, pack_ws, (posA, posB, posC) = K.loss_pack(
            q_rgb.contiguous(), q_fb.contiguous(), q_fa.contiguous(), k_rgb.contiguous(), k_fb.contiguous(), k_fa.contiguous(),
            p_fb.contiguous(), p_fa.contiguous(), t, use_aug_mx, w_intra)

        def run(queue_owner, Q, Kp, scale, pos, virt=None):
            lse, loss_rows, rank = K.nce_forward(queue_owner.queue, queue_owner.count, Q, pos, inv_T, virt)
            dq = K.nce_backward(queue_owner.queue, queue_owner.count, Q, lse, scale, inv_T, virt, pos_pair=(Kp, pos))
            return loss_rows, rank, dq

        # pass A: RGB queue before this step's enqueue (moco.py:484-488 snapshot; fr logits moco_head_v2.py:44,47): query rows
        # [q_rgb | q_fb | q_fa] against k_rgb.  It shares nothing with passes B -> enqueue -> C on the flow queue, and this phase is
        # a string of small launches with every other chain already joined: A runs on the (now idle) RGB-key stream beside B / C.
        main = torch.cuda.current_stream()
        fork = model._side_stream(1) if (model.two_streams and model.loss_fork) else None
        if fork is not None:
            fork.wait_stream(main)
            pack_ws.record_stream(fork)
        with torch.cuda.stream(fork if fork is not None else main):
            lossA, rankA, dA = run(rec, QA, KA, sA, posA)
        kg = model._kglobal
        # pass C reads the flow queue AFTER the base-flow enqueue (App. E-3): flow-aug intra loss, rf, rf_aug -- query rows
        # [q_fa | q_rgb | q_rgb] against keys [k_fa | k_fb | k_fa], the first group scaled by weight_aug_flow[0].  With loss_fork_c it
        # takes that snapshot "virtually" -- ages + 1, the columns the enqueue will write replaced by the keys it will write
        # (kernels.nce_forward(virt=)) -- so it does not wait for pass B and the write: it runs on the idle flow stream beside B
        forkC = model._side_stream(0) if (model.two_streams and model.loss_fork_c) else None
        newk = kg.get('fb') if kg.get('fb') is not None else k_fb          # what dequeue_and_enqueue will write (gathered at W > 1)
        if forkC is not None:
            forkC.wait_stream(main)
            for tns in (pack_ws, newk):
                tns.record_stream(forkC)
            with torch.cuda.stream(forkC):
                lossC, rankC, dC = run(recf, QC, KC, sC, posC, virt=(newk.contiguous(), recf.queue_ptr))
        # pass B: flow queue before enqueue -> loss_cls_flow
        lossB, rankB, dB = run(recf, q_fb.contiguous(), k_fb.contiguous(), ones, posB)
        if forkC is not None:
            main.wait_stream(forkC)                   # C has read the queue: the enqueue may overwrite it
            for tns in (lossC, rankC, dC):
                tns.record_stream(main)
        recf.dequeue_and_enqueue(k_fb, kg.get('fb'))                     # base pass: update_queue=True (mscl.py:239)
        if forkC is None:
            lossC, rankC, dC = run(recf, QC, KC, sC, posC)
        if model.update_aug_flow:
            recf.dequeue_and_enqueue(k_fa, kg.get('fa'))
        if fork is not None:
            main.wait_stream(fork)                    # pass A has read the RGB queue: the enqueue below may overwrite it
            for tns in (lossA, rankA, dA):
                tns.record_stream(main)
        rec.dequeue_and_enqueue(k_rgb, kg.get('rgb'))
        # LMCL (local_cl_head.py:57-73): RGB frame-slot features vs [base flow | rotated flow] frames (`flow`, laid out by loss_pack)
        trans = getattr(model.sup_head, 'trans_flow', None)
        if trans is not None:                 # Conv1d(Cf, 128, 1) over the frame axis == a linear map per frame (local_cl_head.py:65)
            trt = trans._rt
            flow_in = flow.view(B * 2 * t, Cf)
            flow = K.linear_fwd(flow_in, trt['w'], trt['b'], False).view(B, 2 * t, C)
        elif Cf != C:
            raise ValueError(f'LMCL needs equal channel counts (rgb {C}, flow {Cf}) or a flow transform (bkb_channels)')
        lsum, hits, dpr, dpf = K.lmcl(p_rgb.view(B, t, C), flow, 1.0 / model.sup_head.T)
        ctx.trans_rt = None
        if trans is not None:
            # The transform's parameter gradients are STAGED here and added into the arena by backward(): the step drivers follow
            # mmcv's order -- train_step, optimizer.zero_grad(), loss.backward() (OptimizerHook.after_train_iter) -- so anything
            # written into the arena during the forward is wiped before backward runs (round-2 advisor finding: the transform
            # trained on weight decay alone).
            stage = trt.get('stage')
            if stage is None or stage[0].device != dev:
                stage = trt['stage'] = (torch.empty_like(trt['dw']), torch.empty_like(trt['db']))
            stage[0].zero_(); stage[1].zero_()
            dpf = K.linear_bwd(flow_in, trt['w'], flow.view(B * 2 * t, C), dpf.view(B * 2 * t, C).contiguous(), stage[0], stage[1], False)
            ctx.trans_rt = trt

        # the 17 / 23 log entries (heads/moco_head.py:60-77 per group, base.py:297-298 total) in one launch
        logs = K.step_logs(rankA, lossA, rankB, lossB, rankC, lossC, lsum, hits, B, 3 if use_aug_mx else 2, w_intra, float(B * t))
        ctx.log_keys = LOG_KEYS if use_aug_mx else LOG_KEYS_NO_AUG_MX
        total = logs[-1]
        # gradients w.r.t. the differentiable inputs (loss weights folded in by the row scales), summed over the passes and laid
        # out in ONE buffer by one launch (kernels.loss_unpack): backward() then scales it once
        flat, sizes = K.loss_unpack(dA, dB, dC, dpr.reshape(B * t, C), dpf.reshape(B * 2 * t, Cf), B, dim, t, C, Cf, use_aug_mx)
        ctx.save_for_backward(flat)
        ctx.sizes, ctx.shapes = sizes, ((B, dim), (B, dim), (B, dim), (B * t, C), (B * t, Cf), (B * t, Cf))
        ctx.mark_non_differentiable(logs)
        model._log_keys = ctx.log_keys
        return total.clone(), logs

    @staticmethod
    def backward(ctx, g, _glogs):
        flat = ctx.saved_tensors[0] * g
        grads, o = [], 0
        for sz, shp in zip(ctx.sizes, ctx.shapes):
            grads.append(flat[o:o + sz].view(shp)); o += sz
        grads = tuple(grads)
        trt = ctx.trans_rt
        if trt is not None:                  # LMCL flow transform (mscl_r50): parameter gradients into the arena, scaled by g
            trt['dw'].add_(trt['stage'][0] * g)
            trt['db'].add_(trt['stage'][1] * g)
            trt['slot_w'].touched = True
            trt['slot_b'].touched = True
        return grads + (None, None, None, None)


@RECOGNIZERS.register_module()
class MSCLWithAug(nn.Module):
    def __init__(self, recognizer, recognizer_flow, moco_mx_head, sup_head, im_key='imgs', flow_key='flow_imgs',
                 aux_info=[], aug=dict(type='SyncMoCoAugmentV5', crop_size=112, t=(8, 8)), same_kn=True,
                 update_aug_flow=False, weight_aug_flow=(1.0, 1.0), train_cfg=None, test_cfg=None):
        super().__init__()
        from .registry import build_recognizer
        self.recognizer = build_recognizer(recognizer)
        self.recognizer_flow = build_recognizer(recognizer_flow)
        self.im_key, self.same_kn = im_key, same_kn
        self.update_aug_flow, self.weight_aug_flow = update_aug_flow, tuple(weight_aug_flow)
        if isinstance(flow_key, (list, tuple)):
            raise NotImplementedError('separate base/rotated flow keys (cat_flow=False) are not used by mscl_r18')
        self.cat_flow, self.flow_key = True, (flow_key,)
        self.aux_info = aux_info
        self.moco_mx_head = build_head(moco_mx_head)
        self.sup_head = build_head(sup_head)
        self.aug_gpu = build_ssl_aug(aug)
        self.arena = None
        self._log_keys = LOG_KEYS
        self._step = 0
        self._scal = self._idx = self._inv = None         # staging.StagingRing: per-step words, host -> device
        self._bg = 0
        self.shuffle_mode = 'a2a'      # 'a2a' | 'gather' (shuffle-BN exchange, world size > 1; graph.py sets 'gather')
        self._a2a = False
        self.two_streams = os.environ.get('MSCL_STREAMS', '3') != '1'
        self.stream_probing = True     # side streams chosen by the overlap probe (streams.py); False: the first ones created
        self.defer_transpose = True
        self._wt = None
        self.loss_fork = True          # RGB-queue InfoNCE pass beside the flow-queue passes
        # ... and (attribute, off) the post-enqueue flow-queue pass on a "virtual" snapshot beside the pre-enqueue one:
        # exact (test_nce_virtual_enqueue_equals_real_enqueue) but no faster -- 957.8 vs 957.6 clip-pairs/s over five alternating pairs
        self.loss_fork_c = False
        self.key_graphs = True          # key branches as sub-graphs in eager steps
        self._key_graph = [KeyGraph(), KeyGraph(), KeyGraph()]                  # RGB, flow base, flow rotated
        self.query_graphs = True      # flow query passes (fwd + bwd) likewise
        self._query_graph = [QueryGraph(), QueryGraph()]                         # flow base, flow rotated
        self._graph_anchor = None
        # base || rotated flow query clips in ONE trunk pass with two BatchNorm statistics groups (halves the ~250 launches of the
        # two query passes, forward and backward); needs a flow neck without parameters of its own (BaseMoCo)
        self.flow_batch = True
        # RGB weight gradients (leaves of the backward chain) off the main stream -- an experiment that stays OFF: '1' = a stream of
        # their own, 'flow' / 'key' = the flow / RGB-key stream (idle during most of the backward).  Measured in round 2 against
        # 954-957 clip-pairs/s: 889 / 851-856 / 892.  Two MFMA-heavy kernels side by side lose more than the shorter chain gains.
        # Also NOT safe under whole-step capture: the final loss of those runs sat 1.2 below the default's (36.0 vs 37.2 +- 0.2) --
        # tensor.record_stream() is what keeps dy / x alive for the side stream, and the allocator does not honour it for
        # allocations of a graph's private pool.
        self._side = None

    # ------------------------------------------------------------------ device placement
    def materialize(self, device='cuda'):
        """Move buffers to `device`, re-home every parameter into the flat arenas and bind the kernels'
        runtime views.  Must be called once before the first step; state_dict()/load_state_dict() keep
        working afterwards (parameters are views into the arenas)."""
        device = torch.device(device)
        if device.type != 'cuda':
            raise MsclError('materialize() needs a GPU device: the HIP path has no CPU fallback')
        from .lib import load
        load()
        ar = ParamArena(device)
        plan = []
        for name, rec in (('rgb', self.recognizer), ('flow', self.recognizer_flow)):
            ar.begin_group(name)
            for mq, mk in zip(rec.q_modules(), rec.k_modules()):
                for (nq, pq), (nk, pk) in zip(mq.named_parameters(), mk.named_parameters()):
                    assert nq == nk and pq.shape == pk.shape
                    plan.append((ar.add(nq, pq.shape), pq, pk))
            ar.end_group(name)
        # trainable parameters outside the two recognizers (mscl_r50: the LMCL head's flow transform): no key twin, no EMA
        ar.begin_group('head')
        for nq, pq in self.sup_head.named_parameters():
            plan.append((ar.add('sup_head.' + nq, pq.shape), pq, None))
        ar.end_group('head')
        ar.allocate()
        for slot, pq, pk in plan:
            vq = ar.view('Q', slot)
            vq.copy_(pq.data.to(device))
            pq.data = vq
            pq.grad = ar.view('G', slot)
            pq._mscl_slot = slot
            if pk is not None:
                vk = ar.view('KX', slot)
                vk.copy_(pk.data.to(device))
                pk.data = vk
                pk._mscl_slot = slot
        for mod in self.modules():
            for bname, buf in list(mod._buffers.items()):
                if buf is not None:
                    mod._buffers[bname] = buf.to(device)
        self.arena = ar
        # base || rotated flow query clips in one pass: only the TRUNK keeps one set of BatchNorm statistics per call, so the flow
        # neck must be parameter- and buffer-free (BaseMoCo: global average pool); anything else takes the two reference passes
        nq = self.recognizer_flow.neck_q
        if self.flow_batch and (any(True for _ in nq.parameters()) or any(True for _ in nq.buffers())):
            self.flow_batch = False
        for name, rec in (('rgb', self.recognizer), ('flow', self.recognizer_flow)):
            rec._arena, rec._range = ar, tuple(ar.ranges[name])
            rec._k_refresh, rec._q_refresh = [], []
            for mods, key in ((rec.q_modules(), False), (rec.k_modules(), True)):
                for top in mods:
                    for m in top.modules():
                        self._bind(m, ar, key, rec)
        for m in self.sup_head.modules():
            self._bind(m, ar, False, None)
        entries = []
        for rec in (self.recognizer, self.recognizer_flow):
            for top in rec.q_modules():
                for m in top.modules():
                    if isinstance(m, Conv3dHip) and m._rt.get('wT') is not None:
                        entries.append((m._rt['w'], m._rt['wT'], m.out_channels, m.taps, m.in_channels))
        self._tr_table = K.build_transpose_table(entries, device)
        self._wt = nn_hip.TransposeState(self._tr_table)
        for rec in (self.recognizer, self.recognizer_flow):
            for top in rec.q_modules():
                for m in top.modules():
                    if isinstance(m, Conv3dHip) and m._rt.get('wT') is not None:
                        m._rt['wt_state'] = self._wt      # every reader of a transposed kernel checks it (Conv3dHip.wT)
        # gradient all-reduce buckets (contiguous arena ranges, in backward-completion order) and their triggers
        def span(mods):
            slots = [p._mscl_slot for m in mods for p in m.parameters()]
            return (min(s.off for s in slots), max(s.off + (s.numel + 63) // 64 * 64 for s in slots))
        rgb, flw = self.recognizer, self.recognizer_flow
        eq = rgb.encoder_q
        # (range, trigger module whose backward completes it, traversals per step).  Order = the order backward completes them:
        # neck + projection MLP (14 MB; counted trigger: its convs are applied several times per step in an order autograd does
        # not promise -- nn.BucketCounter fires when the last application has run backward), layer 4 (100 MB), layer 3, layer 2,
        # stem + layer 1 (1.8 MB: all that is left to send when backward ends), the flow trunk.  Rule for the transport (DESIGN.md
        # section 5, parallel.exposed_wire_ms): what matters is the wire time of the buckets that fire with less backward left than
        # they take to send -- with this split 1.8 MB, 20 us on a 7-link ring at fp32 -- so gradients travel as fp32.
        buckets = [(span([eq.layer4]), eq.layer4[0], 1), (span([eq.layer3]), eq.layer3[0], 1), (span([eq.layer2]), eq.layer2[0], 1),
                   (span([eq.stem, eq.layer1]), eq.stem, 1), (span([rgb.neck_q, rgb.mlp_q]), None, 1),
                   (span([flw.encoder_q, flw.neck_q, flw.mlp_q, self.sup_head]), flw.encoder_q.stem,
                    1 if self.flow_batch else 2)]            # head group follows the flow group
        self.reducer = parallel.GradReducer(ar.G, [b[0] for b in buckets], need=[b[2] for b in buckets])
        for i, (_, trig, _n) in enumerate(buckets):
            if trig is not None:
                trig._grad_buckets = getattr(trig, '_grad_buckets', ()) + ((self.reducer, i),)
        neck_counter = nn_hip.BucketCounter(self.reducer, 4)
        for m in list(rgb.neck_q.modules()) + [rgb.mlp_q]:
            if isinstance(m, Conv3dHip) or m is rgb.mlp_q:
                m._bucket_counter = neck_counter
        self.sync_shadows()
        # split-K policy of the chains beside the RGB query chain: 4 slabs instead of 16 measured the same step rate within noise
        # (1197 vs 1193 clip-pairs/s, six alternating graphs in one process: profiles/r06_ab_sweeps.md) at a quarter of the slab traffic
        self.set_side_split(4, 4)
        return self

    def _bind(self, m, ar, key, rec):
        bind_module(m, ar, key, rec)

    @torch.no_grad()
    def sync_shadows(self):
        """bf16 shadows / transposed kernels / padded stems from the fp32 masters (after load or fill)."""
        ar = self.arena
        K.cast_bf16(ar.Q, ar.Qb)
        K.cast_bf16(ar.KX, ar.Kb)
        for rec in (self.recognizer, self.recognizer_flow):
            for fn in rec._q_refresh + rec._k_refresh:
                fn()
        self._wt.refresh()

    @torch.no_grad()
    def refresh_after_optimizer(self):
        if self._wt is None:                # before materialize(): no shadows or transposed copies exist yet (sync_shadows builds them)
            return
        for rec in (self.recognizer, self.recognizer_flow):
            for fn in rec._q_refresh:
                fn()
        # The transposed kernels are read by the input-gradient kernels only, i.e. not before the NEXT step's backward: their
        # refresh (one 45-us launch over every conv weight) leaves the serial tail of the step and runs at the head of the next
        # one on the flow stream, beside the forward (_device_step): 1033 vs 1026 clip-pairs/s.  (The same for the gradient clear --
        # on the key stream from the start of the loss phase on -- lost: 1016-1023 vs 1032-1035; it competes with the queue passes.)
        # Any backward that does not come through _device_step (encode_q + backward, a custom step function) refreshes them
        # lazily at its first input gradient (nn.TransposeState / Conv3dHip.wT).
        if self.defer_transpose:
            self._wt.stale = True
        else:
            self._wt.refresh()

    def zero_grad(self, set_to_none=False):
        if self.arena is not None:
            self.arena.G.zero_()
        else:
            super().zero_grad(set_to_none)

    # ------------------------------------------------------------------ step
    def train_step(self, data_batch, optimizer=None, sync_logs=True, **kwargs):
        """ref: mscl.py:192-212.  Returns dict(loss, log_vars, num_samples).  With sync_logs=False
        `log_vars` holds 0-d device tensors (no host sync in the step), else Python floats."""
        im_q, im_k = data_batch[self.im_key][0], data_batch[self.im_key][1]
        aux = {}
        for fk in self.flow_key:
            aux[f'{fk}_q'], aux[f'{fk}_k'] = data_batch[fk][0], data_batch[fk][1]
        for item in self.aux_info:
            assert item in data_batch
            aux[item] = data_batch[item]
        data_batch = self.with_aug_draw(data_batch)
        if 'flip_mask' in data_batch:           # [mask_q, mask_k], uint8 (B,): the Bernoulli draw of ssl_aug_v2.py:107-110
            aux['flip_q'], aux['flip_k'] = data_batch['flip_mask'][0], data_batch['flip_mask'][1]
        if 'aug_params' in data_batch:          # [rows_q, rows_k], fp32 (B,16): jitter / grayscale / blur parameters
            aux['color_q'], aux['color_k'] = data_batch['aug_params'][0], data_batch['aug_params'][1]
        loss, logs = self.forward_train(im_q, im_k, aux)
        log_vars = self._parse_logs(logs, sync_logs)
        return dict(loss=loss, log_vars=log_vars, num_samples=im_q.shape[0])

    def with_aug_draw(self, data_batch):
        """a stochastic augmenter draws this step's flip masks and colour parameters unless the batch brings its own"""
        aug = self.aug_gpu
        if not getattr(aug, 'stochastic', False) or 'aug_params' in data_batch:
            return data_batch
        dev = data_batch[self.im_key][0].device
        drawn = aug.draw(data_batch[self.im_key][0].shape[0])
        out = dict(data_batch)
        for k, v in drawn.items():
            if k not in out:
                out[k] = [t.to(dev, non_blocking=True) for t in v]
        return out

    def forward(self, im_q, im_k, aux_info, return_loss=True, **kwargs):
        if not return_loss:
            raise NotImplementedError('MoCo doesnt support test mode')
        return self.forward_train(im_q, im_k, aux_info)

    def _parse_logs(self, logs, sync):
        """ref: base.py:287-306, as ONE packed all-reduce instead of 23 scalar collectives."""
        if not parallel.single():
            logs = logs.clone()
            dist.all_reduce(logs)
            logs = logs / parallel.world_size()
        keys = self._log_keys
        if sync:
            vals = logs.tolist()
            return OrderedDict(zip(keys, vals))
        return OrderedDict((k, logs[i]) for i, k in enumerate(keys))

    def forward_train(self, im_q, im_k, aux_info):
        """ref: mscl.py:225-277."""
        if self.arena is None:
            raise MsclError('call model.materialize("cuda") before the first step')
        fk = self.flow_key[0]
        self._pre_step_host(im_q.shape[0])
        self._upload_step_words()
        loss, logs = self._device_step(im_q, im_k, aux_info[f'{fk}_q'], aux_info[f'{fk}_k'],
                                       aux_info.get('flip_q'), aux_info.get('flip_k'),
                                       aux_info.get('color_q'), aux_info.get('color_k'))
        self._post_step_host()
        return loss, logs

    # The step is split into host bookkeeping (Python ints / the momentum schedule) and device work, so that
    # the device work can be captured once into a HIP graph and replayed (mscl_amd/graph.py).
    def _pre_step_host(self, B):
        rec, recf = self.recognizer, self.recognizer_flow
        W = parallel.world_size()
        bg = B * W
        if self._scal is None or tuple(self._idx.dev.shape) != (18, B) or tuple(self._inv.dev.shape) != (3, bg):
            # (re)built when the per-GPU batch or the world size changes; the values of a step are assembled in plain host
            # tensors here and travel through rings of pinned slots in _upload_step_words (staging.py: a pinned word
            # rewritten while its asynchronous copy is still queued hands step n the permutation of step n+k)
            dev = self.arena.device
            self._scal = StagingRing((4,), torch.float32, dev)
            self._idx = StagingRing((18, B), torch.long, dev)
            self._inv = StagingRing((3, bg), torch.long, dev)               # argsort(perm): gathered keys -> global order
            self._scal_cpu = torch.zeros(4, dtype=torch.float32)
            self._idx_cpu = torch.zeros((18, B), dtype=torch.long)
            self._inv_cpu = torch.zeros((3, bg), dtype=torch.long)
        # all-to-all split sizes change every step, so a captured graph (graph.py) switches to the all-gather formulation
        self._a2a = (not parallel.single()) and self.shuffle_mode == 'a2a'
        rec.m = momentum_at(rec.iters, rec.max_iters, rec.m_base)
        m1 = momentum_at(recf.iters, recf.max_iters, recf.m_base)
        recf.m = momentum_at(recf.iters + (bg if self.training else 0), recf.max_iters, recf.m_base)   # value after the 2nd pass
        self._scal_cpu[0], self._scal_cpu[1], self._scal_cpu[2] = rec.m, m1, recf.m
        if not parallel.single():
            r = parallel.rank()
            self._plans = [None] * 3
            for slot in range(3):
                perm = parallel.shuffle_perm(W * B, self._step, slot)
                self._inv_cpu[slot] = torch.argsort(perm)
                if self._a2a:                   # two all-to-alls move B rows per rank instead of gathering W * B
                    plan = self._plans[slot] = parallel.ShufflePlan(W, B, r, perm)
                    self._idx_cpu[6 + 4 * slot:10 + 4 * slot] = plan.index_rows()
                else:
                    self._idx_cpu[slot] = perm.view(W, B)[r]
        self._bg = bg

    def _upload_step_words(self):
        """send what _pre_step_host assembled to the device words the kernels read, on the current stream.  Never inside a
        graph capture: the whole-step graph (graph.py) calls this before each replay."""
        self._scal.push(self._scal_cpu)
        if not parallel.single():
            self._idx.push(self._idx_cpu)
            self._inv.push(self._inv_cpu)

    def _post_step_host(self):
        rec, recf = self.recognizer, self.recognizer_flow
        rec.batch_size = recf.batch_size = self._bg
        if self.training:                       # moco.py:504-505; the flow recognizer ran twice
            rec.iters += rec.batch_size
            recf.iters += 2 * recf.batch_size
        self._step += 1

    def _side_stream(self, i=0):
        if self._side is None:
            self._side = self._pick_streams(3)
        return self._side[i]

    def _pick_streams(self, n):
        """n side streams that really run next to the current stream (and, when a process group is up, next to the
        communicator): the selection logic and why it exists are in streams.py; here are the timing primitives."""
        import time
        from .streams import pick_side_streams
        dev = self.arena.device
        cand = [torch.cuda.Stream(device=dev) for _ in range(12 if self.stream_probing else n)]
        if len(cand) == n or torch.cuda.is_current_stream_capturing():
            return cand[:n]
        main = torch.cuda.current_stream()

        def spin(streams, cycles):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for st in [main] + list(streams):
                with torch.cuda.stream(st):
                    torch.cuda._sleep(cycles)
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        class _Comm:
            def __init__(self):
                self.buf = torch.zeros(48 << 20, device=dev)

            def agree_min(self, v):
                t = torch.tensor([int(v)], device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                return int(t.item())

            def timed(self, streams, cyc):
                dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for st in streams:
                    with torch.cuda.stream(st):
                        torch.cuda._sleep(cyc)
                dist.all_reduce(self.buf)                               # issued from the idle main stream
                torch.cuda.synchronize()
                return time.perf_counter() - t0
        use_comm = not parallel.single() and dist.get_backend() == 'nccl'
        chosen, self.stream_probe = pick_side_streams(cand, n, spin, _Comm() if use_comm else None)
        return chosen

    def set_side_split(self, cap_key=16, cap_flow=16):
        """Split-K policy per chain (round 6).  The step follows the SUM of kernel time over its three streams, and a split-K launch
        buys its latency with slab traffic (k fp32 copies of the output written, read again by splitk_finalize): worth it on the RGB
        query chain, which paces the step, not on the chains that run beside it.  cap_key / cap_flow: the most slabs a conv launch
        of the RGB key encoder / of the flow recognizer (query and key) may use; the RGB query chain keeps 16.
        (measured: profiles/r06_ab_sweeps.md)"""
        for mods, cap in ((self.recognizer.k_modules(), cap_key),
                          (self.recognizer_flow.q_modules() + self.recognizer_flow.k_modules(), cap_flow)):
            for m in mods:
                for c in m.modules():
                    if isinstance(c, Conv3dHip):
                        c.split_cap = int(cap)

    def sync_streams(self):
        """make the current stream wait for the flow stream (parameter gradients are written by kernels, not by
        autograd's AccumulateGrad, so the optimizer orders itself explicitly)"""
        if self._side is not None:
            for st in self._side:
                torch.cuda.current_stream().wait_stream(st)

    def _shuffle(self, x, slot):
        if parallel.single():
            return x          # a within-batch permutation does not change per-GPU BN statistics
        if self._a2a:
            p, ix = self._plans[slot], self._idx.dev
            return parallel.exchange_rows(x, ix[6 + 4 * slot], ix[7 + 4 * slot], p.send_splits, p.recv_splits)
        return parallel.all_gather_cat(x).index_select(0, self._idx.dev[slot])

    def _shuffle_mask(self, m, slot):
        if m is None or parallel.single():
            return m
        return self._shuffle(m.view(-1, 1), slot).view(-1).contiguous()

    def _encode_key(self, slot, rec, x, m_dev):
        """EMA update (moco.py:408-421) then the key forward (moco.py:535-545), replayed from a sub-graph when allowed"""
        if self.key_graphs and self.training:
            return self._key_graph[slot].run(rec, x, m_dev)
        rec.momentum_update(m_dev)
        return rec.encode_k(x)[0]

    def _key_slot(self, slot, shape):
        """where the packed key clip of call site `slot` goes: the key sub-graph's own input buffer when that graph is about to be
        replayed (no copy into it then), else a fresh tensor (None)"""
        if self.key_graphs and self.training:
            return self._key_graph[slot].input_slot(shape)
        return None

    def active_query_graphs(self):
        """the QueryGraph holders the step uses: one (base || rotated in one pass) with flow_batch, else one per pass"""
        return self._query_graph[:1] if self.flow_batch else self._query_graph

    def _flow_query_body(self, x):
        q, maps = self.recognizer_flow.encode_q(x)
        m = maps[self.sup_head.mlvl_ids[1]]
        return q, pool(m, m.shape[0] * m.shape[1], m.shape[2] * m.shape[3]), tuple(m.shape)

    def _flow_query_body2(self, x):
        """base || rotated clips (2B samples) in one pass: two BatchNorm statistics groups, rows [0, B) are the base call"""
        q, maps = self.recognizer_flow.encode_q(x, bn_groups=2)
        m = maps[self.sup_head.mlvl_ids[1]]
        return q, pool(m, m.shape[0] * m.shape[1], m.shape[2] * m.shape[3]), tuple(m.shape)

    def _flow_query(self, slot, x):
        """one flow query pass (moco.py:517-529 on the flow recognizer) + the LMCL pooling of its layer-4 map
        (local_cl_head.py:57-62); replayed from a forward and a backward sub-graph in eager training steps.
        slot 0 / 1: the base / rotated pass alone; slot 2: both in one pass over 2B samples (flow_batch)"""
        body = self._flow_query_body2 if slot == 2 else self._flow_query_body
        if not (self.query_graphs and self.training):
            return body(x)
        if self._graph_anchor is None:
            self._graph_anchor = torch.zeros(1, device=x.device, requires_grad=True)
        return self._query_graph[0 if slot == 2 else slot].run(body, x, self.recognizer_flow.encoder_q.stem, self._graph_anchor)

    def _device_step(self, im_q, im_k, flow_q, flow_k, flip_q=None, flip_k=None, color_q=None, color_k=None):
        rec, recf = self.recognizer, self.recognizer_flow
        T2 = flow_q.shape[2]
        if T2 % 2:
            raise ValueError('flow clips must hold base and rotated halves along T')
        Th = T2 // 2
        aug = self.aug_gpu
        K.ZEROS.reset(im_q.device)
        # weight gradients left queued by a backward pass that raised (its end-of-backward callback never ran) belong to no pass any
        # more: they are dropped here rather than launched by this step's backward on operands that may be gone
        nn_hip.WGRADS.clear()
        dp = not parallel.single()
        sc = self._scal.dev                         # this step's words: uploaded by _upload_step_words on this stream
        ids = self.sup_head.mlvl_ids
        hw = lambda m: m.shape[2] * m.shape[3]
        # Three independent chains meet only in the loss: RGB query (current stream), RGB key, flow (query + key).
        # The flow chain is ~200 small, latency-bound launches and the key chains carry no gradient, so they
        # run on side HIP streams and fill the CUs the big RGB-query kernels leave idle (tail waves, small layers);
        # autograd replays each node's backward on its forward stream, so the backward passes overlap too.
        # World size > 1: every collective runs on the communicator's ONE stream in host program order, each waiting for
        # the work already issued to the stream it was called from.  So the three shuffle-BN exchanges of the key clips
        # are issued first (their inputs exist at step start; issued after the flow query passes they would hold the RGB
        # key branch back by that whole chain), and the encoded keys come back in ONE all-gather at the join below, which
        # also serves both queue writes (moco.py:426) -- 4 collectives per step in the forward instead of 9.
        main = torch.cuda.current_stream()
        multi = self.two_streams
        s_fq = self._side_stream(0) if multi else main       # flow query passes (base, rotated): share BN running stats -> in order
        s_fk = s_fq                                          # flow key passes share the flow stream (a 4th stream measured 3-6 % slower)
        side = s_fq
        for st in (s_fq, s_fk):
            if st is not main:
                st.wait_stream(main)
        side_k = self._side_stream(1) if side is not main else main   # RGB key chain on its own stream (on the flow stream: -7 %)
        if side_k is not main:
            side_k.wait_stream(main)
        if self._wt.stale or torch.cuda.is_current_stream_capturing():      # (a captured step always carries the refresh)
            with torch.cuda.stream(s_fq):
                self._wt.refresh()                                          # deferred by refresh_after_optimizer; joined before the loss
        if dp:
            # Issued from a stream of their own, never from one that later captures a sub-graph: a collective whose
            # completion the process group's watchdog has not yet polled, issued from stream S, made the watchdog's
            # event query fail ("event last recorded in a capturing stream") once S began the backward capture.
            xs = self._side_stream(2) if multi else main
            if xs is not main:
                xs.wait_stream(main)
            with torch.cuda.stream(xs):
                im_k = aug.color(im_k, color_k, 1)           # on the owner, before the shuffle (mscl.py:227 precedes moco.py:532)
                im_k_x, flip_k0 = self._shuffle(im_k, 0), self._shuffle_mask(flip_k, 0)
                fk_b, flip_k1 = self._shuffle(flow_k[:, :, :Th], 1), self._shuffle_mask(flip_k, 1)
                fk_a, flip_k2 = self._shuffle(flow_k[:, :, Th:], 2), self._shuffle_mask(flip_k, 2)
            if xs is not main:
                for st in (side_k, s_fk):
                    st.wait_stream(xs)
                for tns in (im_k_x, fk_b, fk_a, flip_k0, flip_k1, flip_k2):
                    if tns is not None:
                        tns.record_stream(side_k if tns is im_k_x or tns is flip_k0 else s_fk)
        # -- RGB query branch (the step's critical chain, on the main stream)
        def issue_query():
            x_q = aug.pack_rgb(aug.color(im_q, color_q, 0), flip_q)
            return rec.encode_q(x_q, levels=(ids[0],))       # LMCL reads one pyramid level (local_cl_head.py:59)
        # (only while a whole-step capture records: there the order is just the graph's node order.  Eager launches keep the query
        #  chain last -- its ~250 host-side launches would otherwise delay the side chains' cheap sub-graph replays by ~2.5 ms)
        # under whole-step capture the RGB query chain (the step's critical chain) is recorded before the side chains (+0.3 %);
        # eager launches always issue it last
        q_first = torch.cuda.is_current_stream_capturing()
        if q_first:
            q_rgb, maps_rgb = issue_query()
        with torch.cuda.stream(s_fq):
            if self.flow_batch:
                Bq = flow_q.shape[0]
                xshape = (2 * Bq, Th, flow_q.shape[3], flow_q.shape[4], 8)
                xq = self._query_graph[0].input_slot(xshape) if (self.query_graphs and self.training) else None
                if xq is None:
                    xq = torch.empty(xshape, dtype=torch.bfloat16, device=flow_q.device)
                aug.pack_flow(flow_q, 0, Th, flip_q, out=xq[:Bq])
                aug.pack_flow(flow_q, Th, Th, flip_q, out=xq[Bq:])
                q_f, p_f, fs = self._flow_query(2, xq)
                tq = fs[1]
                q_fb, q_fa = _Split2Fn.apply(q_f, Bq)
                p_fb, p_fa = _Split2Fn.apply(p_f, Bq * tq)
                fmap_shape = (Bq,) + tuple(fs[1:])
            else:
                q_fb, p_fb, fmap_shape = self._flow_query(0, aug.pack_flow(flow_q, 0, Th, flip_q))
                q_fa, p_fa, _ = self._flow_query(1, aug.pack_flow(flow_q, Th, Th, flip_q))
        with torch.cuda.stream(s_fk):
            # two EMA updates, two BN-statistics passes (App. E-5)
            kshape = (flow_k.shape[0], Th, flow_k.shape[3], flow_k.shape[4], 8)
            if dp:
                k_fb = self._encode_key(1, recf, aug.pack_flow(fk_b, 0, Th, flip_k1, out=self._key_slot(1, kshape)), sc[1:2])
                k_fa = self._encode_key(2, recf, aug.pack_flow(fk_a, 0, Th, flip_k2, out=self._key_slot(2, kshape)), sc[2:3])
            else:
                k_fb = self._encode_key(1, recf, aug.pack_flow(flow_k, 0, Th, flip_k, out=self._key_slot(1, kshape)), sc[1:2])
                k_fa = self._encode_key(2, recf, aug.pack_flow(flow_k, Th, Th, flip_k, out=self._key_slot(2, kshape)), sc[2:3])
        # -- RGB key branch (no gradient): a third stream, it only meets the query branch in the loss
        with torch.cuda.stream(side_k):
            rshape = (im_k.shape[0], im_k.shape[2], im_k.shape[3], im_k.shape[4], 8)
            if dp:
                x_k = aug.pack_rgb(im_k_x, flip_k0, out=self._key_slot(0, rshape))
            else:
                x_k = aug.pack_rgb(aug.color(im_k, color_k, 1), flip_k, out=self._key_slot(0, rshape))
            k_rgb = self._encode_key(0, rec, x_k, sc[0:1])
        if not q_first:
            q_rgb, maps_rgb = issue_query()
        if side_k is not main:
            main.wait_stream(side_k)
            k_rgb.record_stream(main)
        # -- LMCL inputs (local_cl_head.py:57-62): TPN level 0 of RGB, raw layer-4 maps of both flow passes
        m_rgb = maps_rgb[ids[0]]
        p_rgb = pool(m_rgb, m_rgb.shape[0] * m_rgb.shape[1], hw(m_rgb))
        if multi:
            main.wait_stream(s_fq)
            main.wait_stream(s_fk)
            for tns in (q_fb, k_fb, q_fa, k_fa, p_fb, p_fa):
                tns.record_stream(main)
        self._kglobal = {}
        if dp:
            # keys come back: one all-gather of the three (B,128) blocks, put into global sample order with the inverse
            # permutations; the own rows are this rank's keys (moco.py:174-191), the whole is what the queues enqueue
            full, (k_rgb, k_fb, k_fa) = parallel.gather_unshuffle([k_rgb, k_fb, k_fa], self._inv.dev)
            self._kglobal = dict(rgb=full[0], fb=full[1], fa=full[2])
        if m_rgb.shape[1] != fmap_shape[1] or m_rgb.shape[1] != self.sup_head.t:
            raise ValueError(f'LMCL needs equal frame-slot counts: rgb {m_rgb.shape[1]}, flow {fmap_shape[1]}, head t={self.sup_head.t}')
        self._dbg = dict(q_rgb=q_rgb.detach(), q_fb=q_fb.detach(), q_fa=q_fa.detach(), k_rgb=k_rgb, k_fb=k_fb, k_fa=k_fa)
        return _MSCLLossFn.apply(q_rgb, q_fb, q_fa, p_rgb, p_fb, p_fa, k_rgb, k_fb, k_fa, self)

    def forward_test(self, imgs):
        raise NotImplementedError('Not support for ssl recognizer !!!')
