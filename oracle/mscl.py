"""Oracle MoCoV2 / MSCLWithAug step (fp32, plain torch, any device).  TEST INFRASTRUCTURE ONLY.

Distributed behaviour: if torch.distributed is initialised the same collectives as the reference
are issued (all_gather / broadcast); otherwise world size 1 semantics.  No `.cuda()` anywhere.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from .nets import BaseMoCo, TPNMoCo, build_trunk

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _world():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def _rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


@torch.no_grad()
def concat_all_gather(t):
    """ref: recognizers/moco.py:558-568."""
    if _world() == 1:
        return t.clone()
    parts = [torch.ones_like(t) for _ in range(_world())]
    dist.all_gather(parts, t.contiguous())
    return torch.cat(parts, dim=0)


def top_k_accuracy(scores, labels, topk=(1,)):
    """ref: core/evaluation/accuracy.py:130-149 (host argsort; label hit if among the k largest)."""
    scores = np.asarray(scores)
    labels = np.asarray(labels)[:, None]
    res = []
    for k in topk:
        top = np.argsort(scores, axis=1)[:, -k:][:, ::-1]
        hit = np.logical_or.reduce(top == labels, axis=1)
        res.append(hit.sum() / hit.shape[0])
    return res


def cross_entropy(logits, labels, ignore_index=-1, loss_weight=1.0):
    """ref: losses/cross_entropy_loss.py:122-138 (CrossEntropyLoss_torch, mean reduction)."""
    return loss_weight * F.cross_entropy(logits, labels, ignore_index=ignore_index, reduction='mean')


def ce_with_topk(logits, labels, suffix):
    """ref: heads/moco_head.py:38-77 == heads/moco_head_v2.py:55-94: top-1/top-5 then CE."""
    out = OrderedDict()
    acc = top_k_accuracy(logits.detach().float().cpu().numpy(), labels.detach().cpu().numpy(), (1, 5))
    out[f'top1_acc{suffix}'] = torch.tensor(acc[0], device=logits.device)
    out[f'top5_acc{suffix}'] = torch.tensor(acc[1], device=logits.device)
    out[f'loss_cls{suffix}'] = cross_entropy(logits, labels)
    return out


def momentum_at(iters, max_iters, m_base):
    """ref: recognizers/moco.py:413-415 (cosine anneal of the key-encoder momentum)."""
    factor = min(iters / max_iters, 1)
    return 1 - 0.5 * (1 - m_base) * (math.cos(math.pi * factor) + 1)


class MoCoV2(nn.Module):
    """ref: recognizers/moco.py:324-406 (construction) and :408-547 (step pieces)."""

    def __init__(self, kind, dim_in, dim=128, K=65536, m_base=0.994, max_iters=1, T=0.07,
                 neck=None, basename=''):
        super().__init__()
        self.K, self.m_base, self.m, self.T = K, m_base, m_base, T
        self.iters, self.max_iters, self.batch_size = 0, max_iters, 0
        self.suffix = ('_' + basename) if basename else ''
        mk_neck = (lambda: TPNMoCo(**neck)) if neck is not None else BaseMoCo
        self.encoder_q, self.encoder_k = build_trunk(kind), build_trunk(kind)
        self.neck_q, self.neck_k = mk_neck(), mk_neck()
        mlp = lambda: nn.Sequential(nn.Linear(dim_in, dim_in), nn.ReLU(), nn.Linear(dim_in, dim))
        self.mlp_q, self.mlp_k = mlp(), mlp()
        for q, k in self._qk_pairs():
            k.data.copy_(q.data)
            k.requires_grad = False
        self.register_buffer('queue', F.normalize(torch.randn(dim, K), dim=0))
        self.register_buffer('queue_ptr', torch.zeros(1, dtype=torch.long))
        self.register_buffer('count', torch.zeros(K, dtype=torch.long))
        self.weight = None

    def _qk_pairs(self):
        for mq, mk in ((self.encoder_q, self.encoder_k), (self.neck_q, self.neck_k), (self.mlp_q, self.mlp_k)):
            yield from zip(mq.parameters(), mk.parameters())

    @torch.no_grad()
    def momentum_update(self):
        """ref: moco.py:408-421: parameters only, BN buffers are not averaged."""
        self.m = momentum_at(self.iters, self.max_iters, self.m_base)
        for q, k in self._qk_pairs():
            k.data = k.data * self.m + q.data * (1.0 - self.m)

    @torch.no_grad()
    def batch_shuffle(self, x):
        """ref: moco.py:146-172: gather, randperm on the default CPU generator, rank-0 broadcast."""
        b = x.shape[0]
        xg = concat_all_gather(x)
        perm = torch.randperm(xg.shape[0]).to(x.device)
        if _world() > 1:
            dist.broadcast(perm, src=0)
        inv = torch.argsort(perm)
        return xg[perm.view(-1, b)[_rank()]], inv

    @torch.no_grad()
    def batch_unshuffle(self, x, inv):
        """ref: moco.py:174-191."""
        b = x.shape[0]
        return concat_all_gather(x)[inv.view(-1, b)[_rank()]]

    @torch.no_grad()
    def dequeue_and_enqueue(self, keys):
        """ref: moco.py:423-440: count += 1 everywhere, write columns, new slots' count = 1."""
        keys = concat_all_gather(keys)
        self.count += 1
        n = keys.shape[0]
        self.batch_size = n
        ptr = int(self.queue_ptr)
        assert self.K % n == 0
        self.queue[:, ptr:ptr + n] = keys.T
        self.count[ptr:ptr + n] = 1
        self.queue_ptr[0] = (ptr + n) % self.K

    def extract_feat(self, im_q, im_k):
        """ref: moco.py:517-547."""
        q_emb, q_mlvl = self.neck_q(self.encoder_q(im_q))
        q = F.normalize(self.mlp_q(q_emb), dim=1)
        with torch.no_grad():
            self.momentum_update()
            im_k, inv = self.batch_shuffle(im_k)
            k_emb, k_mlvl = self.neck_k(self.encoder_k(im_k))
            k = F.normalize(self.mlp_k(k_emb), dim=1)
            k = self.batch_unshuffle(k, inv)
            k_mlvl = [self.batch_unshuffle(l, inv) for l in k_mlvl]
        return q, q_mlvl, k, k_mlvl

    def forward_train(self, im_q, im_k, update_queue=True):
        """ref: moco.py:473-515 (return_features=True path)."""
        q, q_mlvl, k, k_mlvl = self.extract_feat(im_q, im_k)
        l_pos = torch.einsum('nc,nc->n', [q, k]).unsqueeze(-1)
        snapshot = (self.queue * (0.99999 ** (1.0 * self.count))).clone().detach()   # moco.py:484-488
        self.weight = snapshot
        l_neg = torch.einsum('nc,ck->nk', [q, snapshot])
        logits = torch.cat([l_pos, l_neg], dim=1) / self.T
        labels = torch.zeros(logits.shape[0], dtype=torch.long, device=logits.device)
        if update_queue:
            self.dequeue_and_enqueue(k)
        if self.training:
            self.iters += self.batch_size
        losses = ce_with_topk(logits, labels, self.suffix)
        return losses, dict(q=q, q_mlvl=q_mlvl, k=k, k_mlvl=k_mlvl, q_neg=l_neg)


def cross_modal_logits(q, k, q_flow, k_flow, w_rgb, w_flow, T, same_kn=True):
    """ref: heads/moco_head_v2.py:38-53."""
    rf_pos = torch.einsum('nc,nc->n', [q, k_flow]).unsqueeze(-1)
    fr_pos = torch.einsum('nc,nc->n', [q_flow, k]).unsqueeze(-1)
    rf_neg = torch.einsum('nc,ck->nk', [q, w_flow if same_kn else w_rgb])
    fr_neg = torch.einsum('nc,ck->nk', [q_flow, w_rgb if same_kn else w_flow])
    rf = torch.cat([rf_pos, rf_neg], dim=1) / T
    fr = torch.cat([fr_pos, fr_neg], dim=1) / T
    return rf, fr, torch.zeros(rf.shape[0], dtype=torch.long, device=rf.device)


def lmcl_scores(rgb_map, flow_base_map, flow_aug_map, T, trans_flow=None):
    """LMCL similarity.  ref: heads/local_cl_head.py:57-73: spatial mean -> (flow: Conv1d 1x1 when bkb_channels[1] is set,
    :30-33; Identity for mscl_r18) -> L2 normalise over C -> bmm -> /T; labels arange(t) per clip."""
    xf = torch.cat((flow_base_map, flow_aug_map), dim=2)
    xr = F.normalize(F.adaptive_avg_pool3d(rgb_map, (None, 1, 1)).flatten(2), dim=1)   # b,c,t
    xf = F.adaptive_avg_pool3d(xf, (None, 1, 1)).flatten(2)                            # b,c,2t
    if trans_flow is not None:
        xf = trans_flow(xf)
    xf = F.normalize(xf, dim=1)
    sim = torch.bmm(xr.transpose(1, 2), xf)                      # b,t,2t
    t = xr.shape[2]
    labels = torch.arange(t, device=sim.device).unsqueeze(0).repeat(xr.shape[0], 1).flatten()
    return sim.flatten(0, 1) / T, labels


class MSCLWithAug(nn.Module):
    """ref: recognizers/mscl.py:158-277 for configs/recognition/moco/mscl_r18_cosm_lr2e-2.py."""

    def __init__(self, num_frames=8, K=65536, dim=128, m_base=0.994, max_iters=219136 * 400, T=0.07,
                 weight_aug_flow=(1.0, 1.0), update_aug_flow=False, same_kn=True, normalize_rgb=True, arch='r18'):
        """arch 'r18': mscl_r18_cosm_lr2e-2.py; 'r50': mscl_r50_cosm_lr3e-2.py:14-61 (SlowOnly-50 + r2d_50, sepc stride (1,2,2),
        one PConv, a Conv1d(256,128,1) on the flow side of the LMCL head)."""
        super().__init__()
        if arch == 'r18':
            neck = dict(in_channels=[128, 256, 512], out_channels=128,
                        sepc_cfg=dict(in_channels=[128, 128, 128], out_channels=128, stride=(2, 2, 2),
                                      iBN=False, Pconv_num=2))
            self.recognizer = MoCoV2('rgb', 512, dim, K, m_base, max_iters, T, neck=neck, basename='')
            self.recognizer_flow = MoCoV2('flow', 128, dim, K, m_base, max_iters, T, neck=None, basename='flow')
        elif arch == 'r50':
            neck = dict(in_channels=[512, 1024, 2048], out_channels=128,
                        sepc_cfg=dict(in_channels=[128, 128, 128], out_channels=128, stride=(1, 2, 2),
                                      iBN=False, Pconv_num=1))
            self.recognizer = MoCoV2('rgb50', 2048, dim, K, m_base, max_iters, T, neck=neck, basename='')
            self.recognizer_flow = MoCoV2('flow50', 256, dim, K, m_base, max_iters, T, neck=None, basename='flow')
        else:
            raise ValueError(arch)
        self.sup_head = nn.Module()
        if arch == 'r50':
            self.sup_head.trans_flow = nn.Conv1d(256, 128, 1)
        self.sup_head.register_buffer('labels', torch.arange(num_frames // 2).unsqueeze(0))
        self.T, self.same_kn = T, same_kn
        self.weight_aug_flow, self.update_aug_flow = weight_aug_flow, update_aug_flow
        self.normalize_rgb = normalize_rgb

    def aug(self, im_q, im_k):
        """Deterministic part of SyncMoCoAugmentV5 (common/ssl_aug_v2.py:66-68,90-97): ImageNet
        normalise on RGB only; flow views arrive visualised and flow_normalizer is Identity."""
        if not self.normalize_rgb:
            return im_q, im_k
        mean = torch.tensor(IMAGENET_MEAN, device=im_q.device).view(1, 3, 1, 1, 1)
        std = torch.tensor(IMAGENET_STD, device=im_q.device).view(1, 3, 1, 1, 1)
        return (im_q - mean) / std, (im_k - mean) / std

    def forward_train(self, im_q, im_k, flow_q, flow_k):
        """ref: mscl.py:225-277.  flow_q/flow_k are base||rotated flow along T (cat_flow=True)."""
        im_q, im_k = self.aug(im_q, im_k)
        loss_img, f_img = self.recognizer.forward_train(im_q, im_k)
        fq, fq_aug = (t.contiguous() for t in flow_q.chunk(2, 2))
        fk, fk_aug = (t.contiguous() for t in flow_k.chunk(2, 2))
        loss_flow, f_base = self.recognizer_flow.forward_train(fq, fk)
        loss_aug, f_aug = self.recognizer_flow.forward_train(fq_aug, fk_aug, update_queue=self.update_aug_flow)
        for key, v in loss_aug.items():
            if key.startswith('loss'):
                loss_flow[key + '_aug'] = v * self.weight_aug_flow[0]
        w_rgb, w_flow = self.recognizer.weight, self.recognizer_flow.weight      # mscl.py:247-248
        losses = OrderedDict()
        losses.update(loss_img)
        losses.update(loss_flow)
        rf, fr, lab = cross_modal_logits(f_img['q'], f_img['k'], f_base['q'], f_base['k'], w_rgb, w_flow,
                                         self.T, self.same_kn)
        losses.update(ce_with_topk(rf, lab, '_mx'))
        losses.update(ce_with_topk(fr, lab, '_mx_r'))
        if self.weight_aug_flow[1] > 0:
            rf, fr, lab = cross_modal_logits(f_img['q'], f_img['k'], f_aug['q'], f_aug['k'], w_rgb, w_flow,
                                             self.T, self.same_kn)
            losses.update(ce_with_topk(rf, lab, '_mx_aug'))
            losses.update(ce_with_topk(fr, lab, '_mx_r_aug'))
        scores, labels = lmcl_scores(f_img['q_mlvl'][0], f_base['q_mlvl'][-1], f_aug['q_mlvl'][-1], self.T,
                                     getattr(self.sup_head, 'trans_flow', None))
        losses['loss_pos'] = cross_entropy(scores, labels)
        acc = top_k_accuracy(scores.detach().float().cpu().numpy(), labels.cpu().numpy(), (1, 5))
        losses['top1_acc_pos'] = torch.tensor(acc[0], device=scores.device)
        losses['top5_acc_pos'] = torch.tensor(acc[1], device=scores.device)
        self._features = dict(img=f_img, base=f_base, aug=f_aug)
        return losses

    def train_step(self, data_batch, optimizer=None):
        """ref: mscl.py:192-212 + recognizers/base.py:274-308 (_parse_losses)."""
        im_q, im_k = data_batch['imgs']
        flow_q, flow_k = data_batch['flow_imgs']
        losses = self.forward_train(im_q, im_k, flow_q, flow_k)
        loss, log_vars = parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=im_q.shape[0])


def parse_losses(losses):
    """ref: recognizers/base.py:274-308: loss = sum of entries whose key contains 'loss'."""
    log_vars = OrderedDict((k, v.mean()) for k, v in losses.items())
    loss = sum(v for k, v in log_vars.items() if 'loss' in k)
    log_vars['loss'] = loss
    for k, v in log_vars.items():
        v = v.data.clone()
        if _world() > 1:
            dist.all_reduce(v.div_(_world()))
        log_vars[k] = v.item()
    return loss, log_vars


class SGDClip:
    """mmcv OptimizerHook + torch.optim.SGD as configured at mscl_r18_cosm_lr2e-2.py:114-119
    (wired at mmaction/apis/train.py:111-119).  mmcv is not vendored in the reference; its hook is
    `clip_grad_norm_(params with grad, max_norm=40, norm_type=2)` followed by `optimizer.step()`,
    so the restatement calls exactly those two PyTorch entry points: SGD(momentum 0.9, dampening 0,
    weight decay 1e-4 on every tensor, no nesterov); tensors whose grad is None are skipped entirely
    (no weight decay either)."""

    def __init__(self, params, lr=0.02, momentum=0.9, weight_decay=1e-4, max_norm=40.0):
        self.params = [p for p in params if p.requires_grad]
        self.max_norm = max_norm
        self.opt = torch.optim.SGD(self.params, lr=lr, momentum=momentum, weight_decay=weight_decay)

    def zero_grad(self):
        self.opt.zero_grad()

    def set_lr(self, lr):
        for g in self.opt.param_groups:
            g['lr'] = lr

    def step(self):
        with_grad = [p for p in self.params if p.grad is not None]
        total = torch.nn.utils.clip_grad_norm_(with_grad, max_norm=self.max_norm, norm_type=2)
        self.opt.step()
        return float(total)
