"""Loss and head classes with the reference's registry names / constructor kwargs.

On the HIP path the (N, 65537) logits are never materialised: the queue pass (kernels.nce_forward)
returns per-row losses and ranks, and the heads turn those into the reference's log entries
(`loss_cls*`, `top1_acc*`, `top5_acc*`).  Each head also keeps the reference's logits-based
signature for small, explicit score tables.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from .registry import HEADS, LOSSES, build_loss


@LOSSES.register_module()
class CrossEntropyLoss_torch(nn.Module):
    """ref: losses/cross_entropy_loss.py:122-138 (torch CrossEntropyLoss with a loss_weight)."""

    def __init__(self, weight=None, size_average=None, ignore_index=-100, reduce=None, reduction='mean', loss_weight=1.0):
        super().__init__()
        if weight is not None or reduction != 'mean':
            raise NotImplementedError('only unweighted mean reduction is used by the MSCL configs')
        self.ignore_index, self.reduction, self.loss_weight = ignore_index, reduction, loss_weight

    def forward(self, input, target):
        return self.loss_weight * F.cross_entropy(input, target, ignore_index=self.ignore_index, reduction=self.reduction)

    def from_rows(self, loss_rows):
        """mean CE given per-row losses from the streaming pass (labels are all the positive column)."""
        return self.loss_weight * loss_rows.mean()


def rank_topk(rank, ks=(1, 5)):
    """top-k accuracy from the number of negatives that beat the positive: hit iff rank < k.
    Equals core/evaluation/accuracy.py:130-149 (argsort-based) except on exact ties."""
    return [(rank < k).float().mean() for k in ks]


def logits_topk(scores, labels, ks=(1, 5)):
    pos = scores.gather(1, labels[:, None])
    return rank_topk((scores > pos).sum(1), ks)


class _BaseHead(nn.Module):
    def __init__(self, loss_cls, num_classes=2, in_channels=128):
        super().__init__()
        self.num_classes, self.in_channels = num_classes, in_channels
        self.loss_cls = build_loss(loss_cls)
        self.multi_class, self.label_smooth_eps = False, 0.0

    def init_weights(self):
        pass

    def _entries(self, loss, top1, top5, basename):
        out = OrderedDict()
        out[f'top1_acc{basename}'] = top1
        out[f'top5_acc{basename}'] = top5
        out[f'loss_cls{basename}'] = loss
        return out

    def loss_streamed(self, loss_rows, rank, basename):
        t1, t5 = rank_topk(rank)
        return self._entries(self.loss_cls.from_rows(loss_rows), t1, t5, basename)

    def loss_logits(self, cls_score, labels, basename):
        t1, t5 = logits_topk(cls_score.detach(), labels)
        return self._entries(self.loss_cls(cls_score, labels), t1, t5, basename)


@HEADS.register_module()
class MoCoHead(_BaseHead):
    """ref: heads/moco_head.py:9-81."""

    def __init__(self, basename='', loss_cls=dict(type='CrossEntropyLoss'), num_classes=2, in_channels=128):
        super().__init__(loss_cls, num_classes, in_channels)
        self.basename = ('_' + basename) if basename else ''

    def forward(self, **kwargs):
        return dict()

    def loss(self, cls_score, labels, basename=None, **kwargs):
        return self.loss_logits(cls_score, labels, self.basename if basename is None else basename)


@HEADS.register_module()
class MSCLWithAugMxHead(_BaseHead):
    """ref: heads/moco_head_v2.py:15-106 (cross-modal rgb<->flow InfoNCE)."""

    def __init__(self, basename='', loss_cls=dict(type='CrossEntropyLoss'), num_classes=2, in_channels=128,
                 same_kn=True, T=0.07):
        super().__init__(loss_cls, num_classes, in_channels)
        self.basename = ('_' + basename) if basename else ''
        self.same_kn, self.T = same_kn, T

    def loss(self, rf_logits, fr_logits, ssl_label, suffix=''):
        out = self.loss_logits(rf_logits, ssl_label, self.basename + suffix)
        out.update(self.loss_logits(fr_logits, ssl_label, self.basename + '_r' + suffix))
        return out


@HEADS.register_module()
class MSCLWithAugPosHeadV2(_BaseHead):
    """LMCL head.  ref: heads/local_cl_head.py:10-81 (bkb_channels=(None, None) -> identity transforms; (None, C) -> a
    Conv1d(C, 128, 1) on the flow side, the mscl_r50 configuration)."""

    def __init__(self, basename='', loss_cls=dict(type='CrossEntropyLoss_torch'), loss_pos=dict(type='CrossEntropyLoss_torch'),
                 num_classes=2, in_channels=128, mlvl_ids=(0, -1), bkb_channels=(512, 128), t=8, T=0.07, aux_keys=dict()):
        super().__init__(loss_cls, num_classes, in_channels)
        if bkb_channels[0] is not None:
            raise NotImplementedError('the RGB-side Conv1d transform (bkb_channels[0]) is not used by the MSCL configs')
        if bkb_channels[1] is not None:       # mscl_r50: Conv1d(256, 128, 1) on the pooled flow frames (local_cl_head.py:30-33,65)
            from .nn import Conv1dK1Hip
            self.trans_flow = Conv1dK1Hip(bkb_channels[1], 128)
        self.loss_pos = build_loss(loss_pos)
        self.basename = ('_' + basename) if basename else ''
        self.T, self.aux_keys, self.mlvl_ids, self.t = T, dict(aux_keys), tuple(mlvl_ids), t
        self.register_buffer('labels', torch.arange(t).unsqueeze(0))

    def update_aux_info(self, info_name, info_dict, target):
        """ref: local_cl_head.py:75-81 (incl. the key-collision assert)."""
        if info_name in self.aux_keys:
            for k in self.aux_keys[info_name]:
                dst = self.aux_keys[info_name][k]
                assert dst not in target, f'Find key-{dst} in target dict with keys:{target.keys()}'
                target[dst] = info_dict[k]
        return target
