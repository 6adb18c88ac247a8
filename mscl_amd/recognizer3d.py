"""Downstream consumer of the pre-trained RGB encoder: supervised fine-tuning / evaluation /
feature extraction with the registry names of configs/recognition/ssl_test/test_ssv2_r18.py:10-28.

  Recognizer3D     ref: mmaction/models/recognizers/recognizer3d.py:9-96, recognizers/base.py:38-275
  I3DHead          ref: mmaction/models/heads/i3d_head.py:9-73, heads/base.py:42-118
  CrossEntropyLoss ref: mmaction/models/losses/cross_entropy_loss.py:9-86 (hard labels), losses/base.py

The trunk is the same `VideoResNetHip` the MSCL step trains (conv / BatchNorm / residual HIP kernels, training-mode and
evaluation-mode BatchNorm); global average pooling is the channel-mean kernel; the classifier is the fp32 linear kernel.
Dropout, the (N, classes) cross-entropy and top-k run as PyTorch ops on a few kilobytes.  Only the
'torchvision.r3d_18' backbone of the ssl_test configs is built.  There is no CPU fallback.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import kernels as K
from . import parallel
from .arena import ParamArena
from .lib import MsclError
from .nn import BatchNorm3dHip, Conv3dHip, LinearHip, VideoResNetHip, pool
from .registry import HEADS, LOSSES, RECOGNIZERS, build_head, build_loss


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    """hard-label branch of losses/cross_entropy_loss.py:60-86 times loss_weight (losses/base.py:33-45)"""

    def __init__(self, loss_weight=1.0, class_weight=None):
        super().__init__()
        self.loss_weight = loss_weight
        self.class_weight = None if class_weight is None else torch.tensor(class_weight, dtype=torch.float32)

    def forward(self, cls_score, label, **kwargs):
        if cls_score.size() == label.size():
            raise NotImplementedError('soft labels are not used by the ssl_test configs')
        w = None if self.class_weight is None else self.class_weight.to(cls_score.device)
        return F.cross_entropy(cls_score, label, weight=w, **kwargs) * self.loss_weight


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b on the fp32 linear kernels; dW / db accumulate straight into the gradient arena"""

    @staticmethod
    def forward(ctx, x, lin):
        x = x.contiguous()
        y = K.linear_fwd(x, lin._rt['w'], lin._rt['b'], False)
        ctx.lin = lin
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        rt = ctx.lin._rt
        dx = K.linear_bwd(x, rt['w'], y, dy.contiguous(), rt['dw'], rt['db'], False, need_dx=ctx.needs_input_grad[0])
        rt['slot_w'].touched = rt['slot_b'].touched = True
        return dx, None


@HEADS.register_module()
class I3DHead(nn.Module):
    def __init__(self, num_classes, in_channels, loss_cls=dict(type='CrossEntropyLoss'), spatial_type='avg', dropout_ratio=0.5,
                 init_std=0.01, multi_class=False, label_smooth_eps=0.0, **kwargs):
        super().__init__()
        if multi_class or label_smooth_eps:
            raise NotImplementedError('multi_class / label smoothing are not used by the ssl_test configs')
        self.num_classes, self.in_channels = num_classes, in_channels
        self.loss_cls = build_loss(dict(loss_cls))
        self.spatial_type, self.dropout_ratio, self.init_std = spatial_type, dropout_ratio, init_std
        self.fc_cls = LinearHip(in_channels, num_classes)
        self.init_weights()

    def init_weights(self):
        """ref: i3d_head.py:49-51 (normal_init std=init_std, bias 0)"""
        nn.init.normal_(self.fc_cls.weight, 0.0, self.init_std)
        nn.init.constant_(self.fc_cls.bias, 0.0)

    def forward(self, x):
        """x: pooled features (N, in_channels) fp32 -- with spatial_type='avg' the pooling happened in the recognizer
        (the same kernel); dropout then the classifier (i3d_head.py:53-73)"""
        if self.dropout_ratio:
            x = F.dropout(x, self.dropout_ratio, self.training)
        return _LinearFn.apply(x, self.fc_cls)

    def loss(self, cls_score, labels, **kwargs):
        """ref: heads/base.py:82-118"""
        losses = OrderedDict()
        if labels.shape == torch.Size([]):
            labels = labels.unsqueeze(0)
        k5 = min(5, cls_score.shape[1])
        top = cls_score.detach().topk(k5, dim=1).indices
        hit = top == labels.view(-1, 1)
        losses['top1_acc'] = hit[:, :1].any(1).float().mean()
        losses['top5_acc'] = hit.any(1).float().mean()
        losses['loss_cls'] = self.loss_cls(cls_score, labels, **kwargs)
        return losses


@RECOGNIZERS.register_module()
class Recognizer3D(nn.Module):
    def __init__(self, backbone, cls_head=None, neck=None, train_cfg=None, test_cfg=None):
        super().__init__()
        backbone = dict(backbone)
        typ = backbone.pop('type')
        if typ != 'torchvision.r3d_18' or neck is not None:
            raise NotImplementedError(f'only the torchvision.r3d_18 trunk of the ssl_test configs is built (got {typ}, neck={neck})')
        self.backbone_from = 'torchvision'
        self.backbone = VideoResNetHip('rgb')               # fc / classifier are Identity (recognizers/base.py:66-68)
        self.cls_head = build_head(dict(cls_head)) if cls_head else None
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.feature_extraction = bool(test_cfg and test_cfg.get('feature_extraction', False))
        self.max_testing_views = test_cfg.get('max_testing_views') if test_cfg else None
        self.arena, self._q_refresh, self._k_refresh, self._side = None, [], [], None

    with_neck = False

    @property
    def with_cls_head(self):
        return self.cls_head is not None

    # ---------------------------------------------------------------- device state
    def materialize(self, device):
        from .lib import load
        from .recognizers import MSCLWithAug
        device = torch.device(device)
        if device.type != 'cuda':
            raise MsclError('materialize() needs a GPU device: the HIP path has no CPU fallback')
        load()
        ar = ParamArena(device)
        ar.begin_group('all')
        plan = [(ar.add(n, p.shape), p) for n, p in self.named_parameters()]
        ar.end_group('all')
        ar.allocate(with_key=False)
        for slot, p in plan:
            v = ar.view('Q', slot)
            v.copy_(p.data.to(device))
            p.data, p.grad, p._mscl_slot = v, ar.view('G', slot), slot
        for mod in self.modules():
            for bname, buf in list(mod._buffers.items()):
                if buf is not None:
                    mod._buffers[bname] = buf.to(device)
        self.arena = ar
        self._q_refresh, self._k_refresh = [], []
        for m in self.modules():
            MSCLWithAug._bind(None, m, ar, False, self)
        entries = [(m._rt['w'], m._rt['wT'], m.out_channels, m.taps, m.in_channels) for m in self.modules()
                   if isinstance(m, Conv3dHip) and m._rt.get('wT') is not None]
        self._tr_table = K.build_transpose_table(entries, device)
        self.sync_shadows()
        return self

    @torch.no_grad()
    def sync_shadows(self):
        K.cast_bf16(self.arena.Q, self.arena.Qb)
        self.refresh_after_optimizer()

    @torch.no_grad()
    def refresh_after_optimizer(self):
        for fn in self._q_refresh:
            fn()
        K.weight_transpose_batched(*self._tr_table)

    def sync_streams(self):
        pass

    def load_state_dict(self, state_dict, strict=True):
        out = super().load_state_dict(state_dict, strict=strict)
        if self.arena is not None:
            self.sync_shadows()
        return out

    def init_from_ssl_pretrain(self, name, state_dict, ssl_cfg):
        """ref: recognizers/base.py:191-205: keep the keys under `prefix` (minus `extras`), strip it, load non-strictly.
        With prefix 'recognizer.encoder_q' this takes the RGB query encoder out of an MSCLWithAug checkpoint."""
        prefix, extras = ssl_cfg['prefix'], ssl_cfg.get('extras', ['fc'])
        sub = {k[len(prefix) + 1:]: v for k, v in state_dict.items()
               if k.startswith(prefix + '.') and not any(k.startswith(prefix + '.' + ex) for ex in extras)}
        missing, unexpected = getattr(self, name).load_state_dict(sub, strict=False)
        if self.arena is not None:
            self.sync_shadows()
        return missing, unexpected

    # ---------------------------------------------------------------- compute
    def extract_feat(self, imgs):
        """(N,3,T,H,W) fp32, already normalised by the data pipeline (test_ssv2_r18.py:37-38) -> (N,512) fp32: trunk,
        AdaptiveAvgPool3d(1), flatten -- what torchvision's VideoResNet.forward returns once fc is Identity"""
        if self.arena is None:
            raise MsclError('call model.materialize("cuda") first')
        m = self.backbone(K.pack_input(imgs.contiguous()))[-1]
        return pool(m, m.shape[0], m.shape[1] * m.shape[2] * m.shape[3])

    def forward_train(self, imgs, labels, **kwargs):
        """ref: recognizer3d.py:12-31"""
        assert self.with_cls_head
        imgs = imgs.reshape((-1,) + imgs.shape[2:])
        K.ZEROS.reset(imgs.device)
        cls_score = self.cls_head(self.extract_feat(imgs))
        return self.cls_head.loss(cls_score, labels.squeeze(-1) if labels.dim() > 1 else labels, **kwargs)

    @torch.no_grad()
    def _do_test(self, imgs):
        """ref: recognizer3d.py:33-96 (features for feature_extraction, else class scores averaged over the clips)"""
        num_segs = imgs.shape[1]
        imgs = imgs.reshape((-1,) + imgs.shape[2:])
        K.ZEROS.reset(imgs.device)
        step = self.max_testing_views or imgs.shape[0]
        feat = torch.cat([self.extract_feat(imgs[i:i + step]) for i in range(0, imgs.shape[0], step)])
        if self.feature_extraction:
            return feat
        assert self.with_cls_head
        return self.average_clip(self.cls_head(feat), num_segs)

    def average_clip(self, cls_score, num_segs=1):
        """ref: recognizers/base.py:224-256"""
        if 'average_clips' not in self.test_cfg:
            raise KeyError('"average_clips" must defined in test_cfg\'s keys')
        mode = self.test_cfg['average_clips']
        if mode not in ('score', 'prob', None):
            raise ValueError(f'{mode} is not supported. Currently supported ones are ["score", "prob", None]')
        if mode is None:
            return cls_score
        cls_score = cls_score.view(cls_score.shape[0] // num_segs, num_segs, -1)
        return F.softmax(cls_score, dim=2).mean(dim=1) if mode == 'prob' else cls_score.mean(dim=1)

    def forward_test(self, imgs):
        return self._do_test(imgs).cpu().numpy()

    def forward(self, imgs, label=None, return_loss=True, **kwargs):
        if return_loss:
            if label is None:
                raise ValueError('Label should not be None.')
            return self.forward_train(imgs, label, **kwargs)
        return self.forward_test(imgs, **kwargs)

    @staticmethod
    def _parse_losses(losses):
        """ref: recognizers/base.py:274-308"""
        log_vars = OrderedDict((k, v.mean()) for k, v in losses.items())
        loss = sum(v for k, v in log_vars.items() if 'loss' in k)
        log_vars['loss'] = loss
        out = OrderedDict()
        for k, v in log_vars.items():
            v = v.detach().clone()
            if not parallel.single():
                torch.distributed.all_reduce(v.div_(parallel.world_size()))
            out[k] = v.item()
        return loss, out

    def train_step(self, data_batch, optimizer=None, **kwargs):
        """ref: recognizers/base.py:327-367"""
        loss, log_vars = self._parse_losses(self(data_batch['imgs'], data_batch['label'], return_loss=True))
        return dict(loss=loss, log_vars=log_vars, num_samples=len(next(iter(data_batch.values()))))

    val_step = train_step
