"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the supervised consumer of the pre-trained RGB encoder
(SURVEY.md §8(f)#4): Recognizer3D + I3DHead + CrossEntropyLoss as configured by
configs/recognition/ssl_test/test_ssv2_r18.py:10-28.  Never imported by the product path.

Pinned against the reference's own classes run in the development container (tools/oracle/make_golden_finetune.py:
mmaction/models/recognizers/recognizer3d.py, heads/i3d_head.py, heads/base.py, losses/cross_entropy_loss.py with the
vendored R3D trunk, spatial_type='avg'): losses, scores and gradient norms are bit-identical on the closed-form weights,
outputs in tests/golden/finetune_g9.json.  Unpinned: torchvision's own r3d_18 wrapper (avgpool + flatten + Identity fc),
which the config names and which is absent here; it is the same arithmetic as the head's spatial_type='avg' pooling.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from .mscl import parse_losses, top_k_accuracy
from .nets import VideoResNet18


class I3DHead(nn.Module):
    """ref: heads/i3d_head.py:27-73 (pool -> dropout -> fc_cls), heads/base.py:82-118 (top-k then CE)"""

    def __init__(self, num_classes, in_channels=512, dropout_ratio=0.5, init_std=0.01, loss_weight=1.0):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout_ratio) if dropout_ratio else None
        self.fc_cls = nn.Linear(in_channels, num_classes)
        nn.init.normal_(self.fc_cls.weight, 0, init_std)
        nn.init.constant_(self.fc_cls.bias, 0)
        self.loss_weight = loss_weight

    def forward(self, x):
        x = x.mean(dim=(2, 3, 4))                      # AdaptiveAvgPool3d((1,1,1)) + view(N, -1)
        if self.dropout is not None:
            x = self.dropout(x)
        return self.fc_cls(x)

    def loss(self, cls_score, labels):
        out = OrderedDict()
        acc = top_k_accuracy(cls_score.detach().float().cpu().numpy(), labels.detach().cpu().numpy(), (1, 5))
        out['top1_acc'] = torch.tensor(acc[0])
        out['top5_acc'] = torch.tensor(acc[1])
        out['loss_cls'] = F.cross_entropy(cls_score, labels) * self.loss_weight      # cross_entropy_loss.py:60-86, losses/base.py:33-45
        return out


class Recognizer3D(nn.Module):
    def __init__(self, num_classes, dropout_ratio=0.5, average_clips='prob'):
        super().__init__()
        self.backbone = VideoResNet18('rgb')
        self.cls_head = I3DHead(num_classes, 512, dropout_ratio)
        self.average_clips = average_clips

    def forward_train(self, imgs, labels):
        """ref: recognizer3d.py:12-31; imgs (B, clips, 3, T, H, W)"""
        imgs = imgs.reshape((-1,) + imgs.shape[2:])
        cls_score = self.cls_head(self.backbone(imgs)[-1])
        return self.cls_head.loss(cls_score, labels.squeeze(-1) if labels.dim() > 1 else labels)

    @torch.no_grad()
    def forward_test(self, imgs, feature_extraction=False):
        """ref: recognizer3d.py:33-96, base.py:224-256"""
        num_segs = imgs.shape[1]
        imgs = imgs.reshape((-1,) + imgs.shape[2:])
        feat = self.backbone(imgs)[-1]
        if feature_extraction:
            return feat.mean(dim=(2, 3, 4))
        s = self.cls_head(feat)
        s = s.view(s.shape[0] // num_segs, num_segs, -1)
        if self.average_clips == 'prob':
            return F.softmax(s, dim=2).mean(dim=1)
        return s.mean(dim=1) if self.average_clips == 'score' else s.view(-1, s.shape[-1])

    def train_step(self, data_batch):
        loss, log_vars = parse_losses(self.forward_train(data_batch['imgs'], data_batch['label']))
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data_batch['imgs']))
