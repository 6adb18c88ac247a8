"""G9: the reference's Recognizer3D + I3DHead (the fine-tune / evaluation consumer) against
oracle/recognizer3d.py on closed-form weights; writes tests/golden/finetune_g9.json.  Development container only."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness                                           # noqa: E402
from oracle import fill as ofill                             # noqa: E402
from oracle import recognizer3d as orec                      # noqa: E402

h = ref_harness.install()
import importlib                                             # noqa: E402
importlib.import_module('mmaction.models.heads.base')
importlib.import_module('mmaction.models.heads.i3d_head')
importlib.import_module('mmaction.models.recognizers.recognizer3d')
NUM_CLASSES, B, CLIPS, T, H = 10, 2, 2, 8, 32
cfg = dict(type='Recognizer3D', backbone=dict(type='torchvision.r3d_18'),
           cls_head=dict(type='I3DHead', num_classes=NUM_CLASSES, in_channels=512, spatial_type='avg', dropout_ratio=0.0),
           test_cfg=dict(average_clips='prob'))
ref = h['builder'].build_recognizer(cfg)
ofill.fill_module(ref)
ora = orec.Recognizer3D(NUM_CLASSES, dropout_ratio=0.0)
ofill.fill_module(ora)
assert list(ref.state_dict()) == list(ora.state_dict()), 'state-dict names differ'
for (k, a), (_, b) in zip(ref.state_dict().items(), ora.state_dict().items()):
    assert torch.equal(a, b), k
g = torch.Generator().manual_seed(91)
imgs = torch.randn((B, 1, 3, T, H, H), generator=g)           # training: one clip per sample (test_ssv2_r18.py:39-48)
test_imgs = torch.randn((B, CLIPS, 3, T, H, H), generator=g)  # testing: several clips, probabilities averaged
label = torch.tensor([[3], [7]])
out = {}
for name, m in (('ref', ref), ('oracle', ora)):
    m.train()
    m.zero_grad()
    losses = m.forward_train(imgs, label)
    loss, log_vars = (m._parse_losses(losses) if name == 'ref' else orec.parse_losses(losses))
    loss.backward()
    gn = torch.sqrt(sum((p.grad ** 2).sum() for p in m.parameters() if p.grad is not None)).item()
    gfc = m.cls_head.fc_cls.weight.grad.norm().item()
    gstem = m.backbone.stem[0].weight.grad.norm().item()
    m.eval()
    with torch.no_grad():
        probs = torch.as_tensor(m.forward_test(test_imgs)).float()
    out[name] = dict(log_vars={k: float(v) for k, v in log_vars.items()}, grad_norm=gn, grad_fc=gfc, grad_stem=gstem,
                     probs=probs.tolist(), running_mean_stem=m.backbone.stem[1].running_mean[:4].tolist())
for k in out['ref']:
    assert out['ref'][k] == out['oracle'][k], (k, out['ref'][k], out['oracle'][k])
print('reference and oracle agree exactly:', json.dumps(out['ref']['log_vars']))
fixture = dict(config=dict(num_classes=NUM_CLASSES, B=B, clips=CLIPS, T=T, H=H, seed=91, labels=[3, 7], dropout_ratio=0.0),
               **out['ref'])
with open(os.path.join(ROOT, 'tests/golden/finetune_g9.json'), 'w') as f:
    json.dump(fixture, f, indent=1)
print('wrote tests/golden/finetune_g9.json')
