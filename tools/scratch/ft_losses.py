"""loss trajectories of the fine-tune loop of tests/test_finetune_gpu.py::test_ssl_pretrain_loading_feature_extraction_and_finetuning"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_finetune_gpu import build
from mscl_amd import ClipSGD
dev = torch.device('cuda:0')
for rep in range(8):
    model = build(7, 0.5, dev, test_cfg=dict(average_clips='score', feature_extraction=True))
    g = torch.Generator().manual_seed(3)
    imgs = torch.randn((4, 1, 3, 8, 32, 32), generator=g).to(dev)
    label = torch.tensor([[0], [3], [6], [3]], device=dev)
    model.feature_extraction = False
    opt = ClipSGD(model, lr=0.05, momentum=0.9, weight_decay=1e-4, grad_clip=dict(max_norm=40, norm_type=2))
    model.train()
    losses = []
    for _ in range(12):
        out = model.train_step(dict(imgs=imgs, label=label))
        opt.zero_grad(); out['loss'].backward(); opt.step()
        losses.append(round(float(out['log_vars']['loss_cls']), 3))
    print(rep, 'ratio %.3f' % (losses[-1] / losses[0]), losses, flush=True)
