import os, sys, torch
sys.path.insert(0, '/root/repo')
from mscl_amd import ClipSGD, Config, build_model
from mscl_amd.fill import fill_module
from mscl_amd.synthetic import synthetic_batch
dev = torch.device('cuda', 0)
cfg = Config.fromfile('/root/repo/configs/recognition/moco/mscl_r18_cosm_lr2e-2.py')
cfg.model.sup_head.t = 4
cfg.model.recognizer.K = 256; cfg.model.recognizer_flow.K = 256
m = build_model(cfg.model); fill_module(m); m.materialize(dev).train()
opt = ClipSGD.from_cfg(m, cfg.optimizer, cfg.optimizer_config)
batches = [synthetic_batch(4, 8, 64, 64, 0, s, device=dev) for s in range(4)]
for it in range(120):
    out = m.train_step(batches[it % 4])
    opt.zero_grad(); out['loss'].backward(); opt.step()
    if it % 10 == 0 or it == 119:
        lv = out['log_vars']
        print(it, round(lv['loss'], 3), round(lv['loss_cls'], 3), round(lv['loss_cls_flow'], 3), round(lv['loss_pos'], 3), round(lv['top1_acc'], 2), float(opt.grad_norm()))
