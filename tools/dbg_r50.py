"""HIP vs oracle on the mscl_r50 step at a golden's shape: log entries side by side, feature and stage-map cosines.
usage: python tools/dbg_r50.py [H] [B] [T]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mscl_amd.synthetic import synthetic_batch          # noqa: E402
from oracle import fill as ofill, mscl as om             # noqa: E402
import test_model_gpu as tm                              # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
T = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device('cuda:0')
model, cfg = tm.build(T, 65536, dev, 'r50')
orc = om.MSCLWithAug(num_frames=T, K=65536, arch='r50'); ofill.fill_module(orc); orc.train()
batch = synthetic_batch(B, T, H, H, 0, 0)
cos = torch.nn.functional.cosine_similarity
# trunks alone first
mean = torch.tensor((0.485, 0.456, 0.406)).view(1, 3, 1, 1, 1); std = torch.tensor((0.229, 0.224, 0.225)).view(1, 3, 1, 1, 1)
x = batch['imgs'][0]
maps = model.recognizer.encoder_q(model.aug_gpu.pack_rgb(x.to(dev)))
omaps = orc.recognizer.encoder_q((x - mean) / std)
for li, (a, b) in enumerate(zip(maps, omaps)):
    a = a.detach().float().cpu().permute(0, 4, 1, 2, 3)
    print(f'rgb layer{li + 1}', tuple(a.shape), 'cos %.5f' % float(cos(a.flatten(), b.detach().flatten(), dim=0)))
fq = batch['flow_imgs'][0]
Th = fq.shape[2] // 2
xf = model.aug_gpu.pack_flow(fq.to(dev), 0, Th, None)
fmaps = model.recognizer_flow.encoder_q(xf)
fx = fq[:, :, :Th].contiguous()          # synthetic flow views arrive visualised (3 channels); no normalisation on the flow side
if fx is not None:
    ofm = orc.recognizer_flow.encoder_q(fx)
    for li, (a, b) in enumerate(zip(fmaps, ofm)):
        a = a.detach().float().cpu().permute(0, 4, 1, 2, 3)
        print(f'flow layer{li + 1}', tuple(a.shape), 'cos %.5f' % float(cos(a.flatten(), b.detach().flatten(), dim=0)))
# BN running stats were touched by the probes above: rebuild both sides for the step
model, cfg = tm.build(T, 65536, dev, 'r50')
orc = om.MSCLWithAug(num_frames=T, K=65536, arch='r50'); ofill.fill_module(orc); orc.train()
out = model.train_step({k: [t.to(dev) for t in v] for k, v in batch.items()})
torch.manual_seed(100)
oo = orc.train_step(batch)
for k, v in oo['log_vars'].items():
    print(f'{k:24s} hip {out["log_vars"][k]:12.6f}  oracle {v:12.6f}')
for nm, a, b in (('q_rgb', model._dbg['q_rgb'], orc._features['img']['q']), ('k_rgb', model._dbg['k_rgb'], orc._features['img']['k']),
                 ('q_flow', model._dbg['q_fb'], orc._features['base']['q']), ('k_flow', model._dbg['k_fb'], orc._features['base']['k']),
                 ('q_flow_aug', model._dbg['q_fa'], orc._features['aug']['q']), ('k_flow_aug', model._dbg['k_fa'], orc._features['aug']['k'])):
    print(nm, 'row cosines', [round(float(c), 5) for c in cos(a.float().cpu(), b.detach(), dim=1)])
