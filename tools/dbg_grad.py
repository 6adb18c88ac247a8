import sys, os, json, numpy as np, torch
sys.path.insert(0, '/root/repo')
from tests.test_model_gpu import build
from mscl_amd import ClipSGD
from mscl_amd.synthetic import synthetic_batch
from oracle import fill as ofill, mscl as om
dev = torch.device('cuda:0')
B, T, H, Kq = 2, 8, 112, 65536
model, cfg = build(T, Kq, dev)
orc = om.MSCLWithAug(num_frames=T, K=Kq); ofill.fill_module(orc); orc.train()
batch = synthetic_batch(B, T, H, H, 0, 0)
out = model.train_step({k: [t.to(dev) for t in v] for k, v in batch.items()})
model.zero_grad(); out['loss'].backward()
oo = orc.train_step(batch)
f = orc._features
for nm in ('img', 'base', 'aug'):
    f[nm]['q'].retain_grad()
    for m in f[nm]['q_mlvl']: m.retain_grad()
oo['loss'].backward()
cos = torch.nn.functional.cosine_similarity
for (n, p), (n2, q) in zip(model.named_parameters(), orc.named_parameters()):
    if not p.requires_grad or q.grad is None: continue
    gh = p.grad.detach().float().cpu(); go = q.grad
    print(f'{n:70s} cos {float(cos(gh.flatten(), go.flatten(), dim=0)):.4f}  |hip| {float(gh.norm()):.4e} |orc| {float(go.norm()):.4e}')
