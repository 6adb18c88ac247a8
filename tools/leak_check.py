import os, sys, torch
sys.path.insert(0, '/root/repo')
from mscl_amd import ClipSGD, Config, build_model
from mscl_amd.fill import fill_module
from mscl_amd.graph import GraphedStep
from mscl_amd.synthetic import synthetic_batch
dev = torch.device('cuda', 0)
cfg = Config.fromfile('/root/repo/configs/recognition/moco/mscl_r18_cosm_lr2e-2.py')
cfg.model.sup_head.t = 8
m = build_model(cfg.model); fill_module(m); m.materialize(dev).train()
opt = ClipSGD.from_cfg(m, cfg.optimizer, cfg.optimizer_config)
batches = [synthetic_batch(8, 16, 112, 112, 0, s, device=dev) for s in range(2)]
def eager(i):
    out = m.train_step(batches[i % 2], sync_logs=False); opt.zero_grad(); out['loss'].backward(); opt.step()
for i in range(5): eager(i)
torch.cuda.synchronize(); a0 = torch.cuda.memory_allocated(); r0 = torch.cuda.memory_reserved()
for i in range(200): eager(i)
torch.cuda.synchronize(); a1 = torch.cuda.memory_allocated(); r1 = torch.cuda.memory_reserved()
print('eager 200 steps: allocated %.1f -> %.1f MB, reserved %.1f -> %.1f MB' % (a0/2**20, a1/2**20, r0/2**20, r1/2**20))
gs = GraphedStep(m, opt, batches[0], warmup=2)
for i in range(5): gs.step(batches[i % 2])
torch.cuda.synchronize(); a0 = torch.cuda.memory_allocated(); r0 = torch.cuda.memory_reserved()
for i in range(300): loss, _ = gs.step(batches[i % 2])
torch.cuda.synchronize(); a1 = torch.cuda.memory_allocated(); r1 = torch.cuda.memory_reserved()
print('graph 300 steps: allocated %.1f -> %.1f MB, reserved %.1f -> %.1f MB, loss %.3f' % (a0/2**20, a1/2**20, r0/2**20, r1/2**20, float(loss)))
print('queue_ptr', int(m.recognizer.queue_ptr), 'iters', m.recognizer.iters)
