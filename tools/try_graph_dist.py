"""Can the whole step, INCLUDING its RCCL collectives, be captured into one HIP graph?  One-rank NCCL group with the
collectives forced on (MSCL_FORCE_DIST=1); prints per-step time for eager and graph."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ.get('MASTER_PORT', '29533'), MSCL_FORCE_DIST='1')
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
from mscl_amd import ClipSGD, Config, build_model          # noqa: E402
from mscl_amd.fill import fill_module                       # noqa: E402
from mscl_amd.graph import GraphedStep                      # noqa: E402
from mscl_amd.synthetic import synthetic_batch              # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config.fromfile(os.path.join(ROOT, 'configs/recognition/moco/mscl_r18_cosm_lr2e-2.py'))
cfg.model.sup_head.t = 8
model = build_model(cfg.model); fill_module(model); model.materialize(dev).train()
opt = ClipSGD.from_cfg(model, cfg.optimizer, cfg.optimizer_config)
batches = [synthetic_batch(8, 16, 112, 112, 0, s, device=dev) for s in range(4)]
mode = sys.argv[1] if len(sys.argv) > 1 else 'graph'
if mode == 'graph':
    gs = GraphedStep(model, opt, batches[0], warmup=2)
    step = lambda i: gs.step(batches[i % 4])[0]
else:
    def step(i):
        out = model.train_step(batches[i % 4], sync_logs=False)
        opt.zero_grad(); out['loss'].backward(); opt.step()
        return out['loss']
for i in range(5):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    loss = step(i)
torch.cuda.synchronize()
print(f'{mode}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step, loss {float(loss):.4f}, queue_ptr {int(model.recognizer.queue_ptr)}', flush=True)
dist.destroy_process_group()
