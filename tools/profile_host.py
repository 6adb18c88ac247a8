"""cProfile of the eager step's host side (launch overhead decides the world-size > 1 path, which does not use a HIP graph)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mscl_amd import ClipSGD, Config, build_model          # noqa: E402
from mscl_amd.fill import fill_module                       # noqa: E402
from mscl_amd.synthetic import synthetic_batch              # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device('cuda', 0)
if os.environ.get('MSCL_FORCE_DIST') == '1':            # profile the world-size > 1 host path on a 1-rank RCCL group
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29534')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
cfg = Config.fromfile(os.path.join(ROOT, 'configs/recognition/moco/mscl_r18_cosm_lr2e-2.py'))
cfg.model.sup_head.t = 8
model = build_model(cfg.model); fill_module(model); model.materialize(dev).train()
opt = ClipSGD.from_cfg(model, cfg.optimizer, cfg.optimizer_config)
batch = synthetic_batch(8, 16, 112, 112, 0, 0, device=dev)


def step():
    out = model.train_step(batch, sync_logs=False)
    opt.zero_grad(); out['loss'].backward(); opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats(os.environ.get('SORT', 'tottime')).print_stats(int(os.environ.get('ROWS', '28')))
