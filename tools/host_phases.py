"""Host-side duration of the three phases of an eager step (no device sync inside): where the launch overhead sits."""
import os
import sys
import time

import torch
if os.environ.get('HANG_DUMP'):
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ['HANG_DUMP']), exit=True)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mscl_amd import ClipSGD, Config, build_model          # noqa: E402
from mscl_amd.fill import fill_module                       # noqa: E402
from mscl_amd.synthetic import synthetic_batch              # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device('cuda', 0)
if os.environ.get('MSCL_FORCE_DIST') == '1':            # the world-size > 1 host path on a 1-rank RCCL group
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29535')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
cfg = Config.fromfile(os.path.join(ROOT, 'configs/recognition/moco/mscl_r18_cosm_lr2e-2.py'))
cfg.model.sup_head.t = 8
if os.environ.get('HANG_DUMP'):
    import traceback
    import torch.distributed as dist
    for name in ('all_reduce', 'all_gather_into_tensor', 'all_to_all_single', 'barrier', 'all_gather'):
        def wrap(fn, name=name):
            def inner(*a, **k):
                cap = torch.cuda.is_current_stream_capturing()
                print('[coll]', name, 'capturing' if cap else '', 'stream', hex(torch.cuda.current_stream().cuda_stream), flush=True)
                if cap:
                    traceback.print_stack(limit=6)
                return fn(*a, **k)
            return inner
        setattr(dist, name, wrap(getattr(dist, name)))
    import mscl_amd.recognizers as R
    oc = R.QueryGraph.capture
    def cap2(self, *a, **k):
        print('[capture] QueryGraph begin, stream', hex(torch.cuda.current_stream().cuda_stream), flush=True)
        r = oc(self, *a, **k)
        print('[capture] QueryGraph end', flush=True)
        return r
    R.QueryGraph.capture = cap2
model = build_model(cfg.model); fill_module(model); model.materialize(dev).train()
opt = ClipSGD.from_cfg(model, cfg.optimizer, cfg.optimizer_config)
batch = synthetic_batch(8, 16, 112, 112, 0, 0, device=dev)
acc = [0.0, 0.0, 0.0]
for it in range(25):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.train_step(batch, sync_logs=False)
    t1 = time.perf_counter()
    opt.zero_grad(); out['loss'].backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    if it >= 5:
        acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(20):
    out = model.train_step(batch, sync_logs=False)
    opt.zero_grad(); out['loss'].backward(); opt.step()
torch.cuda.synchronize()
print('wall ms/step %.2f' % (1e3 * (time.perf_counter() - t0) / 20))
print('stream probe', getattr(model, 'stream_probe', None))
print('host ms/step: forward %.2f  backward %.2f  optimizer %.2f' % tuple(1e3 * a / 20 for a in acc))
