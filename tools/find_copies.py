"""Which framework-level copies run inside one training step (torch profiler, shapes + Python stack)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mscl_amd import ClipSGD, Config, build_model
from mscl_amd.fill import fill_module
from mscl_amd.synthetic import synthetic_batch
from torch.profiler import profile, ProfilerActivity
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device('cuda', 0)
cfg = Config.fromfile(os.path.join(ROOT, 'configs/recognition/moco/mscl_r18_cosm_lr2e-2.py'))
cfg.model.sup_head.t = 8
model = build_model(cfg.model); fill_module(model); model.materialize(dev).train()
model.key_graphs = False
opt = ClipSGD.from_cfg(model, cfg.optimizer, cfg.optimizer_config)
batch = synthetic_batch(8, 16, 112, 112, 0, 0, device=dev)
def step():
    out = model.train_step(batch, sync_logs=False)
    opt.zero_grad(); out['loss'].backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
from torch.autograd import DeviceType
kern = [e for e in prof.events() if e.device_type == DeviceType.CUDA and 'elementwise_kernel_manual_unroll' in e.name]
kern.sort(key=lambda e: -e.device_time_total)
print('strided-copy style kernels:', len(kern))
for e in kern[:12]:
    print('  %8.1f us  %s' % (e.device_time_total, e.name[:110]))
print('large aten::copy_ / contiguous / clone / cat / index_select calls:')
for e in prof.events():
    if e.device_type == DeviceType.CPU and e.name in ('aten::copy_', 'aten::contiguous', 'aten::clone', 'aten::cat', 'aten::index_select', 'aten::repeat', 'aten::to', 'aten::_to_copy'):
        shp = e.input_shapes
        n = 1
        for d in (shp[0] if shp and shp[0] else []):
            n *= d
        if n >= 1 << 20:
            print('  %-18s %-60s %s' % (e.name, str(shp)[:60], [s.split('/')[-1] for s in (e.stack or []) if 'mscl_amd' in s][:3]))
