"""Long-run stability: N eager steps (sub-graph replay) then N whole-step-graph replays at the benchmark size; loss must stay
finite, allocator footprint flat.  MSCL_FORCE_DIST=1 adds every collective on a one-rank RCCL group."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mscl_amd import ClipSGD, Config, build_model
from mscl_amd.fill import fill_module
from mscl_amd.graph import GraphedStep
from mscl_amd.synthetic import synthetic_batch
N = int(os.environ.get('SOAK_STEPS', '1000'))
dev = torch.device('cuda', 0)
if os.environ.get('MSCL_FORCE_DIST') == '1':
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29541')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config.fromfile(os.path.join(ROOT, 'configs/recognition/moco/mscl_r18_cosm_lr2e-2.py'))
cfg.model.sup_head.t = 8
m = build_model(cfg.model); fill_module(m); m.materialize(dev).train()
if os.environ.get('SOAK_AUG') == '1':
    m.aug_gpu.stochastic = True
opt = ClipSGD.from_cfg(m, cfg.optimizer, cfg.optimizer_config)
batches = [synthetic_batch(8, 16, 112, 112, 0, s, device=dev) for s in range(8)]
def report(tag, losses, t0):
    torch.cuda.synchronize()
    ls = torch.stack(losses).float().cpu()
    print('%s: %d steps, %.2f ms/step, loss first %.3f last %.3f min %.3f max %.3f finite %s, allocated %.0f MB reserved %.0f MB' % (
        tag, len(losses), 1e3 * (time.perf_counter() - t0) / len(losses), ls[0], ls[-1], ls.min(), ls.max(), bool(torch.isfinite(ls).all()),
        torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20), flush=True)
    print('   every 250th:', ' '.join('%.1f' % v for v in ls[::250].tolist()), flush=True)
    assert torch.isfinite(ls).all()
MODE = os.environ.get('SOAK_MODE', 'both')
if MODE in ('both', 'eager'):
    losses = []
    t0 = time.perf_counter()
    for i in range(N):
        out = m.train_step(batches[i % 8], sync_logs=False); opt.zero_grad(); out['loss'].backward(); opt.step()
        losses.append(out['loss'].detach())
    report('eager + sub-graphs', losses, t0)
if (os.environ.get('MSCL_FORCE_DIST') != '1' or os.environ.get('MSCL_GRAPH_DP') == '1') and MODE in ('both', 'graph'):
    gs = GraphedStep(m, opt, batches[0], warmup=2)
    losses = []
    t0 = time.perf_counter()
    for i in range(N):
        loss, _ = gs.step(batches[i % 8])
        losses.append(loss.clone())
    report('whole-step graph', losses, t0)
print('queue_ptr', int(m.recognizer.queue_ptr), 'iters', m.recognizer.iters, 'sub-graphs', sum(g.graph is not None for g in m._key_graph) + sum(g.fwd is not None for g in m.active_query_graphs()))
