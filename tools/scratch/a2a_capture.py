import faulthandler, os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29577', MSCL_FORCE_DIST='1')
log = open('gpurun_out/a2a/progress.log', 'w')
faulthandler.dump_traceback_later(50, file=log, exit=True)
def say(*a):
    print(*a, file=log, flush=True); print(*a, flush=True)
import torch, torch.distributed as dist
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
import test_model_gpu as tm
from mscl_amd import ClipSGD, lib, parallel
from mscl_amd.graph import GraphedStep
from mscl_amd.synthetic import synthetic_batch
mode = sys.argv[1] if len(sys.argv) > 1 else 'a2a'
det = len(sys.argv) > 2 and sys.argv[2] == 'det'
if det: lib.set_deterministic(True)
B, T, H, Kq = 2, 8, 32, 64
m, c = tm.build(T, Kq, dev)
o = ClipSGD.from_cfg(m, c.optimizer, c.optimizer_config)
bs = [synthetic_batch(B, T, H, H, 0, s, device=dev) for s in range(3)]
say('built; mode', mode, 'det', det)
if mode == 'gather':
    import mscl_amd.parallel as P
    P.balanced_world = lambda n, world=None: 0          # force the all-gather formulation
gs = GraphedStep(m, o, bs[0], warmup=2)
say('captured; shuffle_mode', m.shuffle_mode)
for s in range(3):
    l = gs.step(bs[s])[0]
    torch.cuda.synchronize()
    say('step', s, float(l))
say('done')
dist.destroy_process_group()
