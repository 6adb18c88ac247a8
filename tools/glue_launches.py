"""Which Python lines of one eager MSCL step call PyTorch's own device ops (copies, fills, element-wise glue) rather than the HIP
library: TorchDispatchMode records every aten op on a GPU tensor during one step with the innermost mscl_amd frame that
reached it.  usage: python tools/glue_launches.py [--streams]"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mscl_amd import ClipSGD, Config, build_model          # noqa: E402
from mscl_amd.fill import fill_module                      # noqa: E402
from mscl_amd.synthetic import synthetic_batch             # noqa: E402

LAUNCHING = ('copy_', 'clone', 'cat', 'stack', 'fill_', 'zero_', 'add', 'mul', 'sub', 'div', 'repeat', 'contiguous', 'zeros', 'full', 'ones',
             'empty_like', 'to', '_to_copy', 'index', 'sum', 'mean', 'lt', 'gt', 'where', 'neg', 'sqrt', 'clamp', 'expand')


class Rec(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split('.')[0]
        tens = [a for a in list(args) + [out] if isinstance(a, torch.Tensor)]
        if name in ('empty', 'empty_like', 'view', 'as_strided', 'detach', 'slice', 'select', 'reshape', '_unsafe_view', 'unsqueeze',
                    'squeeze', 't', 'transpose', 'permute', 'alias', 'expand', 'empty_strided', 'new_empty', 'record_stream', 'split',
                    'unbind', 'chunk', 'narrow', 'set_', 'is_pinned', '_local_scalar_dense', 'lift_fresh', 'new_empty_strided', 'view_as'):
            return out
        if any(t.is_cuda for t in tens):
            fr = [f for f in traceback.extract_stack() if '/mscl_amd/' in f.filename and not f.filename.endswith('lib.py')]
            site = f'{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}' if fr else 'autograd engine / other'
            self.sites[(name, site)] += 1
        return out


def main():
    dev = torch.device('cuda', 0)
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/recognition/moco/mscl_r18_cosm_lr2e-2.py'))
    cfg.model.sup_head.t = 8
    model = build_model(cfg.model); fill_module(model); model.materialize(dev).train()
    if '--streams' not in sys.argv:
        model.two_streams = False
    model.key_graphs = model.query_graphs = False
    opt = ClipSGD.from_cfg(model, cfg.optimizer, cfg.optimizer_config)
    batch = synthetic_batch(8, 16, 112, 112, 0, 0, device=dev)

    def step():
        out = model.train_step(batch, sync_logs=False)
        opt.zero_grad(); out['loss'].backward(); opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    rec = Rec()
    with rec:
        step()
    torch.cuda.synchronize()
    print(f'{sum(rec.sites.values())} aten device ops in one step')
    for (name, site), n in rec.sites.most_common(80):
        print(f'{n:4d}  {name:22s} {site}')


if __name__ == '__main__':
    main()
