"""A/B of a Python-level step setting inside ONE process: the whole MSCL step is captured twice, once per arm, and the two HIP graphs
are replayed alternately (rounds x 20 replays each); prints clip-pairs/s per arm and round.  Arms are `module.attr=value` settings
applied before each capture, e.g.
    python tools/ab_step.py "nn.WGRAD_LANE['on']=False" "nn.WGRAD_LANE['on']=True" [--rounds 4]
(`nn`, `K`, `lib`, `model` are in scope; several assignments per arm separate with ';')."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mscl_amd import ClipSGD, Config, build_model, kernels as K, lib, nn       # noqa: E402,F401
from mscl_amd.fill import fill_module                                            # noqa: E402
from mscl_amd.graph import GraphedStep                                           # noqa: E402
from mscl_amd.synthetic import synthetic_batch                                   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('arms', nargs='+')
    ap.add_argument('--rounds', type=int, default=4)
    ap.add_argument('--steps', type=int, default=20)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/recognition/moco/mscl_r18_cosm_lr2e-2.py'))
    cfg.model.sup_head.t = 8
    model = build_model(cfg.model)
    fill_module(model)
    model.materialize(dev).train()
    opt = ClipSGD.from_cfg(model, cfg.optimizer, cfg.optimizer_config)
    batches = [synthetic_batch(8, 16, 112, 112, 0, s, device=dev) for s in range(4)]
    graphs = []
    for arm in a.arms:
        exec(arm, globals(), dict(model=model))
        graphs.append(GraphedStep(model, opt, batches[0], warmup=2))
    for g in graphs:
        for i in range(3):
            g.step(batches[i % 4])
    torch.cuda.synchronize()
    res = [[] for _ in graphs]
    for r in range(a.rounds):
        for gi, g in enumerate(graphs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(a.steps):
                loss = g.step(batches[i % 4])[0]
            torch.cuda.synchronize()
            res[gi].append(8 * a.steps / (time.perf_counter() - t0))
    for arm, rr in zip(a.arms, res):
        print(f'{arm:60s} ' + ' '.join(f'{v:7.1f}' for v in rr) + f'   median {sorted(rr)[len(rr) // 2]:7.1f}   loss {float(loss):.3f}', flush=True)


if __name__ == '__main__':
    main()
