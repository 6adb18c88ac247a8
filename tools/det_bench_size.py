"""Deterministic mode at the BENCHMARK size (B = 8, T = 16, 112^2, K = 65536), the whole step replayed from one captured three-stream HIP
graph: RUNS runs of STEPS optimizer steps from the same weights on the same four rotating batches; every run's loss sequence, final
parameter / key / momentum arenas and both queues are compared with the first run's BIT FOR BIT.  (The round-5 review's task 1c: the
bitwise test of the suite runs at B = 2, 32^2; this is where the product runs.)
usage: python tools/det_bench_size.py [STEPS [RUNS]]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mscl_amd import ClipSGD, Config, build_model, lib          # noqa: E402
from mscl_amd.fill import fill_module                           # noqa: E402
from mscl_amd.graph import GraphedStep                          # noqa: E402
from mscl_amd.synthetic import synthetic_batch                  # noqa: E402


def run(steps, dev, batches):
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/recognition/moco/mscl_r18_cosm_lr2e-2.py'))
    cfg.model.sup_head.t = 8
    torch.manual_seed(0)
    model = build_model(cfg.model)
    fill_module(model)
    model.materialize(dev).train()
    opt = ClipSGD.from_cfg(model, cfg.optimizer, cfg.optimizer_config)
    g = GraphedStep(model, opt, batches[0], warmup=0)
    losses = []
    for i in range(steps):
        losses.append(g.step(batches[i % 4])[0].clone())
    torch.cuda.synchronize()
    ar = model.arena
    state = dict(loss=torch.stack(losses), Q=ar.Q.clone(), KX=ar.KX.clone(), MOM=ar.MOM.clone(),
                 queue_rgb=model.recognizer.queue.clone(), queue_flow=model.recognizer_flow.queue.clone())
    del g, opt, model
    return state


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device('cuda', 0)
    lib.set_deterministic(True)
    batches = [synthetic_batch(8, 16, 112, 112, 0, s, device=dev) for s in range(4)]
    ref = run(steps, dev, batches)
    print(f'run 0: {steps} steps, loss[0] {float(ref["loss"][0]):.6f} loss[-1] {float(ref["loss"][-1]):.6f}', flush=True)
    bad = 0
    for r in range(1, runs):
        cur = run(steps, dev, batches)
        diffs = {k: int((cur[k] != ref[k]).sum()) for k in ref}
        first = int((cur['loss'] != ref['loss']).nonzero()[0]) if diffs['loss'] else -1
        print(f'run {r}: differing elements {diffs}' + (f'; first differing step {first}' if first >= 0 else ''), flush=True)
        bad += any(diffs.values())
    print(f'{runs - 1} runs compared with the first: {bad} differ')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
