"""BASELINE.json configs[4]: the full MSCL step of configs/recognition/moco/mscl_r50_cosm_lr3e-2.py (ResNet3dSlowOnly-50 +
r2d_50, TPN + SEPC neck, MoCo queues K = 65536, cross-modal InfoNCE, LMCL with its flow transform, backward, clip + SGD) on
synthetic clips of --frames x --side^2 (default 32 x 224^2, the deep / large-activation case; the shipped config's own clip
is 8 x 224^2), --batch clip pairs on one GPU.  Prints one JSON line: clip-pairs/s, ms per step, peak memory.
usage: python tools/bench_step_r50.py [--frames 32] [--side 224] [--batch 8] [--steps 10] [--warmup 3] [--no-graph]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=32)
    ap.add_argument('--side', type=int, default=224)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-graph', action='store_true')
    a = ap.parse_args()
    from mscl_amd import ClipSGD, Config, build_model
    from mscl_amd.fill import fill_module
    from mscl_amd.synthetic import synthetic_batch
    dev = torch.device('cuda:0')
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/recognition/moco/mscl_r50_cosm_lr3e-2.py'))
    cfg.model.sup_head.t = a.frames // 2            # the config derives it from num_frames
    model = build_model(cfg.model)
    fill_module(model)
    model.materialize(dev).train()
    opt = ClipSGD.from_cfg(model, cfg.optimizer, cfg.optimizer_config)
    batches = [synthetic_batch(a.batch, a.frames, a.side, a.side, 0, s, device=dev) for s in range(2)]

    def eager(i):
        out = model.train_step(batches[i % 2], sync_logs=False)
        opt.zero_grad(); out['loss'].backward(); opt.step()
        return out['loss']
    graphed = None
    if not a.no_graph:
        try:
            from mscl_amd.graph import GraphedStep
            graphed = GraphedStep(model, opt, batches[0], warmup=2)
        except Exception as e:      # noqa: BLE001
            print(f'[bench_step_r50] graph capture failed ({type(e).__name__}: {e}); eager launches', file=sys.stderr)
    step = (lambda i: graphed.step(batches[i % 2])[0]) if graphed is not None else eager
    if graphed is None:
        for i in range(3):
            eager(i)
    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(a.warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lv = float(loss.detach())
    if lv != lv:
        raise SystemExit('loss is NaN')
    print(json.dumps({'metric': f'clip-pairs/sec (mscl_r50 full step, {a.frames}x{a.side}^2, bs{a.batch}, 1 GPU)',
                      'value': a.batch * a.steps / dt, 'unit': 'clip-pairs/s', 'ms_per_step': 1e3 * dt / a.steps, 'steps': a.steps,
                      'warmup': a.warmup, 'dtype': 'bf16', 'data': 'synthetic', 'final_loss': lv,
                      'peak_mem_gb': torch.cuda.max_memory_allocated() / 2 ** 30,
                      'launch': 'one captured HIP graph per step' if graphed is not None else 'eager launches',
                      'config': {'workload': 'full MSCLWithAug step, mscl_r50_cosm_lr3e-2.py (ResNet3dSlowOnly-50 + r2d_50)',
                                 'clip': f'{a.frames}x{a.side}x{a.side}', 'batch_per_gpu': a.batch}}))


if __name__ == '__main__':
    main()
