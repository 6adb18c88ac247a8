"""One MSCL training step at a few clip shapes other than the benchmark's, HIP path against the CPU oracle (losses within the
step test's tolerance, queue bookkeeping equal).  Covers the paths a shape can switch: odd widths (paired stem), planes too wide
for the window-resident layer-1 kernels (W > 61 after the stem), small maps, B not a power of two.
usage: python tools/shape_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [(2, 8, 64, 64), (3, 8, 90, 90), (1, 4, 224, 224), (2, 8, 112, 144), (2, 4, 70, 58)]      # B, T, H, W


def main():
    from mscl_amd import ClipSGD, Config, build_model
    from mscl_amd.fill import fill_module
    from mscl_amd.synthetic import synthetic_batch
    from oracle import fill as ofill, mscl as om
    dev = torch.device('cuda:0')
    bad = 0
    for B, T, H, W in SHAPES:
        Kq = 16 * B
        cfg = Config.fromfile(os.path.join(ROOT, 'configs/recognition/moco/mscl_r18_cosm_lr2e-2.py'))
        cfg.model.sup_head.t = T // 2
        cfg.model.recognizer.K = Kq
        cfg.model.recognizer_flow.K = Kq
        model = build_model(cfg.model)
        fill_module(model)
        model.materialize(dev).train()
        opt = ClipSGD.from_cfg(model, cfg.optimizer, cfg.optimizer_config)
        batch = synthetic_batch(B, T, H, W, 0, 0)
        out = model.train_step({k: [t.to(dev) for t in v] for k, v in batch.items()})
        opt.zero_grad(); out['loss'].backward(); opt.step()
        torch.cuda.synchronize()
        orc = om.MSCLWithAug(num_frames=T, K=Kq)
        ofill.fill_module(orc); orc.train()
        ref = orc.train_step(batch)
        worst = max(abs(out['log_vars'][k] - v) / (0.02 * max(1.0, abs(v)) + 0.03) for k, v in ref['log_vars'].items() if 'loss' in k)
        ok = worst <= 1.0 and int(model.recognizer.queue_ptr) == int(orc.recognizer.queue_ptr) and \
            torch.equal(model.recognizer_flow.count.cpu(), orc.recognizer_flow.count)
        bad += not ok
        print(f'B={B} T={T} {H}x{W}: loss {out["log_vars"]["loss"]:.4f} oracle {ref["log_vars"]["loss"]:.4f} '
              f'worst loss error / tolerance {worst:.3f} grad_norm {float(opt.grad_norm()):.2f} {"ok" if ok else "MISMATCH"}', flush=True)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
