"""Per-term deviation of the step at the edge shapes of tests/test_model_gpu.py::test_step_other_shapes (odd widths, ragged halo
tiles, W = 112 planes) against the fp32 oracle: each loss term, q / k feature cosines, per-tensor gradient cosines, in the default
mode (two runs: the run-to-run spread of the float atomics) and in deterministic mode.
usage: python tools/other_shapes_diag.py  (GPU box; the oracle runs on the host cores)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    from mscl_amd import lib
    from mscl_amd.synthetic import synthetic_batch
    from oracle import fill as ofill, mscl as om
    from test_model_gpu import build
    dev = torch.device('cuda:0')
    cos = torch.nn.functional.cosine_similarity
    for B, T, H, W in [(3, 8, 90, 90), (2, 4, 70, 58), (1, 4, 224, 224)]:
        Kq = 16 * B
        batch = synthetic_batch(B, T, H, W, 0, 0)
        orc = om.MSCLWithAug(num_frames=T, K=Kq); ofill.fill_module(orc); orc.train()
        torch.manual_seed(100)
        ref = orc.train_step(batch)
        ref['loss'].backward()
        og = {n: p.grad for n, p in orc.named_parameters() if p.grad is not None}
        gn_o = float(torch.sqrt(sum((g.double() ** 2).sum() for g in og.values())))
        print(f'== shape {(B, T, H, W)}: oracle loss {ref["log_vars"]["loss"]:.6f} grad norm {gn_o:.4f}', flush=True)
        for mode in ('default', 'default', 'det'):
            lib.set_deterministic(mode == 'det')
            try:
                model, cfg = build(T, Kq, dev)
                out = model.train_step({k: [t.to(dev) for t in v] for k, v in batch.items()})
                model.zero_grad(); out['loss'].backward()
                model.sync_streams(); torch.cuda.synchronize()
            finally:
                lib.set_deterministic(False)
            devs = {k: (out['log_vars'][k] - v) / max(1.0, abs(v)) for k, v in ref['log_vars'].items() if 'loss' in k}
            worst = max(devs, key=lambda k: abs(devs[k]))
            feats = {}
            for nm, a, grp, w in (('q_rgb', model._dbg['q_rgb'], 'img', 'q'), ('k_rgb', model._dbg['k_rgb'], 'img', 'k'),
                                  ('q_fb', model._dbg['q_fb'], 'base', 'q'), ('q_fa', model._dbg['q_fa'], 'aug', 'q')):
                feats[nm] = cos(a.float().cpu(), orc._features[grp][w].detach(), dim=1).min().item()
            gcos = []
            for n, p in model.named_parameters():
                if p.requires_grad and n in og and float(og[n].norm()) >= 0.01 * gn_o:
                    gcos.append((float(cos(p.grad.detach().float().cpu().flatten(), og[n].flatten(), dim=0)), n))
            gn_h = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.requires_grad and p.grad is not None)))
            gcos.sort()
            print(f'  [{mode}] worst term {worst} {devs[worst]:+.2e}; ' + ' '.join(f'{k}={v:+.1e}' for k, v in devs.items()), flush=True)
            print(f'     features min cosine: ' + ' '.join(f'{k}={v:.5f}' for k, v in feats.items()) +
                  f'; grad norm {gn_h:.4f} ({gn_h / gn_o - 1:+.2%}); lowest per-tensor grad cosines: ' +
                  ', '.join(f'{c:.3f} {n}' for c, n in gcos[:4]), flush=True)


if __name__ == '__main__':
    main()
