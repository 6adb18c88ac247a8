"""HIP path (bf16 convolutions) vs the fp32 oracle over a short training run from the same weights on the same batches: does the
bf16 storage ahead of BatchNorm-backward (per-tensor gradient cosine ~0.92 on trunk kernels) change what the
optimizer does?  60 optimizer steps (clip 40, SGD 0.9 / 1e-4, lr 0.02) on four rotating synthetic batches, B = 4, T = 8, 64^2,
K = 256 -- the setting of tests/test_model_gpu.py::test_training_learns, where the frame-level LMCL term has a learnable answer.
Prints a markdown table of the total loss, loss_pos and loss_cls of both runs every 5 steps.
usage: python tools/train_curve.py [--steps 60] [--batch 4] [--every 5] [--deterministic] > profiles/r03_training_curve.md
(--steps 500 --batch 2 --every 25 --deterministic: the long curve of round 3, the HIP side with fixed-order sums)"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--every', type=int, default=5)
    ap.add_argument('--deterministic', action='store_true')
    ap.add_argument('--frames', type=int, default=8)
    ap.add_argument('--side', type=int, default=64)
    ap.add_argument('--queue', type=int, default=256)
    ap.add_argument('--oracle-only', default=None, metavar='JSON',
                    help='run ONLY the fp32 CPU oracle (no GPU needed) and write its per-step log values to this file -- the long '
                         'full-size curve is hours of CPU work, so it is made once, in the dev container, and committed as a fixture')
    ap.add_argument('--oracle-json', default=None, metavar='JSON', help='read the oracle side from such a file instead of running it')
    a = ap.parse_args()
    if a.oracle_only:
        return oracle_only(a)
    import test_model_gpu as tm
    from mscl_amd import ClipSGD
    from mscl_amd.synthetic import synthetic_batch
    from oracle import fill as ofill, mscl as om
    dev = torch.device('cuda:0')
    B, T, H, Kq = a.batch, a.frames, a.side, a.queue
    if a.deterministic:
        from mscl_amd import lib
        lib.set_deterministic(True)
    model, cfg = tm.build(T, Kq, dev)
    opt = ClipSGD.from_cfg(model, cfg.optimizer, cfg.optimizer_config)
    stored = None
    if a.oracle_json:
        import json
        stored = json.load(open(a.oracle_json))
        assert (stored['batch'], stored['frames'], stored['side'], stored['queue']) == (B, T, H, Kq), 'the fixture was made for another size'
    else:
        orc = om.MSCLWithAug(num_frames=T, K=Kq); ofill.fill_module(orc); orc.train()
        oopt = om.SGDClip(orc.parameters(), lr=cfg.optimizer.lr)
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    batches = [synthetic_batch(B, T, H, H, 0, s) for s in range(4)]
    dbatches = [{k: [t.to(dev) for t in v] for k, v in b.items()} for b in batches]
    keys = ('loss', 'loss_pos', 'loss_cls', 'loss_cls_flow', 'loss_cls_mx')
    print('# HIP path (bf16 convolutions) vs fp32 oracle: %d optimizer steps from the same weights on the same batches\n' % a.steps)
    print('`python tools/train_curve.py --steps %d --batch %d%s` (B = %d, T = %d, %d^2, K = %d, four rotating synthetic batches, lr 0.02, clip 40%s).  After step 0 the two'
          % (a.steps, B, ' --deterministic' if a.deterministic else '', B, T, H, Kq,
             '; oracle side from the committed fixture ' + os.path.basename(a.oracle_json) if stored else ''))
    print('runs are different trajectories of a chaotic system (small-batch BatchNorm), so values are compared as curves, not digit by digit.\n')
    print('| step | ' + ' | '.join(f'{k} hip / oracle' for k in keys) + ' | grad norm hip / oracle |')
    print('|---|' + '---|' * (len(keys) + 1))
    for it in range(a.steps):
        out = model.train_step(dbatches[it % 4])
        opt.zero_grad(); out['loss'].backward(); opt.step()
        gh = float(opt.grad_norm())
        if stored is not None:
            if it >= len(stored['steps']):
                break
            ov, go = stored['steps'][it]['log_vars'], stored['steps'][it]['grad_norm']
        else:
            torch.manual_seed(100 + it)
            oo = orc.train_step(batches[it % 4]); oopt.zero_grad(); oo['loss'].backward()
            go = oopt.step()
            ov = oo['log_vars']
        if it % a.every == 0 or it == a.steps - 1:
            lv = out['log_vars']
            print(f'| {it} | ' + ' | '.join(f'{lv[k]:.4f} / {ov[k]:.4f}' for k in keys) + f' | {gh:.1f} / {go:.1f} |', flush=True)


def oracle_only(a):
    """the CPU side alone, flushed to the JSON file every 10 steps (a long run can be read, or resumed from, at any point)"""
    import json
    import time
    from mscl_amd.synthetic import synthetic_batch
    from oracle import fill as ofill, mscl as om
    B, T, H, Kq = a.batch, a.frames, a.side, a.queue
    orc = om.MSCLWithAug(num_frames=T, K=Kq); ofill.fill_module(orc); orc.train()
    oopt = om.SGDClip(orc.parameters(), lr=0.02)
    batches = [synthetic_batch(B, T, H, H, 0, s) for s in range(4)]
    rec = dict(batch=B, frames=T, side=H, queue=Kq, lr=0.02, steps=[])
    t0 = time.time()
    for it in range(a.steps):
        torch.manual_seed(100 + it)
        oo = orc.train_step(batches[it % 4]); oopt.zero_grad(); oo['loss'].backward()
        go = oopt.step()
        rec['steps'].append(dict(log_vars={k: float(v) for k, v in oo['log_vars'].items()}, grad_norm=float(go)))
        if it % 10 == 9 or it == a.steps - 1:
            json.dump(rec, open(a.oracle_only, 'w'))
            print(f'step {it} loss {rec["steps"][-1]["log_vars"]["loss"]:.4f}  {time.time() - t0:.0f} s', flush=True)


if __name__ == '__main__':
    main()
