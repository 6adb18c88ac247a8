"""Generate tests/golden/*.npz|json by running the REFERENCE's own Python (via ref_harness) on CPU.

Run only in the development container:  python tools/oracle/make_golden.py
Every fixture is data (inputs are re-generated from seeds by mscl_amd.synthetic; expected outputs
are stored).  While generating, the oracle/ restatement is asserted equal to the reference.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness as rh                              # noqa: E402
from mscl_amd.synthetic import synthetic_batch        # noqa: E402
from oracle import fill, mscl as om                   # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


CFG_OF = {'r18': 'mscl_r18_cosm_lr2e-2.py', 'r50': 'mscl_r50_cosm_lr3e-2.py'}


def sgd_ref(model, arch='r18'):
    """The reference's optimizer stack = mmcv OptimizerHook(grad_clip max_norm 40, L2) around
    torch.optim.SGD built from the reference config's `optimizer` dict."""
    cfg = rh.load_ref_cfg(CFG_OF[arch])
    o = dict(cfg['optimizer']); o.pop('type')
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, **o)
    clip = cfg['optimizer_config']['grad_clip']
    def step():
        gn = torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], **clip)
        opt.step()
        return float(gn)
    return opt, step


def stats(t):
    t = t.detach().double().flatten()
    idx = torch.linspace(0, t.numel() - 1, 16).long()
    return np.array([t.mean().item(), t.norm().item()] + t[idx].tolist())


def run_steps(B, T, H, n_steps, K, tag, with_feats=True, arch='r18'):
    torch.manual_seed(0)
    ref, _ = rh.build_ref_model(num_frames=T, K=K, cfg_name=CFG_OF[arch])
    orc = om.MSCLWithAug(num_frames=T, K=K, arch=arch)
    fill.fill_module(ref); fill.fill_module(orc)
    ref.train(); orc.train()
    opt_r, step_r = sgd_ref(ref, arch)
    opt_o = om.SGDClip(orc.parameters(), lr=opt_r.defaults['lr'])
    out = {}
    names = [n for n, p in ref.named_parameters() if p.requires_grad]
    for s in range(n_steps):
        batch = synthetic_batch(B, T, H, H, rank=0, step=s)
        torch.manual_seed(100 + s); o_r = ref.train_step(batch, None)
        torch.manual_seed(100 + s); o_o = orc.train_step(batch, None)
        opt_r.zero_grad(); o_r['loss'].backward()
        opt_o.zero_grad(); o_o['loss'].backward()
        for k in o_r['log_vars']:
            a, b = o_r['log_vars'][k], o_o['log_vars'][k]
            assert abs(a - b) <= (1e-6 if s == 0 else 2e-4) * max(1, abs(a)), (tag, s, k, a, b)
        out[f'log_keys'] = np.array(list(o_r['log_vars'].keys()))
        out[f's{s}_log_vals'] = np.array(list(o_r['log_vars'].values()), dtype=np.float64)
        gr = {n: p.grad for n, p in ref.named_parameters() if p.requires_grad}
        go = {n: p.grad for n, p in orc.named_parameters() if p.requires_grad}
        gl2 = []
        for n in names:
            if gr[n] is None:
                assert go[n] is None; gl2.append(-1.0); continue
            d = (gr[n] - go[n]).abs().max().item()
            assert d <= (1e-6 if s == 0 else 5e-3) * (gr[n].abs().max().item() + 1e-12), (tag, s, n, d)
            gl2.append(gr[n].double().norm().item())
        out[f's{s}_grad_l2'] = np.array(gl2)
        if with_feats and s == 0:
            f = orc._features
            for nm, fd in f.items():
                out[f'feat_{nm}_q'] = fd['q'].detach().numpy()
                out[f'feat_{nm}_k'] = fd['k'].detach().numpy()
                for li, l in enumerate(fd['q_mlvl']):
                    out[f'feat_{nm}_qmlvl{li}'] = stats(l)
        gn_r = step_r(); gn_o = opt_o.step()
        assert abs(gn_r - gn_o) <= (1e-6 if s == 0 else 1e-3) * gn_r, (gn_r, gn_o)
        out[f's{s}_grad_norm'] = np.array(gn_r)
        out[f's{s}_param_l2'] = np.array([p.detach().double().norm().item() for n, p in ref.named_parameters()
                                          if p.requires_grad])
        for nm, rec_r, rec_o in (('rgb', ref.recognizer, orc.recognizer), ('flow', ref.recognizer_flow, orc.recognizer_flow)):
            assert int(rec_r.queue_ptr) == int(rec_o.queue_ptr) and torch.equal(rec_r.count, rec_o.count)
            assert rec_r.iters == rec_o.iters and rec_r.batch_size == rec_o.batch_size
            out[f's{s}_{nm}_ptr'] = np.array(int(rec_r.queue_ptr))
            out[f's{s}_{nm}_iters'] = np.array(rec_r.iters)
            out[f's{s}_{nm}_bs'] = np.array(rec_r.batch_size)
            out[f's{s}_{nm}_m'] = np.array(rec_r.m)
            if K <= 4096:
                out[f's{s}_{nm}_count'] = rec_r.count.numpy().copy()
            else:
                c = rec_r.count.numpy()
                out[f's{s}_{nm}_count_hist'] = np.stack(np.unique(c, return_counts=True))
    out['param_names'] = np.array(names)
    out['meta'] = np.array(json.dumps(dict(B=B, T=T, H=H, K=K, n_steps=n_steps)))
    np.savez_compressed(os.path.join(OUT, f'{tag}.npz'), **out)
    print('wrote', tag, {k: v.shape for k, v in list(out.items())[:3]})


def main():
    os.makedirs(OUT, exist_ok=True)
    rh.install()
    # G6: the reference's model/optimizer dicts, to pin the authored config
    cfg = rh.load_ref_cfg()
    keep = {k: cfg[k] for k in ('model', 'optimizer', 'optimizer_config', 'lr_config', 'total_epochs',
                                'dataset_size', 'num_frames', 'find_unused_parameters')}
    with open(os.path.join(OUT, 'ref_config.json'), 'w') as f:
        json.dump(keep, f, indent=1, sort_keys=True)
    # state-dict manifest (names, shapes, dtypes): checkpoint compatibility surface
    ref, _ = rh.build_ref_model(num_frames=8)
    man = [[n, list(t.shape), str(t.dtype)] for n, t in ref.state_dict().items()]
    with open(os.path.join(OUT, 'state_dict_manifest.json'), 'w') as f:
        json.dump(man, f)
    del ref
    # G3: step-level, real resolution
    run_steps(B=2, T=8, H=112, n_steps=3, K=65536, tag='step_b2_t8_h112')
    run_steps(B=2, T=16, H=112, n_steps=1, K=65536, tag='step_b2_t16_h112')
    # G2/G4: reduced spatial size, small K with wrap-around, 40 steps of integer bookkeeping
    run_steps(B=2, T=8, H=32, n_steps=40, K=64, tag='book_b2_t8_h32_k64', with_feats=True)


if __name__ == '__main__':
    main()
