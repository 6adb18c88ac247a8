"""The oracle (CPU restatement) against the golden vectors captured from the reference's own Python
(tools/oracle/make_golden.py) and against the only known-answer vectors the reference's tests hold for
this path (top_k_accuracy, reference tests/test_metrics/test_accuracy.py:118-163)."""
import json
import os

import numpy as np
import pytest
import torch

from mscl_amd.synthetic import synthetic_batch
from oracle import fill, mscl as om

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _run(tag, max_steps, arch='r18', lr=0.02):
    g = np.load(os.path.join(GOLD, f'{tag}.npz'))
    meta = json.loads(str(g['meta']))
    B, T, H, K = meta['B'], meta['T'], meta['H'], meta['K']
    orc = om.MSCLWithAug(num_frames=T, K=K, arch=arch)
    fill.fill_module(orc)
    orc.train()
    opt = om.SGDClip(orc.parameters(), lr=lr)
    keys = [str(k) for k in g['log_keys']]
    names = [str(n) for n in g['param_names']]
    for s in range(min(max_steps, meta['n_steps'])):
        batch = synthetic_batch(B, T, H, H, 0, s)
        torch.manual_seed(100 + s)
        out = orc.train_step(batch)
        assert list(out['log_vars'].keys()) == keys
        tol = 1e-6 if s == 0 else 2e-3
        for k, ref in zip(keys, g[f's{s}_log_vals']):
            assert abs(out['log_vars'][k] - ref) <= tol * max(1.0, abs(ref)), (tag, s, k, out['log_vars'][k], ref)
        opt.zero_grad()
        out['loss'].backward()
        grads = dict(orc.named_parameters())
        for n, ref in zip(names, g[f's{s}_grad_l2']):
            gr = grads[n].grad
            if ref < 0:
                assert gr is None, n
            else:
                assert abs(float(gr.double().norm()) - ref) <= (1e-5 if s == 0 else 2e-2) * max(ref, 1e-6), (n, s)
        gn = opt.step()
        assert abs(gn - float(g[f's{s}_grad_norm'])) <= (1e-5 if s == 0 else 1e-2) * gn
        for nm, rec in (('rgb', orc.recognizer), ('flow', orc.recognizer_flow)):
            assert int(rec.queue_ptr) == int(g[f's{s}_{nm}_ptr'])
            assert rec.iters == int(g[f's{s}_{nm}_iters']) and rec.batch_size == int(g[f's{s}_{nm}_bs'])
            assert rec.m == float(g[f's{s}_{nm}_m'])
            if f's{s}_{nm}_count' in g:
                assert np.array_equal(rec.count.numpy(), g[f's{s}_{nm}_count'])
    return g


def test_oracle_matches_reference_small_queue_bookkeeping():
    _run('book_b2_t8_h32_k64', 40)


def test_oracle_matches_reference_step_t8():
    _run('step_b2_t8_h112', 1)


def test_oracle_matches_reference_step_t16():
    """the benchmark's clip length (BASELINE.json: 16 x 112^2)"""
    _run('step_b2_t16_h112', 1)


def test_oracle_r50_matches_reference_step():
    """BASELINE.json configs[4] (mscl_r50_cosm_lr3e-2.py: ResNet3dSlowOnly-50 + r2d_50): the restatement against the reference's
    own step on the same weights and clips (tools/oracle/make_golden_r50.py)"""
    _run('r50_step_b2_t8_h64', 1, arch='r50', lr=0.0075)


def test_oracle_r50_state_dict_is_the_reference_manifest():
    man = json.load(open(os.path.join(GOLD, 'state_dict_manifest_r50.json')))
    sd = om.MSCLWithAug(num_frames=8, K=65536, arch='r50').state_dict()
    skip = ('moco_head', 'moco_mx_head')        # parameter-free heads the oracle folds into functions
    assert [(n, list(t.shape), str(t.dtype)) for n, t in sd.items()] == [tuple(m) for m in man if not m[0].startswith(skip)]


def test_log_keys_are_the_23_reference_keys():
    g = np.load(os.path.join(GOLD, 'step_b2_t8_h112.npz'))
    from mscl_amd.recognizers import LOG_KEYS
    assert tuple(str(k) for k in g['log_keys']) == LOG_KEYS and len(LOG_KEYS) == 23


def test_top_k_known_answers():
    # data of reference tests/test_metrics/test_accuracy.py:118-163
    scores = [np.array([-0.2203, -0.7538, 1.8789, 0.4451, -0.2526]), np.array([-0.0413, 0.6366, 1.1155, 0.3484, 0.0395]),
              np.array([0.0365, 0.5158, 1.1067, -0.9276, -0.2124]), np.array([0.6232, 0.9912, -0.8562, 0.0148, 1.6413])]
    cases = [((1,), [0, 2, 0, 3], [0.25]), ((2,), [0, 2, 0, 3], [0.25]), ((1,), [0, 1, 2, 3], [0.25]),
             ((1, 2), [0, 1, 2, 3], [0.25, 0.5]), ((1, 2, 3), [0, 1, 2, 3], [0.25, 0.5, 0.75]),
             ((1, 2, 3, 4), [0, 1, 2, 3], [0.25, 0.5, 0.75, 1.0])]
    for topk, labels, want in cases:
        assert om.top_k_accuracy(scores, labels, topk) == want
        # the product's rank-count form agrees (no ties in this table)
        from mscl_amd.heads import logits_topk
        got = logits_topk(torch.tensor(np.stack(scores)), torch.tensor(labels), topk)
        assert [float(v) for v in got] == want


def test_momentum_schedule_closed_form():
    from mscl_amd.recognizers import momentum_at
    for it in (0, 1, 10 ** 6, 219136 * 400, 10 ** 9):
        assert momentum_at(it, 219136 * 400, 0.994) == om.momentum_at(it, 219136 * 400, 0.994)
    assert om.momentum_at(0, 100, 0.994) == 0.994 and abs(om.momentum_at(100, 100, 0.994) - 1.0) < 1e-15


def test_flow_visualizer_oracle_vs_reference_golden():
    """G7: oracle/flowvis.py reproduces the levels the reference's FlowVisualizer produced (tools/oracle/make_golden_flowvis.py)."""
    import os
    import numpy as np
    import torch
    from oracle import flowvis
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'flowvis_g7.npz'))
    assert np.array_equal(flowvis.make_colorwheel(), g['colorwheel'])
    vis = flowvis.FlowVisualizer()
    for a, b in (('uv', 'levels'), ('uv2', 'levels2')):
        out = vis(torch.from_numpy(g[a]))
        assert np.array_equal(torch.round(out * 255).to(torch.uint8).numpy(), g[b])


def test_fra_oracle_vs_reference_golden():
    """G8: oracle/flowaug.py reproduces the reference's NormFlowWithStidedAug outputs (tools/oracle/make_golden_fra.py)."""
    import os
    import numpy as np
    from oracle import flowaug
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'fra_g8.npz'))
    for b in range(g['uv'].shape[0]):
        out = flowaug.fra([g['uv'][b, t] for t in range(g['uv'].shape[1])], int(g['cid'][b]))
        assert np.array_equal(np.stack(out).astype(np.float64), g['normed'][b])


def test_coloraug_restatement_properties():
    """oracle/coloraug.py has no reference vector to pin it (kornia is absent and unpinned: 'parity unpinned'); what can be
    checked are the identities the colour model guarantees: HSV round trip, unit factors are the identity, hue shift by a
    full turn is the identity, grayscale is channel-constant with the 601 weights, blur preserves constants and mass."""
    import math
    from oracle import coloraug as c
    g = torch.Generator().manual_seed(2)
    x = torch.rand((2, 3, 2, 9, 10), generator=g)
    h, s, v = c.rgb_to_hsv(x[0].permute(1, 0, 2, 3))
    assert (c.hsv_to_rgb(h, s, v) - x[0].permute(1, 0, 2, 3)).abs().max() < 5e-6
    assert float(h.min()) >= 0 and float(h.max()) < 2 * math.pi + 1e-6
    P = torch.zeros(2, 16)
    P[:, 0] = 1; P[:, 1:5] = torch.tensor([0., 1., 2., 3.]); P[:, 5:8] = 1.0
    assert (c.color_aug(x, P) - x).abs().max() < 5e-6
    P[:, 8] = 2 * math.pi
    assert (c.color_aug(x, P) - x).abs().max() < 5e-6
    P[:, 0] = 0; P[:, 9] = 1
    y = c.color_aug(x, P)
    assert torch.equal(y[:, 0], y[:, 1]) and torch.equal(y[:, 1], y[:, 2])
    assert (y[:, 0] - (0.299 * x[:, 0] + 0.587 * x[:, 1] + 0.114 * x[:, 2])).abs().max() < 1e-6
    ones = torch.ones(1, 3, 12, 12)
    assert (c.gaussian_blur(ones, 11, 1.7) - 1).abs().max() < 1e-6
    assert abs(float(c.gaussian_taps(11, 0.1)[5]) - 1.0) < 1e-6           # sigma 0.1: a delta


def test_finetune_oracle_vs_reference_golden():
    """oracle/recognizer3d.py reproduces what the reference's Recognizer3D + I3DHead gave on the same closed-form weights
    and seeded clips (G9, tools/oracle/make_golden_finetune.py): losses, gradient norms, evaluation-mode probabilities."""
    import json
    from oracle import fill as ofill, recognizer3d as orec
    gold = json.load(open(os.path.join(GOLD, 'finetune_g9.json')))
    cfg = gold['config']
    g = torch.Generator().manual_seed(cfg['seed'])
    imgs = torch.randn((cfg['B'], 1, 3, cfg['T'], cfg['H'], cfg['H']), generator=g)
    test_imgs = torch.randn((cfg['B'], cfg['clips'], 3, cfg['T'], cfg['H'], cfg['H']), generator=g)
    m = orec.Recognizer3D(cfg['num_classes'], dropout_ratio=0.0)
    ofill.fill_module(m)
    m.train()
    out = m.train_step(dict(imgs=imgs, label=torch.tensor(cfg['labels']).view(-1, 1)))
    out['loss'].backward()
    for k, v in gold['log_vars'].items():
        assert abs(out['log_vars'][k] - v) <= 1e-5 * max(1.0, abs(v)), k
    gn = torch.sqrt(sum((p.grad ** 2).sum() for p in m.parameters() if p.grad is not None)).item()
    assert abs(gn - gold['grad_norm']) <= 1e-4 * gold['grad_norm']
    m.eval()
    assert (m.forward_test(test_imgs) - torch.tensor(gold['probs'])).abs().max() < 1e-5


def test_retrieval_oracle_and_metric_vs_reference_golden():
    """G10: tools/test_retrival.py:283-303 executed verbatim on seeded features (tools/oracle/make_golden_retrieval.py); both the
    numpy restatement and the product's metric function reproduce its accuracies."""
    from mscl_amd import retrieval
    from oracle import retrieval as oret
    gold = json.load(open(os.path.join(GOLD, 'retrieval_g10.json')))
    for case in gold['cases'].values():
        tf, tl, sf, sl = oret.clustered_features(case['seed'], noise=case['noise'])
        ora = oret.knn_accuracy(tf.numpy(), tl.numpy(), sf.numpy(), sl.numpy())
        got = retrieval.knn_accuracy(tf, tl, sf, sl)
        for k, v in case['acc'].items():
            assert abs(ora[int(k)] - v) < 1e-6 and abs(got[int(k)] - v) < 1e-6, (case['seed'], k, ora[int(k)], got[int(k)], v)
    with pytest.raises(AssertionError):
        retrieval.knn_accuracy(tf, tl[:-1], sf, sl)
