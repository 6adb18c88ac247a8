"""the supervised consumer of the pre-trained RGB encoder (Recognizer3D + I3DHead of
configs/recognition/ssl_test/test_ssv2_r18.py) on the HIP kernels, against oracle/recognizer3d.py -- which is pinned
bit-exactly to the reference's own classes by tools/oracle/make_golden_finetune.py (tests/golden/finetune_g9.json)."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda', 0)


def build(num_classes, dropout, dev, test_cfg=None):
    from mscl_amd import build_model
    from mscl_amd.fill import fill_module
    m = build_model(dict(type='Recognizer3D', backbone=dict(type='torchvision.r3d_18'),
                         cls_head=dict(type='I3DHead', num_classes=num_classes, in_channels=512, spatial_type='none',
                                       dropout_ratio=dropout),
                         test_cfg=test_cfg or dict(average_clips='prob')))
    fill_module(m)
    return m.materialize(dev)


def golden_inputs(cfg):
    g = torch.Generator().manual_seed(cfg['seed'])
    imgs = torch.randn((cfg['B'], 1, 3, cfg['T'], cfg['H'], cfg['H']), generator=g)
    test_imgs = torch.randn((cfg['B'], cfg['clips'], 3, cfg['T'], cfg['H'], cfg['H']), generator=g)
    return imgs, test_imgs, torch.tensor(cfg['labels']).view(-1, 1)


def test_finetune_step_and_eval_match_reference_golden(dev):
    """training forward / backward and evaluation-mode scores vs the values the REFERENCE classes produced (G9) and vs the
    oracle's gradients.  Tolerances: bf16 activations -> 2 % on the loss, 0.02 on probabilities, cosine >= 0.99 per tensor."""
    from oracle import fill as ofill, recognizer3d as orec
    gold = json.load(open(os.path.join(GOLD, 'finetune_g9.json')))
    cfg = gold['config']
    imgs, test_imgs, label = golden_inputs(cfg)
    model = build(cfg['num_classes'], 0.0, dev).train()
    ora = orec.Recognizer3D(cfg['num_classes'], dropout_ratio=0.0)
    ofill.fill_module(ora)
    ora.train()
    o = ora.train_step(dict(imgs=imgs, label=label))
    o['loss'].backward()
    out = model.train_step(dict(imgs=imgs.to(dev), label=label.to(dev)))
    model.arena.G.zero_()
    out['loss'].backward()
    torch.cuda.synchronize()
    for k in ('loss_cls', 'loss'):
        assert abs(out['log_vars'][k] - gold['log_vars'][k]) <= 0.02 * abs(gold['log_vars'][k]), (k, out['log_vars'][k], gold['log_vars'][k])
    assert out['log_vars']['top5_acc'] == gold['log_vars']['top5_acc'] and out['num_samples'] == cfg['B']
    cos = torch.nn.functional.cosine_similarity
    named = dict(model.named_parameters())
    worst, flat_h, flat_o = (1.0, ''), [], []
    for name, po in ora.named_parameters():
        g_hip = named[name].grad.detach().float().cpu().reshape(-1)
        g_ref = po.grad.reshape(-1)
        flat_h.append(g_hip); flat_o.append(g_ref)
        if g_ref.norm() > 1e-8:
            worst = min(worst, (cos(g_hip, g_ref, dim=0).item(), name))
    # Batch 2 at 32x32 leaves 8 positions per channel in layer 4, so bf16 activations alone move the gradient by a few
    # degrees.  Yardstick: the SAME oracle under torch.autocast(cpu, bfloat16) against its fp32 run.
    ora2 = orec.Recognizer3D(cfg['num_classes'], dropout_ratio=0.0)
    ofill.fill_module(ora2)
    ora2.train()
    with torch.autocast('cpu', dtype=torch.bfloat16):
        o2 = ora2.train_step(dict(imgs=imgs, label=label))
    o2['loss'].backward()
    # (the host's bf16 convolution backward has been seen to return non-finite values for single tensors: those are left out)
    keep = [torch.isfinite(p.grad).all().item() for p in ora2.parameters()]
    pick = lambda parts: torch.cat([x for x, k in zip(parts, keep) if k])
    flat_a = pick([p.grad.float().reshape(-1) for p in ora2.parameters()])
    assert sum(keep) >= len(keep) - 3
    yard = cos(flat_a, pick(flat_o), dim=0).item()
    got = cos(pick(flat_h), pick(flat_o), dim=0).item()
    assert got >= min(0.995, yard - 0.03) and got >= 0.9 and worst[0] >= 0.7, (got, yard, worst)
    gn = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in model.parameters())).item()
    assert abs(gn - gold['grad_norm']) <= 0.03 * gold['grad_norm']
    # running statistics moved as the reference's did, then evaluation mode uses them
    rm = model.backbone.stem[1].running_mean[:4].cpu()
    assert (rm - torch.tensor(gold['running_mean_stem'])).abs().max() < 2e-3
    model.eval()
    probs = torch.as_tensor(model(test_imgs.to(dev), return_loss=False))
    assert probs.shape == (cfg['B'], cfg['num_classes'])
    assert (probs - torch.tensor(gold['probs'])).abs().max().item() < 0.02
    assert (probs.sum(1) - 1).abs().max().item() < 1e-4
    probs2 = torch.as_tensor(model(test_imgs.to(dev), return_loss=False))
    assert torch.equal(probs, probs2)                               # evaluation mode writes nothing


def test_ssl_pretrain_loading_feature_extraction_and_finetuning(dev):
    """the link from pre-training to the consumer (recognizers/base.py:191-205, test_ssv2_r18.py:24-27): the RGB query
    encoder of an MSCLWithAug checkpoint loads under prefix 'recognizer.encoder_q'; features come out (N, 512); a few
    supervised steps with dropout on a fixed batch reduce the loss."""
    from mscl_amd import ClipSGD, Config, build_model
    from mscl_amd.fill import fill_module
    root = os.path.join(os.path.dirname(__file__), '..')
    cfg = Config.fromfile(os.path.join(root, 'configs/recognition/moco/mscl_r18_cosm_lr2e-2.py'))
    cfg.model.recognizer.K = cfg.model.recognizer_flow.K = 64
    pre = build_model(cfg.model)
    fill_module(pre)
    ckpt = {k: v.clone() for k, v in pre.state_dict().items()}
    model = build(7, 0.5, dev, test_cfg=dict(average_clips='score', feature_extraction=True))
    missing, unexpected = model.init_from_ssl_pretrain('backbone', dict(ckpt), dict(prefix='recognizer.encoder_q'))
    assert not missing and not unexpected
    for k, v in model.backbone.state_dict().items():
        assert torch.equal(v.cpu(), ckpt['recognizer.encoder_q.' + k]), k
    g = torch.Generator().manual_seed(3)
    imgs = torch.randn((4, 1, 3, 8, 32, 32), generator=g).to(dev)
    label = torch.tensor([[0], [3], [6], [3]], device=dev)
    model.eval()
    feat = model._do_test(imgs)
    assert feat.shape == (4, 512) and torch.isfinite(feat).all()
    model.feature_extraction = False
    opt = ClipSGD(model, lr=0.05, momentum=0.9, weight_decay=1e-4, grad_clip=dict(max_norm=40, norm_type=2))
    model.train()
    losses = []
    for _ in range(12):
        out = model.train_step(dict(imgs=imgs, label=label))
        opt.zero_grad()
        out['loss'].backward()
        opt.step()
        losses.append(out['log_vars']['loss_cls'])
    # the loss of a step is taken WITH that step's dropout mask (p = 0.5 on 512 features of 4 clips): once the batch is fitted most
    # steps read ~0.001 and one in ten spikes (0.1 ... 1.4 seen: a mask that removes the features the fit leans on), so the claim
    # "fine-tuning reduces the loss" is read off the median of the last five steps, not off the last one (round 6: the last-step form
    # failed once in 16 fresh processes, profiles/r06_probe_runs.txt)
    assert sorted(losses[-5:])[2] < 0.5 * losses[0], losses
    with pytest.raises(NotImplementedError):
        build_model(dict(type='Recognizer3D', backbone=dict(type='ResNet3dSlowOnly'), cls_head=None))


def test_retrieval_on_extracted_features(dev):
    """SURVEY section 8(f)#4, retrieval (tools/test_retrival.py:258-303): features of a 'train' and a 'test' split from the HIP
    trunk in evaluation mode vs the oracle's (cosine >= 0.995 per video), and the k-NN accuracies computed from either agree
    (a video's rank may move by bf16 noise: at most one test video of 12 may flip per k)."""
    from mscl_amd import retrieval
    from oracle import fill as ofill, recognizer3d as orec, retrieval as oret
    model = build(5, 0.0, dev, test_cfg=dict(average_clips=None, feature_extraction=True))
    ora = orec.Recognizer3D(5, dropout_ratio=0.0)
    ofill.fill_module(ora)
    model.train(); ora.train()
    g = torch.Generator().manual_seed(3)
    warm = torch.randn((4, 1, 3, 8, 32, 32), generator=g)               # one training step each: running statistics off their init
    model.train_step(dict(imgs=warm.to(dev), label=torch.tensor([[0], [1], [2], [3]], device=dev)))
    ora.train_step(dict(imgs=warm, label=torch.tensor([[0], [1], [2], [3]])))
    n_train, n_test, classes = 24, 12, 4
    base = torch.randn((classes, 1, 3, 8, 32, 32), generator=g)
    tl = torch.arange(n_train) % classes
    sl = torch.arange(n_test) % classes
    tr = base[tl] + 0.5 * torch.randn((n_train, 1, 3, 8, 32, 32), generator=g)
    te = base[sl] + 0.5 * torch.randn((n_test, 1, 3, 8, 32, 32), generator=g)
    batches = lambda x: [dict(imgs=x[i:i + 8].to(dev)) for i in range(0, len(x), 8)]
    f_tr, f_te = retrieval.extract_features(model, batches(tr)), retrieval.extract_features(model, batches(te))
    assert model.training and tuple(f_tr.shape) == (n_train, 512)
    ora.eval()
    o_tr, o_te = ora.forward_test(tr, feature_extraction=True), ora.forward_test(te, feature_extraction=True)
    cos = torch.nn.functional.cosine_similarity
    assert cos(f_tr.cpu(), o_tr, dim=1).min().item() >= 0.995 and cos(f_te.cpu(), o_te, dim=1).min().item() >= 0.995
    ks = (1, 5, 10, 20)
    got = retrieval.knn_accuracy(f_tr, tl, f_te, sl, ks)
    want = oret.knn_accuracy(o_tr.numpy(), tl.numpy(), o_te.numpy(), sl.numpy(), ks)
    for k in ks:
        assert abs(got[k] - want[k]) <= 1.0 / n_test + 1e-6, (k, got, want)
    assert got[20] >= got[1] and want[1] > 1.0 / classes                 # the clustered clips are retrievable at all
    plain = build(5, 0.0, dev)
    with pytest.raises(ValueError):
        retrieval.extract_features(plain, batches(te))
