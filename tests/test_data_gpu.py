"""SURVEY.md section 8(f) row 2 on the GPU: the crop + resize + normalise kernels against the CPU restatement of the
reference's pipeline steps (oracle/datapath.py, whose index / box arithmetic is pinned to the reference's classes by
tests/golden/datapath_g11.json), and the loader end to end: plain-file store -> pinned ring -> GPU views -> train_step."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _boxes(g, B, H, W):
    out = []
    for _ in range(B):
        w, h = g.randint(8, W + 1), g.randint(8, H + 1)
        x, y = g.randint(0, W - w + 1), g.randint(0, H - h + 1)
        out.append([x, y, x + w, y + h])
    return np.array(out, dtype=np.int32)


@pytest.mark.parametrize('H,W,Ho,Wo', [(64, 85, 112, 112), (128, 170, 112, 112), (240, 320, 56, 72), (31, 45, 112, 112)])
def test_crop_resize_u8_bit_exact_vs_oracle(H, W, Ho, Wo, dev):
    """integer arithmetic: the kernel and the restatement must agree on every byte, hence on every float (v / 255);
    boxes include the full frame, an exact 2 x 2 reduction (cv2's area special case) and up- and down-scaling"""
    from mscl_amd import kernels as K
    from oracle import datapath as odp
    g = np.random.RandomState(H * 7 + W)
    B, T = 5, 3
    frames = g.randint(0, 256, (B, T, H, W, 3)).astype(np.uint8)
    boxes = _boxes(g, B, H, W)
    boxes[0] = [0, 0, W, H]
    if 2 * Wo <= W and 2 * Ho <= H:
        boxes[1] = [3, 2, 3 + 2 * Wo, 2 + 2 * Ho]
    boxes[2] = [W - 9, H - 8, W, H]                       # a tiny crop, up-scaled more than tenfold
    got = K.crop_resize(torch.from_numpy(frames).to(dev), torch.from_numpy(boxes).to(dev), (Ho, Wo)).cpu().numpy()
    for b in range(B):
        want = odp.crop_resize_normalize(frames[b], boxes[b], (Ho, Wo), u8=True)
        assert np.array_equal(got[b], want), (b, boxes[b], np.abs(got[b] - want).max())
    assert got.min() >= 0 and got.max() <= 1


def test_crop_resize_f32_and_strided_source(dev):
    """float maps (the (u, v) flow): same taps, float coefficients, cv2's operation order -> equal to the restatement up to
    an fp32 rounding; a source that is a slice along T of a larger upload (batch stride > T*H*W*C) reads the right frames"""
    from mscl_amd import kernels as K
    from mscl_amd.lib import MsclError
    from oracle import datapath as odp
    g = np.random.RandomState(4)
    B, T, H, W = 3, 6, 40, 52
    flow = g.randn(B, T, H, W, 2).astype(np.float32)
    boxes = _boxes(g, B, H, W)
    d = torch.from_numpy(flow).to(dev)
    for sl in (slice(0, T), slice(0, 3), slice(3, 6)):
        got = K.crop_resize(d[:, sl], torch.from_numpy(boxes).to(dev), (28, 36)).cpu().numpy()
        for b in range(B):
            want = odp.crop_resize_normalize(flow[b, sl], boxes[b], (28, 36), u8=False)
            assert np.abs(got[b] - want).max() <= 5e-7 * max(1.0, np.abs(want).max()), (sl, b, np.abs(got[b] - want).max())
    with pytest.raises(MsclError):
        K.crop_resize(d.permute(0, 1, 3, 2, 4), torch.from_numpy(boxes).to(dev), (8, 8))       # not dense in (T,H,W,C)
    with pytest.raises(MsclError):
        K.crop_resize(d, torch.from_numpy(boxes[:2]).to(dev), (8, 8))


def _make_store(root, n_videos, n_frames, hw, flow_hw, seed):
    from mscl_amd.data import ClipStore
    g = np.random.RandomState(seed)
    vids = []
    for i in range(n_videos):
        nf = len(range(0, n_frames - 8, 2))
        base = g.randint(0, 256, (1, hw[0], hw[1], 3))
        frames = np.clip(base + g.randint(-40, 40, (n_frames, hw[0], hw[1], 3)), 0, 255).astype(np.uint8)
        flow = (g.randn(nf, 2, flow_hw[0], flow_hw[1]) * (1 + i)).astype(np.float32)
        vids.append(dict(id=f'v{i:03d}', label=i % 3, chosen_idx=list(range(0, max(1, nf - 64), 5)), frames=frames, flow=flow))
    return ClipStore.write(root, vids), vids


def test_loader_end_to_end_vs_oracle_pipeline(dev, tmp_path):
    """store -> sampler draws -> pinned slots -> GPU FRA + crop/resize: every batch equals the reference pipeline's steps
    composed on the CPU with the same draws (RGB bit-exact; flow to 1e-5: FRA divides in float64 on both sides, the GPU's
    max-radius reduction order differs), in the reference's view layout; then the batches train a step."""
    from mscl_amd import ClipSGD, Config, build_model
    from mscl_amd.data import ClipPairLoader, MSCLPipeline
    from mscl_amd.fill import fill_module
    from oracle import datapath as odp, flowaug
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/recognition/moco/mscl_r18_cosm_lr2e-2.py'))
    T, B, hw, fhw = 4, 2, (40, 52), (20, 26)
    store, vids = _make_store(str(tmp_path / 'store'), 6, 120, hw, fhw, seed=1)
    steps = [dict(s) for s in cfg.train_pipeline]
    for s in steps:
        if s['type'] == 'TemporalShiftChosenSampleFrames':
            s.update(clip_len=T, frame_interval=2)
        if s['type'] == 'MoCoResize':
            s.update(scale=(32, 32))
    pipe = MSCLPipeline.from_cfg(steps)
    loader = ClipPairLoader(store, pipe, B, dev, seed=5, shuffle=True)
    assert len(loader) == 3
    # replay the host draws with the same generators to know what each batch must contain
    order = np.random.RandomState(5).permutation(6)
    rng, pyrng = np.random.RandomState(5), random.Random(5)
    batches = list(loader)
    assert len(batches) == 3
    for bi, batch in enumerate(batches):
        ids = order[bi * B:(bi + 1) * B]
        for b, vid in enumerate(ids):
            v = vids[vid]
            d = pipe.draw(v['frames'].shape[0], v['chosen_idx'], hw, fhw, rng, pyrng)
            fr = v['frames'][d['frame_inds']]
            want_q = odp.crop_resize_normalize(fr[:T], d['box_q'], (32, 32), u8=True)
            want_k = odp.crop_resize_normalize(fr[T:], d['box_k'], (32, 32), u8=True)
            assert np.array_equal(batch['imgs'][0][b].cpu().numpy(), want_q) and np.array_equal(batch['imgs'][1][b].cpu().numpy(), want_k)
            fl = [np.ascontiguousarray(v['flow'][i].transpose(1, 2, 0)) for i in d['flow_inds']]      # (h, w, 2) frames, q clip then k clip
            normed = flowaug.fra(fl, d['cid'])                                                        # 2T base, then 2T rotated
            base = np.stack(normed[:2 * T]).astype(np.float32); rot = np.stack(normed[2 * T:]).astype(np.float32)
            wq = odp.crop_resize_normalize(base, d['fbox_q'], (32, 32), u8=False)
            wk = odp.crop_resize_normalize(rot, d['fbox_k'], (32, 32), u8=False)
            assert np.abs(batch['flow_imgs'][0][b].cpu().numpy() - wq).max() < 1e-5
            assert np.abs(batch['flow_imgs'][1][b].cpu().numpy() - wk).max() < 1e-5
        assert batch['label'].tolist() == [vids[i]['label'] for i in ids]
        assert tuple(batch['imgs'][0].shape) == (B, 3, T, 32, 32) and tuple(batch['flow_imgs'][0].shape) == (B, 2, 2 * T, 32, 32)
    # the batches feed the step unchanged (uv flow goes through the fused visualiser inside train_step)
    cfg.model.sup_head.t = T // 2
    cfg.model.recognizer.K = cfg.model.recognizer_flow.K = 64
    model = build_model(cfg.model); fill_module(model); model.materialize(dev).train()
    opt = ClipSGD.from_cfg(model, cfg.optimizer, cfg.optimizer_config)
    for batch in ClipPairLoader(store, pipe, B, dev, seed=6):
        out = model.train_step(dict(imgs=batch['imgs'], flow_imgs=batch['flow_imgs']))
        opt.zero_grad(); out['loss'].backward(); opt.step()
        assert torch.isfinite(out['loss']).item()
    # validation pipeline: one clip, both views from it, flow views = base || rotated of that clip
    vsteps = [dict(s) for s in cfg.val_pipeline] if 'val_pipeline' in cfg else None
    vpipe = MSCLPipeline.from_cfg([dict(type='MatchFlow', gap=2, adjacent=8), dict(type='ChosenSampleFrames', clip_len=T, frame_interval=2),
                                   dict(type='NormFlowWithStidedAug', ratios=(0.2, 1.8), num_chunks=8), dict(type='MoCoRandomResizedCrop', area_range=(0.2, 1.0)),
                                   dict(type='MoCoResize', scale=(32, 32))])
    vb = next(iter(ClipPairLoader(store, vpipe, B, dev, seed=7, shuffle=False)))
    assert tuple(vb['imgs'][0].shape) == (B, 3, T, 32, 32) and tuple(vb['flow_imgs'][1].shape) == (B, 2, 2 * T, 32, 32)


def test_loader_abandoned_epoch_and_short_last_batch(dev, tmp_path):
    """(round-2 advisor) a consumer that leaves an epoch early must not strand the filler thread on the pinned slots or the random
    streams: the next epoch starts clean and yields every batch; with drop_last=False the short last batch carries its own rows
    only (labels, frames and boxes of equal length)."""
    from mscl_amd import Config
    from mscl_amd.data import ClipPairLoader, MSCLPipeline
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/recognition/moco/mscl_r18_cosm_lr2e-2.py'))
    T, B, hw, fhw = 4, 2, (40, 52), (20, 26)
    store, vids = _make_store(str(tmp_path / 'store'), 7, 120, hw, fhw, seed=2)
    steps = [dict(s) for s in cfg.train_pipeline]
    for s in steps:
        if s['type'] == 'TemporalShiftChosenSampleFrames':
            s.update(clip_len=T, frame_interval=2)
        if s['type'] == 'MoCoResize':
            s.update(scale=(32, 32))
    pipe = MSCLPipeline.from_cfg(steps)
    loader = ClipPairLoader(store, pipe, B, dev, seed=3, shuffle=False, drop_last=False, slots=2)
    assert len(loader) == 4
    for i, batch in enumerate(loader):          # walk away after the first batch: the filler is mid-epoch, blocked on a slot / the queue
        break
    full = list(loader)                         # would deadlock (or race the old filler) without the retire step
    assert [len(b['label']) for b in full] == [2, 2, 2, 1]
    last = full[-1]
    assert tuple(last['imgs'][0].shape) == (1, 3, T, 32, 32) and tuple(last['flow_imgs'][1].shape) == (1, 2, 2 * T, 32, 32)
    assert last['label'].tolist() == [vids[6]['label']]
    assert loader._active is None


def test_loader_old_epoch_finalised_after_a_new_one_started(dev, tmp_path):
    """(round-3 advisor) an old epoch's generator that is kept alive (an `it = iter(loader)` handle, a traceback) and finalised only
    AFTER a newer epoch has started must retire its own -- already joined -- filler and nothing else: the running epoch still
    yields every batch, and a consumer of the retired epoch gets an error instead of waiting for ever."""
    import pytest as _pt
    from mscl_amd import Config
    from mscl_amd.data import ClipPairLoader, MSCLPipeline
    from mscl_amd.lib import MsclError
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/recognition/moco/mscl_r18_cosm_lr2e-2.py'))
    T, B, hw, fhw = 4, 2, (40, 52), (20, 26)
    store, vids = _make_store(str(tmp_path / 'store'), 8, 120, hw, fhw, seed=4)
    steps = [dict(s) for s in cfg.train_pipeline]
    for s in steps:
        if s['type'] == 'TemporalShiftChosenSampleFrames':
            s.update(clip_len=T, frame_interval=2)
        if s['type'] == 'MoCoResize':
            s.update(scale=(32, 32))
    pipe = MSCLPipeline.from_cfg(steps)
    loader = ClipPairLoader(store, pipe, B, dev, seed=3, shuffle=False, slots=2)
    old = iter(loader)
    next(old)                                   # epoch 1 is mid-flight and its handle stays alive
    new = iter(loader)
    first = next(new)                           # epoch 2 starts: it joins epoch 1's filler
    running = loader._active
    assert running is not None
    old.close()                                 # late finalisation of epoch 1: must not stop epoch 2's filler
    assert loader._active is running and not running[0].is_set()
    rest = list(new)
    assert len(rest) == len(loader) - 1 and len(first['label']) == B
    assert loader._active is None
    # a consumer still holding a retired epoch gets an error, not a hang
    old2 = iter(loader)
    next(old2)
    it3 = iter(loader)
    next(it3)
    with _pt.raises((MsclError, StopIteration)):
        for _ in range(len(loader) + 1):
            next(old2)
    it3.close()


@pytest.mark.timeout(180)
def test_loader_walks_away_after_the_filler_has_finished(dev, tmp_path):
    """(round-4 advisor) default ring of 3 slots (queue of 2): the filler can finish and leave [last batch, None] queued.  A
    consumer that breaks on the penultimate batch then drains a batch whose slot never reached _to_device -- its `free` flag stays
    cleared -- so the slot reset must follow OWNERSHIP of the epoch, not whether the filler thread is still alive: the next epoch
    yields every batch instead of waiting on that slot for ever."""
    from mscl_amd import Config
    from mscl_amd.data import ClipPairLoader, MSCLPipeline
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/recognition/moco/mscl_r18_cosm_lr2e-2.py'))
    T, B, hw, fhw = 4, 2, (40, 52), (20, 26)
    store, vids = _make_store(str(tmp_path / 'store'), 8, 120, hw, fhw, seed=6)
    steps = [dict(s) for s in cfg.train_pipeline]
    for s in steps:
        if s['type'] == 'TemporalShiftChosenSampleFrames':
            s.update(clip_len=T, frame_interval=2)
        if s['type'] == 'MoCoResize':
            s.update(scale=(32, 32))
    pipe = MSCLPipeline.from_cfg(steps)
    loader = ClipPairLoader(store, pipe, B, dev, seed=3, shuffle=False, slots=3)
    n = len(loader)
    assert n == 4
    for i, batch in enumerate(loader):
        if i == n - 2:                              # the penultimate batch: wait until the filler has queued the rest and returned
            th = loader._active[1]
            th.join(timeout=30)
            assert not th.is_alive()
            break
    assert all(s.free.is_set() for s in loader._slots)      # the drained batch's slot was handed back
    full = list(loader)
    assert [len(b['label']) for b in full] == [B] * n
    assert loader._active is None
