"""Input data path for real (non-synthetic) training, MI355X-first:

  host   a plain-file clip store (memory-mapped .npy, replacing the reference's Megvii-internal nori / redis storage,
         datasets/pipelines/loading.py:1814-1900 `NoriDecode`; README.md:36-39 lists that replacement as the authors' TODO),
         motion-dense frame sampling, crop-box and rotation-chunk draws -- integers only;
  PCIe   the SELECTED raw frames as uint8 (a quarter of the fp32 bytes), raw (u, v) flow as fp32, through a ring of pinned
         buffers filled by a prefetch thread while the GPU trains on the previous batch;
  GPU    Flow Rotation Augmentation at full resolution (mscl_flow_fra_visualize), then crop + resize + normalise of both
         views of both modalities in one pass each (mscl_crop_resize_u8 / _f32); colour-wheel visualisation, flip, colour
         jitter and blur follow inside `MSCLWithAug.train_step` (augment.py) as in the reference's GPU augmentation.

The pipeline section of the reference config loads unchanged: `MSCLPipeline.from_cfg(cfg.train_pipeline)` reads the
parameters of `MatchFlow`, `TemporalShiftChosenSampleFrames` / `ChosenSampleFrames`, `NormFlowWithStidedAug`,
`MoCoRandomResizedCrop`, `MoCoResize` (configs/recognition/moco/mscl_r18_cosm_lr2e-2.py:66-77) and ignores the steps the
GPU path absorbs (`NoriDecode`, `MoCoNormalize`, `Collect`, `ToTensor`).

ref: datasets/pipelines/loading_mscl.py:52-69,111-283; loading.py:137-177; moco_augmentations.py:11-354;
transforms_motion.py:103-142.  Random draws follow the reference's call order on numpy's and Python's generators (the
reference uses the global ones; a loader owns private, seedable ones), pinned by tests/golden/datapath_g11.json.

View layout, reproduced as the reference's pipeline produces it (not as the model's variable names suggest): the sampler
yields a query clip and a key clip (2T frame indices); FRA appends the rotated copies of ALL 2T flow frames after the 2T
base frames (merge_aug=True); MoCoRandomResizedCrop then halves both lists -- RGB: q = query clip, k = key clip; flow:
q = base flow of (query clip, key clip), k = rotated flow of (query clip, key clip) -- and crops q and k with two
independent boxes (moco_augmentations.py:166-190).  `flow_imgs[i]` therefore holds 2T frames, which
MSCLWithAug.forward_train chunks in two along T (recognizers/mscl.py:229-233).
"""
import json
import os
import queue
import random
import threading

import numpy as np
import torch

from . import kernels as K
from .lib import MsclError

_RETIRED = object()          # what a stopped filler leaves in its queue (ClipLoader.__iter__)


# ----------------------------------------------------------------------------------------------- plain-file store
class ClipStore:
    """root/index.json: [{"id", "label", "chosen_idx": [...]}]; root/<id>/frames.npy uint8 (N,H,W,3) decoded RGB frames;
    root/<id>/flow.npy float32 (Nf,2,h,w) raw (u, v) optical flow, Nf = len(range(0, N - adjacent, gap)) (MatchFlow).
    Arrays are memory-mapped: a batch touches only the frames the sampler picked."""

    def __init__(self, root):
        self.root = root
        self.index = json.load(open(os.path.join(root, 'index.json')))
        self._maps = {}

    def __len__(self):
        return len(self.index)

    def arrays(self, i):
        m = self._maps.get(i)
        if m is None:
            d = os.path.join(self.root, self.index[i]['id'])
            m = self._maps[i] = (np.load(os.path.join(d, 'frames.npy'), mmap_mode='r'), np.load(os.path.join(d, 'flow.npy'), mmap_mode='r'))
        return m

    @staticmethod
    def write(root, videos):
        """videos: iterable of dict(id, label, chosen_idx, frames uint8 (N,H,W,3), flow float32 (Nf,2,h,w))"""
        os.makedirs(root, exist_ok=True)
        index = []
        for v in videos:
            d = os.path.join(root, v['id'])
            os.makedirs(d, exist_ok=True)
            np.save(os.path.join(d, 'frames.npy'), np.ascontiguousarray(v['frames'], dtype=np.uint8))
            np.save(os.path.join(d, 'flow.npy'), np.ascontiguousarray(v['flow'], dtype=np.float32))
            index.append(dict(id=v['id'], label=int(v.get('label', 0)), chosen_idx=[int(c) for c in v['chosen_idx']]))
        json.dump(index, open(os.path.join(root, 'index.json'), 'w'))
        return ClipStore(root)


# ----------------------------------------------------------------------------------------------- host-side draws
def _get_train_clips(num_frames, clip_len, frame_interval, num_clips, rng):
    """loading.py:137-177 (keep_tail_frames=False)"""
    ori = clip_len * frame_interval
    avg = (num_frames - ori + 1) // num_clips
    if avg > 0:
        return np.arange(num_clips) * avg + rng.randint(avg, size=num_clips)
    if num_frames > max(num_clips, ori):
        return np.sort(rng.randint(num_frames - ori + 1, size=num_clips))
    if avg == 0:
        return np.around(np.arange(num_clips) * ((num_frames - ori + 1.0) / num_clips))
    return np.zeros((num_clips,), dtype=int)


class ChosenSampleFrames:
    """loading_mscl.py:111-176: one clip whose start is a motion-dense offset (`chosen_idx`, computed offline)"""

    def __init__(self, clip_len, frame_interval=1, num_clips=1, out_of_bound_opt='loop', test_mode=False, **kw):
        if num_clips != 1 or test_mode or out_of_bound_opt != 'loop':
            raise NotImplementedError('the MSCL configs use num_clips=1, out_of_bound_opt="loop", train mode')
        self.clip_len, self.frame_interval, self.n_views = clip_len, frame_interval, 1

    def _offset(self, num_frames, chosen_idx, rng):
        attempt = 0
        while True:
            off = _get_train_clips(num_frames, self.clip_len, self.frame_interval, 1, rng)
            if off[0] in chosen_idx:
                return off
            attempt += 1
            if attempt > 10:
                return np.array([chosen_idx[0] if len(chosen_idx) else 0], dtype=int)    # video is too short

    def offsets(self, num_frames, chosen_idx, rng):
        return self._offset(num_frames, chosen_idx, rng)

    def __call__(self, total_frames, chosen_idx, rng, start_index=0):
        offs = self.offsets(total_frames, chosen_idx, rng)
        inds = offs[:, None] + np.arange(self.clip_len)[None, :] * self.frame_interval
        return np.concatenate(np.mod(inds.reshape((-1, self.clip_len)), total_frames)).astype(int) + start_index


class TemporalShiftChosenSampleFrames(ChosenSampleFrames):
    """loading_mscl.py:179-283: the query clip as above, the key clip at the motion-dense offset picked by the reference's
    rule from a random temporal shift of up to shift_range * clip_len * frame_interval frames"""

    def __init__(self, clip_len, frame_interval=1, num_clips=1, shift_range=1, **kw):
        super().__init__(clip_len, frame_interval, num_clips, **kw)
        self.shift_range, self.n_views = shift_range * clip_len * frame_interval, 2

    def offsets(self, num_frames, chosen_idx, rng):
        off = self._offset(num_frames, chosen_idx, rng)
        tar = off[0] + rng.randint(-self.shift_range, self.shift_range + 1)
        new = 0
        for cid in chosen_idx:
            if abs(cid - tar) < abs(cid - new):       # (sic, loading_mscl.py:236: distance to the candidate itself)
                new = cid
        return np.concatenate((off, np.array([new], dtype=int)), axis=0)


def get_crop_bbox(img_shape, area_range, aspect_ratio_range, rng, pyrng, max_attempts=10):
    """moco_augmentations.py:45-93 -> (x1, y1, x2, y2)"""
    img_h, img_w = img_shape
    area = img_h * img_w
    lo, hi = aspect_ratio_range
    ar = np.exp(rng.uniform(np.log(lo), np.log(hi), size=max_attempts))
    target = rng.uniform(*area_range, size=max_attempts) * area
    cw = np.round(np.sqrt(target * ar)).astype(np.int32)
    ch = np.round(np.sqrt(target / ar)).astype(np.int32)
    for i in range(max_attempts):
        w, h = int(cw[i]), int(ch[i])
        if h <= img_h and w <= img_w:
            x = pyrng.randint(0, img_w - w)
            y = pyrng.randint(0, img_h - h)
            return x, y, x + w, y + h
    s = min(img_h, img_w)
    x, y = (img_w - s) // 2, (img_h - s) // 2
    return x, y, x + s, y + s


def flow_box(box, img_shape, flow_shape):
    """moco_augmentations.py:148-157"""
    h_rate, w_rate = flow_shape[0] / img_shape[0], flow_shape[1] / img_shape[1]
    l, t, r, b = box
    return int(round(l * w_rate)), int(round(t * h_rate)), int(round(r * w_rate)), int(round(b * h_rate))


class MSCLPipeline:
    """the per-sample draws of the training (or validation) pipeline, in the reference's order:
    frame indices -> rotation chunk id -> query box -> key box"""

    def __init__(self, sampler, gap=2, adjacent=8, ratios=(0.2, 1.8), num_chunks=8, area_range=(0.2, 1.0),
                 aspect_ratio_range=(3 / 4, 4 / 3), out_hw=(112, 112)):
        self.sampler, self.gap, self.adjacent = sampler, gap, adjacent
        self.ratios, self.num_chunks = tuple(ratios), num_chunks
        self.area_range, self.aspect_ratio_range, self.out_hw = tuple(area_range), tuple(aspect_ratio_range), tuple(out_hw)

    @classmethod
    def from_cfg(cls, pipeline):
        kw, sampler = {}, None
        for step in pipeline:
            step = dict(step)
            typ = step.pop('type')
            if typ == 'MatchFlow':
                kw.update(gap=step.get('gap', 2), adjacent=step.get('adjacent', 8))
            elif typ in ('TemporalShiftChosenSampleFrames', 'ChosenSampleFrames'):
                sampler = globals()[typ](**step)
            elif typ == 'NormFlowWithStidedAug':
                if not step.get('merge_aug', True):
                    raise NotImplementedError('merge_aug=False (separate rotated-flow key) is not used by the MSCL configs')
                kw.update(ratios=step['ratios'], num_chunks=step['num_chunks'])
            elif typ == 'MoCoRandomResizedCrop':
                kw.update(area_range=step.get('area_range', (0.08, 1.0)), aspect_ratio_range=step.get('aspect_ratio_range', (3 / 4, 4 / 3)))
            elif typ == 'MoCoResize':
                if step.get('keep_ratio', False):
                    raise NotImplementedError('keep_ratio=True is not used by the MSCL configs')
                w, h = step['scale']
                kw.update(out_hw=(h, w))
            elif typ in ('NoriDecode', 'MoCoNormalize', 'Collect', 'ToTensor'):
                continue              # storage / layout steps the GPU path absorbs
            else:
                raise NotImplementedError(f'pipeline step {typ} is outside the MSCL data path')
        if sampler is None:
            raise ValueError('the pipeline names no frame sampler')
        return cls(sampler, **kw)

    def draw(self, n_raw_frames, chosen_idx, img_shape, flow_shape, rng, pyrng):
        total = len(range(0, n_raw_frames - self.adjacent, self.gap))           # MatchFlow: frames aligned with the flow
        inds = self.sampler(total, chosen_idx, rng)                              # into the gap-subsampled sequence
        cid = int(rng.randint(0, self.num_chunks))                               # transforms_motion.py:121
        boxes = [get_crop_bbox(img_shape, self.area_range, self.aspect_ratio_range, rng, pyrng) for _ in range(2)]
        return dict(flow_inds=inds, frame_inds=inds * self.gap, cid=cid, box_q=boxes[0], box_k=boxes[1],
                    fbox_q=flow_box(boxes[0], img_shape, flow_shape), fbox_k=flow_box(boxes[1], img_shape, flow_shape))


# ----------------------------------------------------------------------------------------------- loader
class _Slot:
    def __init__(self, B, T2, img_hw, flow_hw):
        self.frames = torch.empty((B, T2, img_hw[0], img_hw[1], 3), dtype=torch.uint8).pin_memory()
        self.flow = torch.empty((B, 2, T2, flow_hw[0], flow_hw[1]), dtype=torch.float32).pin_memory()
        self.ints = torch.empty((B, 17), dtype=torch.int32).pin_memory()      # cid | box_q | box_k | fbox_q | fbox_k
        self.event = None                  # recorded after the slot's upload was queued on the consumer's stream
        self.free = threading.Event()      # set by the consumer once that event exists (or the slot was never used)
        self.free.set()
        self.labels = None


class ClipPairLoader:
    """iterable of `data_batch` dicts for MSCLWithAug.train_step: imgs = [q, k] (B,3,T,Ho,Wo) fp32 in [0,1],
    flow_imgs = [q, k] (B,2,2T,Ho,Wo) fp32 (u, v) -- device tensors, produced as described in the module header.
    All videos of a store must share the raw frame size and the flow size (as the reference's pre-resized storage does)."""

    def __init__(self, store, pipeline, batch_size, device, seed=0, shuffle=True, drop_last=True, slots=3, rank=0, world=1):
        self.store, self.pipe, self.B, self.device = store, pipeline, batch_size, torch.device(device)
        self.shuffle, self.drop_last, self.rank, self.world = shuffle, drop_last, rank, world
        self.rng, self.pyrng = np.random.RandomState(seed + 1000 * rank), random.Random(seed + 1000 * rank)
        self.order_rng = np.random.RandomState(seed)                  # the epoch permutation is shared by all ranks
        fr, fl = store.arrays(0)
        self.img_hw, self.flow_hw = tuple(fr.shape[1:3]), tuple(fl.shape[2:4])
        self.T = pipeline.sampler.clip_len
        self.n_clips = pipeline.sampler.n_views          # 2: query clip + key clip (training); 1: validation pipeline
        self._slots = [_Slot(batch_size, self.n_clips * self.T, self.img_hw, self.flow_hw) for _ in range(slots)]

    def __len__(self):
        n = len(self.store) // self.world
        return n // self.B if self.drop_last else (n + self.B - 1) // self.B

    def _fill(self, slot, ids):
        """host part: draws + gather of the picked frames into the pinned slot (no per-pixel arithmetic)"""
        labels = []
        for b, vid in enumerate(ids):
            frames, flow = self.store.arrays(vid)
            meta = self.store.index[vid]
            if tuple(frames.shape[1:3]) != self.img_hw or tuple(flow.shape[2:4]) != self.flow_hw:
                raise MsclError(f'video {meta["id"]}: frame / flow size differs from the store\'s first video')
            d = self.pipe.draw(frames.shape[0], meta['chosen_idx'], self.img_hw, self.flow_hw, self.rng, self.pyrng)
            fr_dst, fl_dst = slot.frames[b].numpy(), slot.flow[b].numpy()
            for t, (fi, gi) in enumerate(zip(d['frame_inds'], d['flow_inds'])):
                fr_dst[t] = frames[fi]
                fl_dst[:, t] = flow[gi]
            slot.ints[b] = torch.tensor([d['cid'], *d['box_q'], *d['box_k'], *d['fbox_q'], *d['fbox_k']], dtype=torch.int32)
            labels.append(meta['label'])
        slot.labels = labels
        return slot

    def _to_device(self, slot):
        """PCIe + the GPU part; returns the data_batch.  Runs on the consumer's (current) stream."""
        dev, T = self.device, self.T
        labels = list(slot.labels)                                   # read BEFORE the slot is released below: the filler may refill it at once
        nb = len(labels)                                             # a short last batch (drop_last=False) travels as its rows only
        frames = slot.frames[:nb].to(dev, non_blocking=True)
        flow = slot.flow[:nb].to(dev, non_blocking=True)
        ints = slot.ints[:nb].to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        slot.event = ev                                              # the filler waits for it before reusing the slot
        slot.free.set()
        cid = ints[:, 0].contiguous()
        box = lambda i: ints[:, 1 + 4 * i:5 + 4 * i].contiguous()
        hw = self.pipe.out_hw
        # FRA at full resolution on every sampled frame (the division by each frame's own maximum radius precedes the crop,
        # as in the reference); `normed` holds the base frames, then their rotated copies, (u, v) last
        _, _, normed = K.flow_fra_visualize(flow, cid, self.pipe.ratios, self.pipe.num_chunks, want_debug=True)
        if self.n_clips == 2:         # training: RGB q / k = the two clips; flow q = base of both clips, k = rotated of both
            imgs = [K.crop_resize(frames[:, :T], box(0), hw), K.crop_resize(frames[:, T:], box(1), hw)]
            flows = [K.crop_resize(normed[:, :2 * T], box(2), hw), K.crop_resize(normed[:, 2 * T:], box(3), hw)]
        else:                         # validation (one clip): both views see the same frames / base||rotated, two boxes
            imgs = [K.crop_resize(frames, box(0), hw), K.crop_resize(frames, box(1), hw)]
            flows = [K.crop_resize(normed, box(2), hw), K.crop_resize(normed, box(3), hw)]
        return dict(imgs=imgs, flow_imgs=flows, label=torch.tensor(labels, device=dev))

    def _retire_worker(self, st=None):
        """stop and join the filler of an epoch the consumer walked away from (break / exception mid-epoch): it may be blocked
        in q.put or on a slot, and it shares the slots and the random streams with the next epoch's filler.
        st = the (stop, thread, queue) of ONE epoch: a generator retires its own filler only -- an old epoch's generator that is
        finalised late (kept alive by a traceback or a dropped `iter(loader)` handle) finds its filler already joined by the next
        `__iter__` and must touch neither the slots nor the filler of the epoch that is running now."""
        mine = st is not None
        st = st if mine else getattr(self, '_active', None)
        if st is None:
            return
        stop, th, q = st
        stop.set()
        # The slots belong to the epoch that is running now (self._active): only that epoch's retirement -- or the loader's own,
        # st = None -- may touch them.  The reset is gated on OWNERSHIP, not on the filler being alive: a filler that has
        # finished can leave [last batch, None] in the queue, and a consumer that walks away before those batches reach
        # _to_device leaves their slots with `free` cleared -- the next epoch's filler would wait on them for ever.
        owner = (not mine) or getattr(self, '_active', None) is st
        if owner:
            if th.is_alive():
                for slot in self._slots:
                    slot.free.set()                                  # wake a filler waiting for a slot
                while th.is_alive():
                    try:
                        q.get_nowait()                               # make room for a filler blocked in q.put
                    except queue.Empty:
                        pass
                    th.join(timeout=0.01)
            for slot in self._slots:                                 # uploads already queued must finish before a slot is refilled
                if slot.event is not None:
                    slot.event.synchronize()
                slot.free.set()
        # whoever still holds this epoch's generator finds ONE word in its queue: "retired" (the drain above may have eaten the
        # filler's own, and a consumer must never block on a queue nobody fills)
        while True:
            try:
                q.get_nowait()
            except queue.Empty:
                break
        q.put_nowait(_RETIRED)
        if getattr(self, '_active', None) is st:
            self._active = None

    def __iter__(self):
        self._retire_worker()
        n = len(self.store)
        order = self.order_rng.permutation(n) if self.shuffle else np.arange(n)
        order = order[self.rank::self.world]
        nb = len(self)
        batches = [order[i * self.B:(i + 1) * self.B] for i in range(nb)]
        q = queue.Queue(maxsize=max(1, len(self._slots) - 1))
        stop = threading.Event()

        def retired():
            # a stopped filler still leaves a word for whoever may be waiting on ITS queue: a consumer never blocks for ever
            try:
                q.put_nowait(_RETIRED)
            except queue.Full:
                pass

        def worker():
            try:
                for i, ids in enumerate(batches):
                    slot = self._slots[i % len(self._slots)]
                    slot.free.wait()                                 # the consumer has queued this slot's previous upload ...
                    if stop.is_set():
                        return retired()
                    slot.free.clear()
                    if slot.event is not None:
                        slot.event.synchronize()                     # ... and the copy engine has finished reading it
                    q.put(self._fill(slot, ids))
                    if stop.is_set():
                        return retired()
                q.put(None)
            except BaseException as e:      # noqa: BLE001 -- surfaced in the consumer
                q.put(e)
        th = threading.Thread(target=worker, daemon=True)
        mine = self._active = (stop, th, q)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if item is _RETIRED:
                    raise MsclError('this epoch of the loader was retired (a newer iter(loader) took over its slots)')
                if isinstance(item, BaseException):
                    raise item
                yield self._to_device(item)
        finally:
            self._retire_worker(mine)       # normal end: the filler has returned; abandoned epoch (GeneratorExit): stop THIS epoch's filler
