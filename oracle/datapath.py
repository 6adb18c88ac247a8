"""Oracle of the host data path.  TEST INFRASTRUCTURE ONLY.

  sampling   ref: datasets/pipelines/loading.py:137-177 (SampleFrames._get_train_clips),
                  loading_mscl.py:111-176 (ChosenSampleFrames), :179-283 (TemporalShiftChosenSampleFrames), :52-69 (MatchFlow)
  crop box   ref: datasets/pipelines/moco_augmentations.py:45-93 (get_crop_bbox), :110-163 (single_cal, flow box)
  resize     mmcv.imresize -> cv2.resize(INTER_LINEAR).  OpenCV is third-party, not vendored and not installed here: its
             published algorithm (imgproc/resize.cpp) is restated; "parity unpinned" for this function.
The reference draws from numpy's and Python's GLOBAL generators; the restatements draw the same numbers in the same order
from the generators passed in (defaults: the globals), which is what tests/golden/datapath_g11.json pins.
"""
import random as _pyrandom

import numpy as np


def match_flow(ids, gap=2, adjacent=8):
    """loading_mscl.py:52-69: frame ids kept so that they align with the pre-computed flow"""
    return [ids[i] for i in range(0, len(ids) - adjacent, gap)]


def get_train_clips(num_frames, clip_len, frame_interval, num_clips=1, rng=np.random):
    """loading.py:137-177 with keep_tail_frames=False"""
    ori = clip_len * frame_interval
    avg = (num_frames - ori + 1) // num_clips
    if avg > 0:
        return np.arange(num_clips) * avg + rng.randint(avg, size=num_clips)
    if num_frames > max(num_clips, ori):
        return np.sort(rng.randint(num_frames - ori + 1, size=num_clips))
    if avg == 0:
        return np.around(np.arange(num_clips) * ((num_frames - ori + 1.0) / num_clips))
    return np.zeros((num_clips,), dtype=int)


def chosen_offset(num_frames, chosen_idx, clip_len, frame_interval, rng=np.random):
    """loading_mscl.py:121-136: a random clip start that is one of the motion-dense starts (11 attempts, then the first)"""
    attempt = 0
    while True:
        off = get_train_clips(num_frames, clip_len, frame_interval, 1, rng)
        if off[0] in chosen_idx:
            return off
        attempt += 1
        if attempt > 10:
            return np.array([chosen_idx[0] if len(chosen_idx) else 0], dtype=int)


def temporal_shift_chosen_sample(total_frames, chosen_idx, clip_len, frame_interval, shift_range=1, start_index=0, rng=np.random):
    """loading_mscl.py:179-283: frame indices of the query clip then the key clip (2 * clip_len ints)"""
    span = shift_range * clip_len * frame_interval
    off = chosen_offset(total_frames, chosen_idx, clip_len, frame_interval, rng)
    shift = rng.randint(-span, span + 1)
    tar = off[0] + shift
    new = 0
    for cid in chosen_idx:
        if abs(cid - tar) < abs(cid - new):           # (sic: distance to the candidate itself, loading_mscl.py:236)
            new = cid
    offs = np.concatenate((off, np.array([new], dtype=int)), axis=0)
    inds = offs[:, None] + np.arange(clip_len)[None, :] * frame_interval
    inds = np.mod(inds.reshape((-1, clip_len)), total_frames)
    return np.concatenate(inds).astype(int) + start_index


def chosen_sample(total_frames, chosen_idx, clip_len, frame_interval, start_index=0, rng=np.random):
    """loading_mscl.py:111-176 (validation pipeline): one clip"""
    off = chosen_offset(total_frames, chosen_idx, clip_len, frame_interval, rng)
    inds = off[:, None] + np.arange(clip_len)[None, :] * frame_interval
    inds = np.mod(inds.reshape((-1, clip_len)), total_frames)
    return np.concatenate(inds).astype(int) + start_index


def get_crop_bbox(img_shape, area_range=(0.2, 1.0), aspect_ratio_range=(3 / 4, 4 / 3), max_attempts=10, rng=np.random, pyrng=_pyrandom):
    """moco_augmentations.py:45-93 -> (x1, y1, x2, y2)"""
    img_h, img_w = img_shape
    area = img_h * img_w
    lo, hi = aspect_ratio_range
    ar = np.exp(rng.uniform(np.log(lo), np.log(hi), size=max_attempts))
    target = rng.uniform(*area_range, size=max_attempts) * area
    cw = np.round(np.sqrt(target * ar)).astype(np.int32)
    ch = np.round(np.sqrt(target / ar)).astype(np.int32)
    for i in range(max_attempts):
        w, h = cw[i], ch[i]
        if h <= img_h and w <= img_w:
            x = pyrng.randint(0, img_w - w)
            y = pyrng.randint(0, img_h - h)
            return x, y, x + w, y + h
    s = min(img_h, img_w)
    x, y = (img_w - s) // 2, (img_h - s) // 2
    return x, y, x + s, y + s


def flow_box(box, img_shape, flow_shape):
    """moco_augmentations.py:148-157: the image box at the flow's resolution"""
    h_rate, w_rate = flow_shape[0] / img_shape[0], flow_shape[1] / img_shape[1]
    l, t, r, b = box
    return int(round(l * w_rate)), int(round(t * h_rate)), int(round(r * w_rate)), int(round(b * h_rate))


def _taps(n_dst, n_src, clamp_no_neighbour):
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * (float(n_src) / float(n_dst)) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_no_neighbour:
        lo = s < 0
        s[lo] = 0; f[lo] = 0
        hi = s >= n_src - 1
        s[hi] = n_src - 1; f[hi] = 0
        return s, np.minimum(s + 1, n_src - 1), f
    return np.clip(s, 0, n_src - 1), np.clip(s + 1, 0, n_src - 1), f


def resize_u8(img, out_w, out_h):
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_LINEAR) for a uint8 HWC image (restated, see the module header)"""
    h, w = img.shape[:2]
    a = img.astype(np.int64)
    if w == 2 * out_w and h == 2 * out_h:
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x0, x1, fx = _taps(out_w, w, True)
    y0, y1, fy = _taps(out_h, h, False)
    a1 = np.rint(fx * np.float32(2048)).astype(np.int64); a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
    b1 = np.rint(fy * np.float32(2048)).astype(np.int64); b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int64)
    hor = a[:, x0] * a0[None, :, None] + a[:, x1] * a1[None, :, None]            # (h, out_w, c), scaled by 2048
    S0, S1 = hor[y0], hor[y1]
    d = (((b0[:, None, None] * (S0 >> 4)) >> 16) + ((b1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(d, 0, 255).astype(np.uint8)


def resize_f32(img, out_w, out_h):
    """the float path of the same call (flow maps): float taps, horizontal then vertical"""
    h, w = img.shape[:2]
    a = img.astype(np.float32)
    x0, x1, fx = _taps(out_w, w, True)
    y0, y1, fy = _taps(out_h, h, False)
    a1, a0 = fx[None, :, None], (np.float32(1) - fx)[None, :, None]
    hor = a[:, x0] * a0 + a[:, x1] * a1
    b1, b0 = fy[:, None, None], (np.float32(1) - fy)[:, None, None]
    return (hor[y0] * b0 + hor[y1] * b1).astype(np.float32)


def crop_resize_normalize(frames, box, out_hw, u8=True):
    """one view of one sample: frames (T,H,W,C) -> (C,T,Ho,Wo) fp32 (moco_augmentations.py:110-163, 236-321, 324-354)"""
    x1, y1, x2, y2 = box
    out = []
    for f in frames:
        c = f[y1:y2, x1:x2]
        r = resize_u8(c, out_hw[1], out_hw[0]) if u8 else resize_f32(c, out_hw[1], out_hw[0])
        r = r.astype(np.float32)
        if u8:
            r = r / np.float32(255.0)
        out.append(r)
    return np.stack(out).transpose(3, 0, 1, 2)
