"""G11: the reference's host data-path classes run here with seeded GLOBAL generators:
`MatchFlow`, `ChosenSampleFrames`, `TemporalShiftChosenSampleFrames` (datasets/pipelines/loading_mscl.py),
`MoCoRandomResizedCrop.get_crop_bbox` / `single_cal`'s flow box (moco_augmentations.py) and the cid draw of
`NormFlowWithStidedAug` (transforms_motion.py:121), in the order one training sample consumes them.
Writes tests/golden/datapath_g11.json (inputs regenerate from the seeds); asserts oracle/datapath.py equal while doing so.

Absent third-party imports of those files (cv2, nori2, mmcv.fileio, torch DataContainer ...) are MagicMock'd: none of
them is touched by the index / box arithmetic recorded here.  The resize itself (cv2.resize) cannot be run: unpinned.
Development container only."""
import importlib
import json
import os
import random
import sys
import types
from unittest.mock import MagicMock

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness as rh                                     # noqa: E402
from oracle import datapath as odp                           # noqa: E402

np.int = int                                                  # the reference predates numpy 1.24 (loading_mscl.py:133)
rh.install()
for name in ('cv2', 'nori2', 'mmcv.fileio', 'mmcv.parallel'):
    sys.modules[name] = MagicMock()
sys.modules['mmcv'].is_tuple_of = lambda seq, typ: isinstance(seq, tuple) and all(isinstance(s, typ) for s in seq)
sys.modules['mmcv.utils'].digit_version = lambda v: tuple(int(x) for x in v.split('.')[:3] if x.isdigit())
sys.modules['mmcv'].__version__ = '1.4.0'
REF = rh.REF
pipelines_reg = rh.Registry('pipeline')
rh._shell('mmaction.datasets', f'{REF}/mmaction/datasets')
rh._mod('mmaction.datasets.builder', PIPELINES=pipelines_reg, DATASETS=rh.Registry('dataset'))
rh._shell('mmaction.datasets.pipelines', f'{REF}/mmaction/datasets/pipelines')
rh._mod('mmaction.datasets.pipelines.formating', to_tensor=lambda x: x)
u = sys.modules['mmaction.utils']
for n in ('get_random_string', 'get_shm_dir', 'get_thread_id', 'imdecode'):
    setattr(u, n, MagicMock())
loading_mscl = importlib.import_module('mmaction.datasets.pipelines.loading_mscl')
moco_aug = importlib.import_module('mmaction.datasets.pipelines.moco_augmentations')
tm = importlib.import_module('mmaction.datasets.pipelines.transforms_motion')

CASES = [  # (seed, raw frames, chosen_idx, clip_len, interval, img (h, w), flow (h, w))
    (1, 300, list(range(0, 140, 7)), 8, 8, (128, 170), (128, 170)),
    (2, 300, [3, 50, 51, 90], 8, 8, (128, 170), (64, 85)),
    (3, 90, [0, 5], 8, 8, (112, 112), (112, 112)),           # shorter than one clip: avg_interval <= 0, looped indices
    (4, 200, [], 16, 4, (256, 340), (128, 170)),             # no motion-dense offsets at all
    (5, 137, list(range(0, 64, 3)), 16, 2, (240, 320), (240, 320)),
]
out = dict(cases=[])
for seed, n_raw, chosen, T, itv, img_hw, flow_hw in CASES:
    rec = dict(seed=seed, n_raw=n_raw, chosen_idx=chosen, clip_len=T, frame_interval=itv, img_hw=img_hw, flow_hw=flow_hw, samples=[])
    np.random.seed(seed); random.seed(seed)
    mf = loading_mscl.MatchFlow(gap=2, adjacent=8, flow_key='nids_flow')
    smp = loading_mscl.TemporalShiftChosenSampleFrames(clip_len=T, frame_interval=itv, num_clips=1, shift_range=1)
    fra = tm.NormFlowWithStidedAug(ratios=(0.2, 1.8), num_chunks=8, merge_aug=True)
    crop = moco_aug.MoCoRandomResizedCrop(area_range=(0.2, 1.0), flow_key='flow_imgs')
    for _ in range(6):
        ids = list(range(n_raw))
        res = dict(nori_id_seq=ids, nids_flow=list(range(len(range(0, n_raw - 8, 2)))), chosen_idx=chosen, start_index=0)
        res = mf(res)
        res = smp(res)
        inds = [int(i) for i in res['frame_inds']]
        picked = [res['nori_id_seq'][i] for i in inds]                       # raw frame ids the decoder would fetch
        cid = int(np.random.randint(0, fra.num_chunks))                      # transforms_motion.py:121 (the class's own first draw)
        boxes, fboxes = [], []
        for _v in range(2):                                                  # single_cal for '_q' then '_k'
            l, t, r, b = crop.get_crop_bbox(img_hw, crop.area_range, crop.aspect_ratio_range)
            boxes.append([int(l), int(t), int(r), int(b)])
            hr, wr = flow_hw[0] / img_hw[0], flow_hw[1] / img_hw[1]          # moco_augmentations.py:148-157
            fboxes.append([int(round(l * wr)), int(round(t * hr)), int(round(r * wr)), int(round(b * hr))])
        rec['samples'].append(dict(flow_inds=inds, frame_inds=picked, cid=cid, boxes=boxes, fboxes=fboxes))
    # the oracle restatement, same seeds, same order
    np.random.seed(seed); random.seed(seed)
    for s in rec['samples']:
        total = len(odp.match_flow(list(range(n_raw))))
        oi = odp.temporal_shift_chosen_sample(total, chosen, T, itv)
        assert [int(i) for i in oi] == s['flow_inds'], (seed, oi, s['flow_inds'])
        assert int(np.random.randint(0, 8)) == s['cid']
        for v in range(2):
            bx = odp.get_crop_bbox(img_hw)
            assert [int(x) for x in bx] == s['boxes'][v], (seed, bx, s['boxes'][v])
            assert list(odp.flow_box(bx, img_hw, flow_hw)) == s['fboxes'][v]
    # validation sampler (one clip)
    np.random.seed(seed + 100)
    vs = loading_mscl.ChosenSampleFrames(clip_len=T, frame_interval=itv, num_clips=1)
    val = []
    for _ in range(4):
        res = vs(mf(dict(nori_id_seq=list(range(n_raw)), nids_flow=list(range(len(range(0, n_raw - 8, 2)))), chosen_idx=chosen, start_index=0)))
        val.append([int(i) for i in res['frame_inds']])
    np.random.seed(seed + 100)
    for v in val:
        assert [int(i) for i in odp.chosen_sample(len(odp.match_flow(list(range(n_raw)))), chosen, T, itv)] == v
    rec['val_flow_inds'] = val
    out['cases'].append(rec)
ref_cfg = rh.load_ref_cfg()                                  # the reference's own pipeline sections, to pin the authored config
out['ref_pipelines'] = dict(train_pipeline=ref_cfg['train_pipeline'], val_pipeline=ref_cfg['val_pipeline'],
                            evaluation=ref_cfg['evaluation'], videos_per_gpu=ref_cfg['data']['videos_per_gpu'])
path = os.path.join(ROOT, 'tests', 'golden', 'datapath_g11.json')
json.dump(out, open(path, 'w'))
print('wrote', path, sum(len(c['samples']) for c in out['cases']), 'training samples')
