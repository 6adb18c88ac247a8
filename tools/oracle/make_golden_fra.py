"""G8: the reference's NormFlowWithStidedAug (Flow Rotation Augmentation) + FlowVisualizer on seeded uv frames
-> tests/golden/fra_g8.npz.  Dev-container only (imports /root/reference/mmaction/datasets/pipelines/transforms_motion.py
and models/common/ssl_aug.py under the harness).  oracle/flowaug.py is asserted bit-identical while generating."""
import importlib
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness as rh                              # noqa: E402
from oracle import flowaug, flowvis                   # noqa: E402


def main():
    rh.install()
    for name in ('kornia', 'kornia.augmentation', 'kornia.augmentation.utils', 'kornia.filters', 'torchvision.transforms',
                 'torchvision.datasets', 'torchvision.datasets.video_utils'):
        sys.modules.setdefault(name, MagicMock())
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    sys.modules.setdefault('mmaction.models.common.motion_map_calculator', MagicMock())
    rh._shell('mmaction.datasets', f'{rh.REF}/mmaction/datasets')
    rh._shell('mmaction.datasets.pipelines', f'{rh.REF}/mmaction/datasets/pipelines')
    rh._mod('mmaction.datasets.builder', PIPELINES=rh.Registry('pipeline'))
    tm = importlib.import_module('mmaction.datasets.pipelines.transforms_motion')
    vis = importlib.import_module('mmaction.models.common.ssl_aug').FlowVisualizer()
    B, T, H, W = 3, 4, 16, 20
    g = np.random.default_rng(3)
    uv = (g.standard_normal((B, T, H, W, 2)) * np.array([3.0, 0.4, 12.0]).reshape(B, 1, 1, 1, 1)).astype(np.float32)
    uv[1, 2] = 0.0                                           # an all-zero frame: rad_max = 0 -> division by 1e-5
    aug = tm.NormFlowWithStidedAug(ratios=(0.2, 1.8), num_chunks=8, merge_aug=True)
    cids, outs, levels = [], [], []
    for b in range(B):
        np.random.seed(100 + b)
        res = aug({'flows': [uv[b, t] for t in range(T)]})
        cid = int(res['ap_labels'])
        mine = flowaug.fra([uv[b, t] for t in range(T)], cid)
        for a, m in zip(res['flow_imgs'], mine):
            assert np.array_equal(a, m), 'oracle/flowaug.py differs from the reference'
        imgs = np.stack(res['flow_imgs'])                   # (2T, H, W, 2)
        cids.append(cid); outs.append(imgs)
        clip = torch.from_numpy(imgs.astype(np.float32)).permute(3, 0, 1, 2)[None]       # (1, 2, 2T, H, W): ToTensor + collate
        lv = vis(clip)
        assert torch.equal(lv, flowvis.FlowVisualizer()(clip))
        levels.append(torch.round(lv[0] * 255).to(torch.uint8).numpy())                  # (3, 2T, H, W)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'fra_g8.npz'), uv=uv, cid=np.array(cids),
                        normed=np.stack(outs).astype(np.float64), levels=np.stack(levels))
    print('wrote fra_g8.npz', cids, np.stack(outs).dtype)


if __name__ == '__main__':
    main()
