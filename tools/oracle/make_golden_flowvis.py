"""G7: the reference's FlowVisualizer on a small uv grid -> tests/golden/flowvis_g7.npz.

Dev-container only: imports /root/reference/mmaction/models/common/ssl_aug.py (kornia / torchvision replaced by
MagicMock: the visualiser itself only needs torch + the reference's own colour wheel).  The oracle restatement
(oracle/flowvis.py) is asserted bit-identical while generating.
"""
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
from oracle import flowvis as ofv                     # noqa: E402


def uv_grid():
    """2 clips x 4 frames x 8x8: radii below and above 1, exact zeros, the atan2 branch cut (v = 0, u > 0 / -0.0),
    axis-aligned and diagonal directions, plus seeded noise."""
    g = torch.Generator().manual_seed(7)
    uv = (torch.rand((2, 2, 4, 8, 8), generator=g) * 2 - 1) * 1.6
    uv[0, :, 0, 0, :] = 0.0                                          # zero flow
    uv[0, 0, 0, 1, :] = torch.linspace(0.1, 2.0, 8); uv[0, 1, 0, 1, :] = 0.0      # +u axis: atan2(-0, -u) = +-pi
    uv[0, 0, 0, 2, :] = torch.linspace(0.1, 2.0, 8); uv[0, 1, 0, 2, :] = -0.0
    uv[0, 0, 0, 3, :] = -torch.linspace(0.1, 2.0, 8); uv[0, 1, 0, 3, :] = 0.0     # -u axis
    uv[0, 0, 0, 4, :] = 0.0; uv[0, 1, 0, 4, :] = torch.linspace(-2.0, 2.0, 8)      # v axis
    d = torch.linspace(-1.5, 1.5, 8)
    uv[0, 0, 0, 5, :] = d; uv[0, 1, 0, 5, :] = d                                   # diagonals
    uv[0, 0, 0, 6, :] = d; uv[0, 1, 0, 6, :] = -d
    ang = torch.linspace(0, 2 * 3.14159265, 8)
    uv[0, 0, 0, 7, :] = torch.cos(ang); uv[0, 1, 0, 7, :] = torch.sin(ang)         # radius exactly ~1
    return uv


def main():
    rh.install()
    for name in ('kornia', 'kornia.augmentation', 'kornia.augmentation.utils', 'kornia.filters', 'torchvision.transforms',
                 'torchvision.datasets', 'torchvision.datasets.video_utils'):
        sys.modules.setdefault(name, MagicMock())
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    sys.modules.setdefault('mmaction.models.common.motion_map_calculator', MagicMock())
    mod = importlib.import_module('mmaction.models.common.ssl_aug')
    ref_vis, orc_vis = mod.FlowVisualizer(), ofv.FlowVisualizer()
    assert np.array_equal(ref_vis.colorwheel.numpy(), ofv.make_colorwheel())
    uv = uv_grid()
    out_ref = ref_vis(uv.clone())
    out_orc = orc_vis(uv.clone())
    assert torch.equal(out_ref, out_orc), 'oracle/flowvis.py differs from the reference'
    g = torch.Generator().manual_seed(11)                    # a larger random case for statistics
    uv2 = torch.randn((2, 2, 8, 32, 32), generator=g) * 0.8
    out2 = ref_vis(uv2.clone())
    assert torch.equal(out2, orc_vis(uv2.clone()))
    lv = lambda t: torch.round(t * 255).to(torch.uint8).numpy()
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'flowvis_g7.npz'), uv=uv.numpy(), levels=lv(out_ref),
                        uv2=uv2.numpy(), levels2=lv(out2), colorwheel=ref_vis.colorwheel.numpy())
    print('wrote flowvis_g7.npz', out_ref.shape, out2.shape)


if __name__ == '__main__':
    main()
