"""CPU restatement of the reference's deterministic flow visualiser (TEST INFRASTRUCTURE, see oracle/__init__.py).

ref: mmaction/models/common/ssl_aug.py:87-120 (flow_uv_to_colors), :122-136 (FlowVisualizer);
colour wheel: tools/RAFT/core/utils/flow_viz.py:19-68 (make_colorwheel, Baker et al. / Middlebury, 55 entries).
Pinned against the reference's own class by tools/oracle/make_golden_flowvis.py -> tests/golden/flowvis_g7.npz.
"""
import math

import numpy as np
import torch


def make_colorwheel():
    """ref: flow_viz.py:19-68; segment lengths RY 15, YG 6, GC 4, CB 11, BM 13, MR 6."""
    segs = ((15, 0, 1, False), (6, 1, 0, True), (4, 1, 2, False), (11, 2, 1, True), (13, 2, 0, False), (6, 0, 2, True))
    wheel = np.zeros((sum(s[0] for s in segs), 3))
    col = 0
    for n, full, ramp, falling in segs:
        r = np.floor(255 * np.arange(n) / n)
        wheel[col:col + n, full] = 255
        wheel[col:col + n, ramp] = 255 - r if falling else r
        col += n
    return wheel


def flow_uv_to_colors(u, v, colorwheel):
    """ref: ssl_aug.py:87-120 with convert_to_bgr=False, div255=True.  u, v: (N,H,W) fp32 -> (N,H,W,3) fp32 in k/255."""
    out = torch.zeros((*u.shape, 3), dtype=torch.uint8)
    ncols = colorwheel.shape[0]
    rad = torch.sqrt(torch.square(u) + torch.square(v))
    a = torch.atan2(-v, -u) / math.pi
    fk = (a + 1) / 2 * (ncols - 1)
    k0 = torch.floor(fk).long()
    k1 = k0 + 1
    k1[k1 == ncols] = 0
    f = fk - k0
    for i in range(3):
        tmp = colorwheel[:, i]
        col0 = tmp[k0] / 255.0
        col1 = tmp[k1] / 255.0
        col = (1 - f) * col0 + f * col1
        idx = rad <= 1
        col[idx] = 1 - rad[idx] * (1 - col[idx])
        col[~idx] = col[~idx] * 0.75
        out[..., i] = torch.floor(255 * col)
    return out.float() / 255


class FlowVisualizer:
    """ref: ssl_aug.py:122-136: (B,2,T,H,W) uv -> (B,3,T,H,W) colours."""

    def __init__(self):
        self.colorwheel = torch.from_numpy(make_colorwheel())

    def __call__(self, flows):
        bs, _, t = flows.shape[:3]
        u, v = flows.chunk(2, dim=1)
        u, v = u[:, 0].flatten(0, 1), v[:, 0].flatten(0, 1)
        img = flow_uv_to_colors(u, v, self.colorwheel)
        return img.unflatten(0, (bs, t)).permute(0, 4, 1, 2, 3)
