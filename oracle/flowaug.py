"""CPU restatement of the reference's Flow Rotation Augmentation (TEST INFRASTRUCTURE, see oracle/__init__.py).

ref: mmaction/datasets/pipelines/transforms_motion.py:7-29 (norm_flow), :103-142 (NormFlowWithStidedAug), as configured at
configs/recognition/moco/mscl_r18_cosm_lr2e-2.py:70,82 (ratios (0.2, 1.8), num_chunks 8, merge_aug=True).
Pinned against the reference class by tools/oracle/make_golden_fra.py -> tests/golden/fra_g8.npz.
"""
import numpy as np


def norm_flow(flow_uv):
    """ref: transforms_motion.py:7-29 with clip_flow=None: divide by (max radius of the frame + 1e-5)."""
    u, v = flow_uv[:, :, 0], flow_uv[:, :, 1]
    rad_max = np.max(np.sqrt(np.square(u) + np.square(v)))
    return np.stack((u / (rad_max + 1e-5), v / (rad_max + 1e-5)), axis=-1)


def fra(flows, cid, ratios=(0.2, 1.8), num_chunks=8):
    """ref: transforms_motion.py:118-141 for a given chunk id: list of (H,W,2) -> base frames + rotated frames."""
    start, stride = ratios[0], (ratios[1] - ratios[0]) / num_chunks
    beta = (start + stride * cid) * np.pi
    s, c = np.sin(beta), np.cos(beta)
    base, rot = [], []
    for f in flows:
        u, v = f[:, :, 0], f[:, :, 1]
        base.append(norm_flow(f))
        rot.append(norm_flow(np.stack((c * u - s * v, s * u + c * v), axis=-1)))
    return base + rot
