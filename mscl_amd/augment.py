"""GPU-side input stage.  ref: mmaction/models/common/ssl_aug_v2.py:50-133 (SyncMoCoAugmentV5),
common/ssl_aug.py:178-183 (IdentityAug).

Implemented: the deterministic part -- ImageNet normalisation of the RGB views (ssl_aug_v2.py:66-68), the
uv -> colour-wheel FlowVisualizer with its uint8 floor (ssl_aug.py:87-136) for 2-channel flow clips, and the
horizontal flip GIVEN its per-sample Bernoulli mask (ssl_aug_v2.py:107-118: RGB clips and the visualised flow image
are mirrored along W) -- all fused into the NCTHW -> NDHWC packing kernels.  3-channel flow clips are taken as already
visualised and pass through un-normalised (normalize_flow=False -> Identity, ssl_aug_v2.py:88).

stochastic=True adds the random part of the pipeline (ssl_aug_v2.py:31-43, 84, 107-110): per step and view the host
draws, from the module's own seeded generator, a flip mask (p 0.5 per clip), ColorJitter(0.4, 0.4, 0.4, 0.1) with p 0.8,
grayscale with p 0.2 and an 11-tap Gaussian blur with p 0.5 (one sigma in [0.1, 2] per call, ssl_aug.py:163-171); the
decisions are per CLIP (the reference's toVideoAug makes them consistent over time, ssl_aug.py:31-53, 66-70) and so
are the jitter factors; the four jitter ops run in one random order per call.  The arithmetic runs in the HIP kernels of
csrc/color_aug.hip from (B,16) parameter rows.  kornia's own random streams cannot be reproduced (it is not available
and unpinned), so parity for this part is on the arithmetic given the parameters, not on kornia's draws.  What the
reference draws in its OWN code is reproduced and pinned (tests/golden/augdraws_g12.json, from the reference's functions):
the per-clip on/off decision of each op is `Bernoulli(p).sample((clips, 1))` on a torch generator (ssl_aug.py:21-53,
`clip_decisions`), the blur takes kernel size int(0.1 * crop) // 2 * 2 + 1 and ONE sigma per call from Python's
random.uniform(0.1, 2.0) (ssl_aug.py:163-171, `blur_sigma`).
"""
import math
import random

import torch

from . import kernels as K
from .registry import SSL_AUGS

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


@SSL_AUGS.register_module()
class SyncMoCoAugmentV5:
    def __init__(self, crop_size, flip_transform=dict(p=0.5, same_on_batch=False), sync_level='batch', t=None,
                 flow_suffix='flow_imgs', img_width=112, visualize=True, weak_aug=(False, False), normalize_flow=False,
                 stochastic=False, seed=0):
        if isinstance(sync_level, str):
            sync_level = (sync_level, sync_level)
        assert all(v in ('batch', 'params') for v in sync_level)
        self.crop_size, self.t, self.flow_suffix = crop_size, t, flow_suffix
        self.visualize, self.normalize_flow = visualize, normalize_flow
        self.stochastic = bool(stochastic)
        self.flip_p = float(flip_transform['p']) if flip_transform else 0.0
        self.weak_aug = tuple(weak_aug)
        self.blur_ksize = int(0.1 * crop_size) // 2 * 2 + 1          # ssl_aug.py:166
        self._gen = torch.Generator().manual_seed(seed)
        self._pygen = random.Random(seed)

    def seed(self, seed):
        self._gen.manual_seed(seed)
        self._pygen.seed(seed)

    def clip_decisions(self, B, p):
        """ssl_aug.py:21-53 (`__video_batch_prob_generator__` with p_batch = 1): one Bernoulli(p) draw per clip, the same for
        all its frames -- the stream of torch.distributions.Bernoulli(p).sample((B, 1)) on this module's generator"""
        if p >= 1:
            return torch.ones(B, dtype=torch.bool)
        if p <= 0:
            return torch.zeros(B, dtype=torch.bool)
        return torch.bernoulli(torch.full((B, 1), float(p)), generator=self._gen).bool().view(-1)

    def blur_sigma(self):
        """ssl_aug.py:168: one sigma per call (i.e. per view and step), from Python's generator"""
        return self._pygen.uniform(0.1, 2.0)

    def _uniform(self, n, lo, hi):
        return lo + (hi - lo) * torch.rand(n, generator=self._gen)

    def draw(self, B):
        """one step's random decisions for the query and key views, as host tensors:
        dict(flip_mask=[(B,) uint8 x2], aug_params=[(B,16) fp32 x2]) -- the keys train_step() looks for."""
        flips, rows = [], []
        for view in range(2):
            flips.append(self.clip_decisions(B, self.flip_p).to(torch.uint8))
            P = torch.zeros(B, K.AUG_PARAMS)
            if not self.weak_aug[view]:
                P[:, 0] = self.clip_decisions(B, 0.8).float()
                P[:, 1:5] = torch.randperm(4, generator=self._gen).float()
                P[:, 5] = self._uniform(B, 0.6, 1.4)
                P[:, 6] = self._uniform(B, 0.6, 1.4)
                P[:, 7] = self._uniform(B, 0.6, 1.4)
                P[:, 8] = self._uniform(B, -0.1, 0.1) * (2.0 * math.pi)
                P[:, 9] = self.clip_decisions(B, 0.2).float()
                sigma = self.blur_sigma()
                P[:, 10] = self.clip_decisions(B, 0.5).float() * sigma
            rows.append(P)
        return dict(flip_mask=flips, aug_params=rows)

    def color(self, x, params, view):
        """jitter / grayscale / blur of an RGB view given its parameter rows (None: nothing to do)"""
        if params is None or self.weak_aug[view]:
            return x
        return K.color_aug(x.contiguous(), params, self.blur_ksize)

    def pack_rgb(self, x, flip=None, out=None):
        return K.pack_input(x.contiguous(), IMAGENET_MEAN, IMAGENET_STD, flip=flip, out=out)

    def pack_flow(self, x, t_off, T, flip=None, out=None):
        if x.shape[1] == 2:                     # raw (u, v): FlowVisualizer fused into the packing pass
            if not self.visualize:
                raise NotImplementedError('visualize=False with 2-channel flow: the flow trunk takes 3 input channels')
            if self.normalize_flow:
                raise NotImplementedError('normalize_flow=True on raw uv flow (the shipped config uses False)')
            return K.flow_visualize(x.contiguous(), t_off=t_off, T=T, flip=flip, out=out)
        if self.normalize_flow:
            return K.pack_input(x.contiguous(), IMAGENET_MEAN, IMAGENET_STD, t_off=t_off, T=T, flip=flip, out=out)
        return K.pack_input(x.contiguous(), t_off=t_off, T=T, flip=flip, out=out)


@SSL_AUGS.register_module()
class IdentityAug:
    stochastic = False

    def __init__(self, **kwargs):
        pass

    def color(self, x, params, view):
        return x

    def __call__(self, clips):
        return clips

    def pack_rgb(self, x, flip=None, out=None):
        return K.pack_input(x.contiguous(), flip=flip, out=out)

    def pack_flow(self, x, t_off, T, flip=None, out=None):
        if x.shape[1] == 2:
            return K.flow_visualize(x.contiguous(), t_off=t_off, T=T, flip=flip, out=out)
        return K.pack_input(x.contiguous(), t_off=t_off, T=T, flip=flip, out=out)
