"""G12: what the reference's stochastic augmentation does OUTSIDE kornia, run here.

kornia (absent, unpinned) owns the colour arithmetic and its own parameter draws; the reference's own code around it is
plain torch / Python and is pinned by this fixture:
  * `__video_batch_prob_generator__` / `_adapted_sampling_video` (common/ssl_aug.py:21-53): the per-CLIP on/off decision
    of every random op -- Bernoulli(p).sample((clips, 1)) from torch's global generator, repeated over the t frames;
  * `VideoRandomApply.forward` (ssl_aug.py:138-153): applies its transform to exactly the frames of the chosen clips;
  * `GaussianBlur` (ssl_aug.py:163-171): kernel size int(0.1 * img_size) // 2 * 2 + 1 and ONE sigma per call from Python's
    random.uniform(0.1, 2.0) -- captured from the arguments it hands to kornia.filters.GaussianBlur2d (a MagicMock here).
Writes tests/golden/augdraws_g12.json.  Development container only."""
import importlib
import json
import os
import random
import sys
from unittest.mock import MagicMock

import torch
from torch.distributions import Bernoulli

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness as rh                              # noqa: E402


def main():
    rh.install()
    for name in ('kornia', 'kornia.augmentation', 'kornia.augmentation.utils', 'kornia.filters', 'torchvision.transforms',
                 'torchvision.datasets', 'torchvision.datasets.video_utils'):
        sys.modules.setdefault(name, MagicMock())
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    sys.modules.setdefault('mmaction.models.common.motion_map_calculator', MagicMock())
    mod = importlib.import_module('mmaction.models.common.ssl_aug')
    gen = getattr(mod, '__video_batch_prob_generator__')
    out = dict(decisions=[], random_apply=[], blur=[])
    for seed in (0, 1, 2):
        for p in (0.8, 0.2, 0.5):                             # ColorJitter, RandomGrayscale, blur (ssl_aug_v2.py:37-39)
            for clips, t in ((8, 8), (32, 16), (3, 4)):
                holder = MagicMock()
                holder._p_gen, holder._p_batch_gen = Bernoulli(p), Bernoulli(1)
                torch.manual_seed(seed)
                mask = gen(torch.Size((clips * t,)), p, 1, False, self=holder, t=t)
                assert mask.shape[0] == clips * t and bool((mask.view(clips, t) == mask.view(clips, t)[:, :1]).all())
                out['decisions'].append(dict(seed=seed, p=p, clips=clips, t=t, per_clip=mask.view(clips, t)[:, 0].int().tolist()))
        # VideoRandomApply: frames of the chosen clips, and only those, go through the transform
        t, clips = 4, 6
        vra = mod.VideoRandomApply(lambda x: x + 100.0, t, p=0.5)
        img = torch.arange(clips * t, dtype=torch.float32).view(clips * t, 1, 1, 1).repeat(1, 3, 2, 2)
        torch.manual_seed(seed)
        res = vra(img.clone())
        out['random_apply'].append(dict(seed=seed, clips=clips, t=t, changed=(res[:, 0, 0, 0] != img[:, 0, 0, 0]).int().tolist()))
        # GaussianBlur: what it asks kornia for
        for size in (112, 224, 128):
            blur = mod.GaussianBlur([0.1, 2.0], size)
            mod.kornia.filters.GaussianBlur2d.reset_mock()
            random.seed(seed)
            sig = []
            for _ in range(3):
                blur(torch.zeros(1, 3, 8, 8))
                (ks, sg), _kw = mod.kornia.filters.GaussianBlur2d.call_args
                assert ks == (blur.radius, blur.radius) and sg[0] == sg[1]
                sig.append(sg[0])
            out['blur'].append(dict(seed=seed, img_size=size, ksize=blur.radius, sigmas=sig))
    path = os.path.join(ROOT, 'tests', 'golden', 'augdraws_g12.json')
    json.dump(out, open(path, 'w'))
    print('wrote', path, len(out['decisions']), 'decision streams')


if __name__ == '__main__':
    main()
