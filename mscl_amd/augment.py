"""GPU-side input stage.  ref: mmaction/models/common/ssl_aug_v2.py:50-133 (SyncMoCoAugmentV5),
common/ssl_aug.py:178-183 (IdentityAug).

Scope this round (SURVEY.md §8f#1 is a "next" row): the deterministic part that the measured path
needs -- ImageNet normalisation of the RGB views (ssl_aug_v2.py:66-68), fused into the NCTHW->NDHWC
packing kernel; flow views arrive already visualised (3 channels) and pass through un-normalised
(normalize_flow=False -> Identity, ssl_aug_v2.py:88).  The stochastic kornia ops (flip, colour jitter,
grayscale, blur) and the uv->colour-wheel visualiser are not implemented; asking for them raises.
"""
from . import kernels as K
from .registry import SSL_AUGS

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


@SSL_AUGS.register_module()
class SyncMoCoAugmentV5:
    def __init__(self, crop_size, flip_transform=dict(p=0.5, same_on_batch=False), sync_level='batch', t=None,
                 flow_suffix='flow_imgs', img_width=112, visualize=True, weak_aug=(False, False), normalize_flow=False,
                 stochastic=False):
        if stochastic:
            raise NotImplementedError('stochastic flip/jitter/grayscale/blur (kornia in the reference) is a "next" row')
        self.crop_size, self.t, self.flow_suffix = crop_size, t, flow_suffix
        self.visualize, self.normalize_flow = visualize, normalize_flow

    def pack_rgb(self, x):
        return K.pack_input(x.contiguous(), IMAGENET_MEAN, IMAGENET_STD)

    def pack_flow(self, x, t_off, T):
        if x.shape[1] != 3:
            raise NotImplementedError('2-channel uv flow needs the colour-wheel visualiser (ssl_aug.py:87-136), a "next" row; '
                                      'feed visualised 3-channel flow clips')
        if self.normalize_flow:
            return K.pack_input(x.contiguous(), IMAGENET_MEAN, IMAGENET_STD, t_off=t_off, T=T)
        return K.pack_input(x.contiguous(), t_off=t_off, T=T)


@SSL_AUGS.register_module()
class IdentityAug:
    def __init__(self, **kwargs):
        pass

    def __call__(self, clips):
        return clips

    def pack_rgb(self, x):
        return K.pack_input(x.contiguous())

    def pack_flow(self, x, t_off, T):
        return K.pack_input(x.contiguous(), t_off=t_off, T=T)
