"""GPU-side input stage.  ref: mmaction/models/common/ssl_aug_v2.py:50-133 (SyncMoCoAugmentV5),
common/ssl_aug.py:178-183 (IdentityAug).

Implemented: the deterministic part -- ImageNet normalisation of the RGB views (ssl_aug_v2.py:66-68), the
uv -> colour-wheel FlowVisualizer with its uint8 floor (ssl_aug.py:87-136) for 2-channel flow clips, and the
horizontal flip GIVEN its per-sample Bernoulli mask (ssl_aug_v2.py:107-118: RGB clips and the visualised flow image
are mirrored along W) -- all fused into the NCTHW -> NDHWC packing kernels.  3-channel flow clips are taken as already
visualised and pass through un-normalised (normalize_flow=False -> Identity, ssl_aug_v2.py:88).  The stochastic kornia
ops (drawing the flip mask, colour jitter, grayscale, blur) are not implemented; asking for them raises.
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

    def pack_rgb(self, x, flip=None):
        return K.pack_input(x.contiguous(), IMAGENET_MEAN, IMAGENET_STD, flip=flip)

    def pack_flow(self, x, t_off, T, flip=None):
        if x.shape[1] == 2:                     # raw (u, v): FlowVisualizer fused into the packing pass
            if not self.visualize:
                raise NotImplementedError('visualize=False with 2-channel flow: the flow trunk takes 3 input channels')
            if self.normalize_flow:
                raise NotImplementedError('normalize_flow=True on raw uv flow (the shipped config uses False)')
            return K.flow_visualize(x.contiguous(), t_off=t_off, T=T, flip=flip)
        if self.normalize_flow:
            return K.pack_input(x.contiguous(), IMAGENET_MEAN, IMAGENET_STD, t_off=t_off, T=T, flip=flip)
        return K.pack_input(x.contiguous(), t_off=t_off, T=T, flip=flip)


@SSL_AUGS.register_module()
class IdentityAug:
    def __init__(self, **kwargs):
        pass

    def __call__(self, clips):
        return clips

    def pack_rgb(self, x, flip=None):
        return K.pack_input(x.contiguous(), flip=flip)

    def pack_flow(self, x, t_off, T, flip=None):
        if x.shape[1] == 2:
            return K.flow_visualize(x.contiguous(), t_off=t_off, T=T, flip=flip)
        return K.pack_input(x.contiguous(), t_off=t_off, T=T, flip=flip)
