"""grad-clip + SGD-momentum over the flat arena.  ref: mmcv OptimizerHook.after_train_iter wired at
mmaction/apis/train.py:111-119 with optimizer / optimizer_config of mscl_r18_cosm_lr2e-2.py:114-119:
clip_grad_norm_(params with grad, max_norm=40, L2) then torch.optim.SGD(lr, momentum 0.9, wd 1e-4).
Three kernels per step (sum of squares, fused clip+decay+momentum+update+bf16 shadow per active range,
kernel re-layout for the input-gradient GEMM) instead of ~600 tiny ones."""
import math

import torch

from . import kernels as K
from . import parallel


def cosine_lr(base_lr, epoch, max_epochs, min_lr=0.0):
    """mmcv CosineAnnealingLrUpdaterHook by epoch (lr_config at mscl_r18_cosm_lr2e-2.py:123; the config has
    no `warmup=` key, so mmcv applies no warm-up -- SURVEY.md §5)."""
    return min_lr + 0.5 * (base_lr - min_lr) * (1 + math.cos(math.pi * epoch / max_epochs))


class ClipSGD:
    def __init__(self, model, lr=0.02, momentum=0.9, weight_decay=1e-4, grad_clip=dict(max_norm=40, norm_type=2)):
        if grad_clip is not None and grad_clip.get('norm_type', 2) != 2:
            raise NotImplementedError('only the L2 norm is used (optimizer_config.grad_clip.norm_type=2)')
        self.model, self.arena = model, model.arena
        self.lr, self.momentum, self.wd = lr, momentum, weight_decay
        self.max_norm = float(grad_clip['max_norm']) if grad_clip else 0.0
        self.steps = 0
        self._ranges = None
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=self.arena.device)
        self.param_groups = [dict(lr=lr, momentum=momentum, weight_decay=weight_decay)]

    @classmethod
    def from_cfg(cls, model, optimizer, optimizer_config=None):
        o = dict(optimizer)
        if o.pop('type', 'SGD') != 'SGD':
            raise NotImplementedError('only SGD is configured for MSCL')
        clip = (optimizer_config or {}).get('grad_clip')
        return cls(model, grad_clip=clip, **o)

    def zero_grad(self):
        self.arena.G.zero_()

    def grad_norm(self):
        """device scalar: L2 norm of the (all-reduced) gradient before clipping"""
        return self._sumsq.sqrt()

    @torch.no_grad()
    def step(self):
        ar = self.arena
        self.model.flush_padded_grads()
        parallel.allreduce_mean_(ar.G)
        ranges = ar.active_ranges()
        if self._ranges is None:
            self._ranges = ranges
        elif ranges != self._ranges:
            raise RuntimeError('the set of parameters receiving gradients changed between steps')
        self._sumsq.zero_()
        K.sumsq(ar.G, self._sumsq)
        lr = self.param_groups[0]['lr']
        for a, b in ranges:
            K.sgd_step(ar.Q[a:b], ar.G[a:b], ar.MOM[a:b], ar.Qb[a:b], self._sumsq, self.max_norm, lr, self.momentum,
                       self.wd, first=(self.steps == 0))
        self.model.refresh_after_optimizer()
        self.steps += 1
