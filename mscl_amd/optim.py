"""grad-clip + SGD-momentum over the flat arena.  ref: mmcv OptimizerHook.after_train_iter wired at
mmaction/apis/train.py:111-119 with optimizer / optimizer_config of mscl_r18_cosm_lr2e-2.py:114-119:
clip_grad_norm_(params with grad, max_norm=40, L2) then torch.optim.SGD(lr, momentum 0.9, wd 1e-4).
Three kernels per step (sum of squares, fused clip+decay+momentum+update+bf16 shadow per active range,
kernel re-layout for the input-gradient GEMM) instead of ~600 tiny ones."""
import math

import torch

from . import kernels as K
from . import parallel
from .staging import StagingRing


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
        # lr is read from a device word, so a captured graph sees schedule changes; the word is fed through a ring of
        # pinned slots (staging.py) so that the host, which runs steps ahead of the GPU, never rewrites a word in flight
        self._lr = StagingRing((1,), torch.float32, self.arena.device)
        self._lr_cpu = torch.full((1,), float(lr), dtype=torch.float32)
        self._lr.push(self._lr_cpu)

    @classmethod
    def from_cfg(cls, model, optimizer, optimizer_config=None):
        o = dict(optimizer)
        if o.pop('type', 'SGD') != 'SGD':
            raise NotImplementedError('only SGD is configured for MSCL')
        clip = (optimizer_config or {}).get('grad_clip')
        return cls(model, grad_clip=clip, **o)

    def zero_grad(self):
        self.model.sync_streams()
        self.arena.G.zero_()

    def grad_norm(self):
        """device scalar: L2 norm of the (all-reduced) gradient before clipping"""
        return self._sumsq.sqrt()

    @torch.no_grad()
    def step(self):
        ar = self.arena
        self.model.sync_streams()
        if getattr(self.model, 'reducer', None) is not None:
            self.model.reducer.finish()         # buckets were launched from backward; wait for them
        else:
            parallel.allreduce_mean_(ar.G)
        ranges = ar.active_ranges()
        if self._ranges is None:
            self._ranges = ranges
        elif ranges != self._ranges:
            raise RuntimeError('the set of parameters receiving gradients changed between steps')
        self._sumsq.zero_()
        K.sumsq(ar.G, self._sumsq)
        if not torch.cuda.is_current_stream_capturing():        # a captured step publishes before each replay (graph.py)
            self.sync_lr()
        for a, b in ranges:     # momentum buffers start at zero: buf = mom*0 + d == torch's first-step buf = d
            K.sgd_step_dev(ar.Q[a:b], ar.G[a:b], ar.MOM[a:b], ar.Qb[a:b], self._sumsq, self.max_norm, self._lr.dev,
                           self.momentum, self.wd)
        self.model.refresh_after_optimizer()
        self.steps += 1

    def sync_lr(self):
        """publish param_groups[0]['lr'] to the device word the SGD kernel reads, in stream order (eager steps: from
        step(); captured steps: before each graph replay).  Unchanged values are not re-sent."""
        lr = float(self.param_groups[0]['lr'])
        if self._lr.pushes and float(self._lr_cpu[0]) == torch.tensor(lr, dtype=torch.float32).item():
            return
        self._lr_cpu[0] = lr
        self._lr.push(self._lr_cpu)

    @property
    def _lr_dev(self):
        return self._lr.dev
