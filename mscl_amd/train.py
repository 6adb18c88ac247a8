"""Training-driver pieces around the step: the epoch-wise cosine learning-rate schedule,
checkpoint save / resume including queue state and optimizer momentum, and a plain loop.

ref: mmaction/apis/train.py:111-238 (runner wiring), lr_config / total_epochs / checkpoint_config of
configs/recognition/moco/mscl_r18_cosm_lr2e-2.py:114-131.  mmcv's EpochBasedRunner is not vendored in the reference;
its behaviour for this config is: lr(epoch) = cosine annealing by epoch to min_lr 0 with no warm-up (the config
gives no `warmup=` key), one optimizer step per iteration, a checkpoint dict {'meta', 'state_dict', 'optimizer'}.
The reference does NOT checkpoint MoCoV2.iters (a plain Python attribute): on resume its momentum schedule restarts
from iters = 0.  `meta['mscl_amd']` carries the counters; resume(..., restore_counters=False) reproduces the reference.
"""
import torch

from .optim import cosine_lr


def save_checkpoint(path, model, optimizer, epoch=0, it=0):
    ar = model.arena
    ckpt = {
        'meta': {'epoch': int(epoch), 'iter': int(it),
                 'mscl_amd': {'rgb_iters': model.recognizer.iters, 'flow_iters': model.recognizer_flow.iters,
                              'step': model._step, 'optimizer_steps': optimizer.steps}},
        'state_dict': {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
        'optimizer': {'momentum_buffer_flat': ar.MOM.detach().cpu().clone(), 'lr': optimizer.param_groups[0]['lr'],
                      'momentum': optimizer.momentum, 'weight_decay': optimizer.wd},
    }
    torch.save(ckpt, path)
    return ckpt['meta']


def resume(path, model, optimizer, restore_counters=True):
    """load a checkpoint written by save_checkpoint (or a reference checkpoint: `state_dict` with the reference's 551
    keys; its optimizer state, stored per parameter by torch.optim.SGD, is not mapped)"""
    ckpt = torch.load(path, map_location='cpu', weights_only=False)
    model.load_state_dict(ckpt['state_dict'])
    model.sync_shadows()
    o = ckpt.get('optimizer') or {}
    if 'momentum_buffer_flat' in o:
        model.arena.MOM.copy_(o['momentum_buffer_flat'].to(model.arena.device))
        optimizer.param_groups[0]['lr'] = o['lr']
    own = ckpt.get('meta', {}).get('mscl_amd')
    if own and restore_counters:
        model.recognizer.iters, model.recognizer_flow.iters = own['rgb_iters'], own['flow_iters']
        model._step, optimizer.steps = own['step'], own['optimizer_steps']
    return ckpt.get('meta', {})


def train(model, optimizer, batches, total_epochs, base_lr=None, min_lr=0.0, start_epoch=0, log=None, step_fn=None):
    """`batches`: callable epoch -> iterable of data_batch dicts (device tensors).  Returns the last log_vars.
    step_fn(data_batch) -> (loss, log_vars) may be a mscl_amd.graph.GraphedStep(...).step for graph replay."""
    base_lr = optimizer.param_groups[0]['lr'] if base_lr is None else base_lr
    last = None
    for epoch in range(start_epoch, total_epochs):
        optimizer.param_groups[0]['lr'] = cosine_lr(base_lr, epoch, total_epochs, min_lr)
        for it, data_batch in enumerate(batches(epoch)):
            if step_fn is not None:
                optimizer.sync_lr()
                loss, last = step_fn(data_batch)
            else:
                out = model.train_step(data_batch, optimizer, sync_logs=False)
                optimizer.zero_grad()
                out['loss'].backward()
                optimizer.step()
                last = out['log_vars']
            if log is not None:
                log(epoch, it, last)
    return last
