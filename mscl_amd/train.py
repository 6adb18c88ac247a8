"""Training-driver pieces around the step: the epoch-wise cosine learning-rate schedule,
checkpoint save / resume including queue state and optimizer momentum, the evaluation pass of the reference's
`SimpleDistEvalHook`, the text logger's cadence, and a plain loop that wires them from the config.

ref: mmaction/apis/train.py:111-238 (runner wiring), lr_config / total_epochs / checkpoint_config of
configs/recognition/moco/mscl_r18_cosm_lr2e-2.py:114-131, configs/_base_/default_runtime.py:1-13,
mmaction/core/evaluation/eval_hooks.py:417-529.  mmcv's EpochBasedRunner is not vendored in the reference;
its behaviour for this config is: lr(epoch) = cosine annealing by epoch to min_lr 0 with no warm-up (the config
gives no `warmup=` key), one optimizer step per iteration, a checkpoint dict {'meta', 'state_dict', 'optimizer'} every
`checkpoint_config.interval` epochs, one log record every `log_config.interval` iterations holding the MEAN of each
log variable over the iterations since the last record (mmcv LogBuffer.average).
The reference does NOT checkpoint MoCoV2.iters (a plain Python attribute): on resume its momentum schedule restarts
from iters = 0.  `meta['mscl_amd']` carries the counters; resume(..., restore_counters=False) reproduces the reference.
"""
import os
import warnings
from collections import OrderedDict

import torch

from . import parallel
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


def _load_reference_sgd_state(model, optimizer, state):
    """torch.optim.SGD.state_dict() of the reference run: per-parameter `momentum_buffer`s keyed by the parameter's index
    in model.parameters() (mmcv's DefaultOptimizerConstructor hands the optimizer every parameter, key encoders included;
    those never receive a gradient and hold no state).  Copied into the momentum arena by parameter order.
    Returns the number of buffers mapped."""
    per = state.get('state', {})
    n = 0
    for idx, p in enumerate(model.parameters()):
        st = per.get(idx, per.get(str(idx)))
        if st is None or st.get('momentum_buffer') is None:
            continue
        slot = getattr(p, '_mscl_slot', None)
        if slot is None or not p.requires_grad or tuple(st['momentum_buffer'].shape) != tuple(p.shape):
            raise ValueError(f'optimizer state {idx} does not belong to trainable parameter {idx} of this model')
        model.arena.view('MOM', slot).copy_(st['momentum_buffer'].to(model.arena.device))
        slot.touched = True            # SGD applies weight decay / momentum to it from now on, like the reference's state does
        n += 1
    groups = state.get('param_groups') or [{}]
    if 'lr' in groups[0]:
        optimizer.param_groups[0]['lr'] = groups[0]['lr']
    return n


def resume(path, model, optimizer, restore_counters=True):
    """load a checkpoint written by save_checkpoint, or a reference checkpoint (`state_dict` with the reference's 551
    keys, `optimizer` = torch.optim.SGD's state dict, mapped by parameter order).  The file is read with
    weights_only=True: tensors, dicts, lists and scalars only -- nothing in it is executed."""
    ckpt = torch.load(path, map_location='cpu', weights_only=True)
    model.load_state_dict(ckpt['state_dict'])
    model.sync_shadows()
    o = ckpt.get('optimizer') or {}
    if 'momentum_buffer_flat' in o:
        model.arena.MOM.copy_(o['momentum_buffer_flat'].to(model.arena.device))
        optimizer.param_groups[0]['lr'] = o['lr']
    elif 'state' in o:
        n = _load_reference_sgd_state(model, optimizer, o)
        if n == 0:
            warnings.warn('checkpoint holds an optimizer entry without momentum buffers: SGD momentum restarts at zero')
    else:
        warnings.warn('checkpoint holds no optimizer state: SGD momentum restarts at zero')
    own = ckpt.get('meta', {}).get('mscl_amd')
    if own and restore_counters:
        model.recognizer.iters, model.recognizer_flow.iters = own['rgb_iters'], own['flow_iters']
        model._step, optimizer.steps = own['step'], own['optimizer_steps']
    return ckpt.get('meta', {})


class LogBuffer:
    """mmcv LogBuffer + TextLoggerHook cadence: every `interval` iterations emit the mean of each variable over the
    iterations since the last record (log_config.interval = 20, default_runtime.py:3).  Values may be 0-d device tensors
    (sync_logs=False): they are stacked and read back once per record, not once per step."""

    def __init__(self, interval=20):
        self.interval, self.rows, self.keys = int(interval), [], None

    def update(self, log_vars):
        self.keys = list(log_vars.keys())
        vals = list(log_vars.values())
        self.rows.append(torch.stack([v.detach().float() for v in vals]) if torch.is_tensor(vals[0])
                         else torch.tensor(vals, dtype=torch.float32))

    def ready(self, it):
        return (it + 1) % self.interval == 0

    def average(self):
        if not self.rows:
            return OrderedDict()
        mean = torch.stack(self.rows).mean(0).tolist()
        self.rows = []
        return OrderedDict(zip(self.keys, mean))


@torch.no_grad()
def evaluate(model, batches):
    """The reference's evaluation pass for the self-supervised recognizers (eval_hooks.py:471-487 `multi_gpu_test`):
    `model.eval()`, `train_step(data, optimizer=None)` under no_grad for every batch, and per variable the average
    weighted by `num_samples` (AverageMeter, eval_hooks.py:399-415).  As in the reference the pass is NOT side-effect
    free: key encoders take their EMA update and the queues are enqueued (moco.py:535-545,496-499 run regardless of the
    mode); only `iters` stands still (`if self.training`, moco.py:504) and BatchNorm uses and keeps its running statistics.
    Returns OrderedDict name -> average; the model is put back into its previous mode."""
    was_training = model.training
    model.eval()
    sums, count = None, 0
    try:
        for data in batches:
            out = model.train_step(data, optimizer=None, sync_logs=False)
            n = int(out['num_samples'])
            row = torch.stack([v.detach().float() for v in out['log_vars'].values()]) * n
            sums = row if sums is None else sums + row
            keys = list(out['log_vars'].keys())
            count += n
    finally:
        model.train(was_training)
    if sums is None:
        return OrderedDict()
    return OrderedDict(zip(keys, (sums / count).tolist()))


def broadcast_bn_buffers(model):
    """eval_hooks.py:489-500: rank 0's BatchNorm running statistics to every rank before an evaluation (per-GPU
    statistics drift apart without SyncBN)."""
    if parallel.single():
        return
    import torch.distributed as dist
    from .nn import BatchNorm3dHip
    for m in model.modules():
        if isinstance(m, BatchNorm3dHip):
            dist.broadcast(m.running_var, 0)
            dist.broadcast(m.running_mean, 0)


def set_random_seed(seed, deterministic=False):
    """ref: apis/train.py `set_random_seed` as tools/train.py:145-150 calls it: seeds Python / NumPy / torch and, with
    `--deterministic`, asks for reproducible kernels -- here the library's deterministic mode (fixed-order sums in place of float
    atomics, lib.set_deterministic) in place of cudnn.deterministic."""
    import random

    import numpy as np
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if deterministic:
        from . import lib
        lib.set_deterministic(True)


def train(model, optimizer, batches, total_epochs, base_lr=None, min_lr=0.0, start_epoch=0, log=None, step_fn=None,
          log_interval=None, work_dir=None, checkpoint_interval=None, val_batches=None, eval_interval=1, cfg=None, start_iter=0):
    """`batches`: callable epoch -> iterable of data_batch dicts (device tensors).  Returns the last log_vars.
    step_fn(data_batch) -> (loss, log_vars) may be a mscl_amd.graph.GraphedStep(...).step for graph replay.
    cfg (a loaded config): fills log_interval / checkpoint_interval / min_lr from `log_config`, `checkpoint_config` and
    `lr_config` unless given.  log(epoch, it, log_vars): called per record when log_interval is set (mean over the
    interval, mmcv's cadence), else per iteration with that iteration's values.  work_dir: `epoch_{n}.pth` every
    checkpoint_interval epochs (mmcv CheckpointHook naming) and `latest.pth`.  val_batches: callable epoch -> iterable;
    evaluated every eval_interval epochs (EvalHook `interval`, by_epoch), results passed to log(epoch, 'val', res).
    start_iter: the global iteration count so far (mmcv's runner.iter; pass `resume(...)['iter']` after a resume) -- checkpoints
    store the running count, whatever the epoch lengths were.  Records of fewer than log_interval iterations at the end of an
    epoch are dropped, as mmcv's LoggerHook does with its default ignore_last=True."""
    if cfg is not None:
        if log_interval is None and log is not None:
            log_interval = (cfg.get('log_config') or {}).get('interval')
        if checkpoint_interval is None:
            checkpoint_interval = (cfg.get('checkpoint_config') or {}).get('interval')
        min_lr = (cfg.get('lr_config') or {}).get('min_lr', min_lr)
    base_lr = optimizer.param_groups[0]['lr'] if base_lr is None else base_lr
    buf = LogBuffer(log_interval) if (log is not None and log_interval) else None
    last = None
    global_iter = int(start_iter)
    for epoch in range(start_epoch, total_epochs):
        optimizer.param_groups[0]['lr'] = cosine_lr(base_lr, epoch, total_epochs, min_lr)
        it = -1
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
            global_iter += 1
            if buf is not None:
                buf.update(last)
                if buf.ready(it):
                    log(epoch, it, buf.average())
            elif log is not None:
                log(epoch, it, last)
        if buf is not None:
            buf.rows = []                       # mmcv clears the buffer at the epoch boundary
        if work_dir is not None and checkpoint_interval and (epoch + 1) % checkpoint_interval == 0 and parallel.rank() == 0:
            os.makedirs(work_dir, exist_ok=True)
            path = os.path.join(work_dir, f'epoch_{epoch + 1}.pth')
            save_checkpoint(path, model, optimizer, epoch=epoch + 1, it=global_iter)
            latest = os.path.join(work_dir, 'latest.pth')
            if os.path.lexists(latest):
                os.remove(latest)
            os.symlink(os.path.basename(path), latest)
        if val_batches is not None and (epoch + 1) % eval_interval == 0:
            broadcast_bn_buffers(model)
            res = evaluate(model, val_batches(epoch))
            if log is not None:
                log(epoch, 'val', res)
    return last
