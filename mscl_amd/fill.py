"""Closed-form, RNG-free fill of every parameter/buffer (product copy, used by bench.py and smoke; oracle/fill.py is the test-side twin).

value(name, i) = amp(name, shape) * u(name, i) [+ 1 for norm scales], where u is a unit-variance
uniform deviate obtained from a 32-bit integer mix (murmur3 finaliser) of (crc32(name) + i).
A counter hash instead of a sine wave: smooth sinusoidal kernels make a degenerate band-pass
network whose BatchNorm layers amplify gradients to ~1e5, useless as a parity test bed.

Both sides of a parity test (reference model via the harness, oracle model, HIP model) are filled
by *state-dict name*, so no 300 MB weight file has to travel.  Key-encoder tensors get the value of
their query twin (the reference copies q -> k at construction, moco.py:379-387).
"""
import math
import zlib

import numpy as np
import torch



def canonical(name):
    return name.replace('encoder_k', 'encoder_q').replace('neck_k', 'neck_q').replace('mlp_k', 'mlp_q')


def mix32(x):
    """murmur3 fmix32 on a uint64 array holding 32-bit values."""
    m = np.uint64(0xFFFFFFFF)
    x = x & m
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & m
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & m
    x ^= x >> np.uint64(16)
    return x


def wave(name, numel):
    """unit-variance uniform deviates in [-sqrt(3), sqrt(3)), float64."""
    seed = np.uint64(zlib.crc32(canonical(name).encode()))
    x = mix32(np.arange(numel, dtype=np.uint64) * np.uint64(0x9E3779B1) + seed)
    u = x.astype(np.float64) / 4294967296.0
    return (2.0 * u - 1.0) * math.sqrt(3.0)


def fill_value(name, shape):
    """float64 ndarray of `shape` for tensor `name`, or None if the tensor keeps its default."""
    n = int(np.prod(shape)) if len(shape) else 1
    leaf = name.rsplit('.', 1)[-1]
    if leaf in ('running_mean', 'running_var', 'num_batches_tracked', 'queue_ptr', 'count', 'labels'):
        return None
    if leaf == 'queue':                      # column-normalised (moco.py:390-391)
        q = wave(name, n).reshape(shape)
        return q / np.maximum(np.sqrt((q * q).sum(0, keepdims=True)), 1e-12)
    w = wave(name, n).reshape(shape)
    if len(shape) == 5:                      # conv kernels: keep the init's variance
        rf = shape[2] * shape[3] * shape[4]
        fan_in, fan_out = shape[1] * rf, shape[0] * rf
        std = math.sqrt(2.0 / (fan_in + fan_out)) if 'neck' in name else math.sqrt(2.0 / fan_out)
        return w * std
    if len(shape) in (2, 3):                 # linear / Conv1d with a 1-wide kernel
        return w / math.sqrt(3.0 * shape[1])
    if len(shape) == 1:
        if leaf == 'weight':                 # norm scale
            return 1.0 + 0.1 * w
        return 0.05 * w                      # norm shift / conv bias / linear bias
    raise ValueError(f'no fill rule for {name} {tuple(shape)}')


@torch.no_grad()
def fill_module(module):
    """Overwrite every state-dict entry of `module` in place; returns the module."""
    for name, t in module.state_dict().items():
        v = fill_value(name, tuple(t.shape))
        if v is not None:
            t.copy_(torch.from_numpy(np.asarray(v)).to(t.dtype))
    return module
