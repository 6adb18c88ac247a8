"""Seeded synthetic clip batches.

One batch = {'imgs': [q, k], 'flow_imgs': [q, k]}; RGB views are (B,3,T,H,W) in U[0,1); flow views
are (B,3,2T,H,W) in U[0,1): the already-visualised base flow followed by the rotated flow along T
(reference layout: recognizers/mscl.py:230-235 chunks dim 2).
"""
import torch


def batch_seed(rank, step):
    return 1234 + 1000 * rank + step


def synthetic_batch(B=8, T=16, H=112, W=112, rank=0, step=0, device='cpu'):
    g = torch.Generator().manual_seed(batch_seed(rank, step))
    mk = lambda t: torch.rand(B, 3, t, H, W, generator=g)
    batch = {'imgs': [mk(T), mk(T)], 'flow_imgs': [mk(2 * T), mk(2 * T)]}
    if device != 'cpu':
        batch = {k: [t.to(device, non_blocking=True) for t in v] for k, v in batch.items()}
    return batch
