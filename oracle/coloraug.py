"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the colour augmentation the reference chains in
mmaction/models/common/ssl_aug_v2.py:31-43 (ColorJitter(0.4, 0.4, 0.4, 0.1), RandomGrayscale, GaussianBlur of
common/ssl_aug.py:163-171), applied GIVEN the sampled parameters.  Never imported by the product path.

PARITY UNPINNED: the arithmetic lives in kornia (unpinned in the reference's requirements, absent from
/root/reference and from this image), and the reference holds no test or golden vector for it.  The functions below
restate kornia's published enhance/colour/filter operations (0.5-series semantics: additive brightness, multiplicative
contrast, saturation and hue through HSV with h in [0, 2 pi), first-index argmax for the hue sector, ITU-R 601 luma,
normalised Gaussian taps, reflect border) as whole-tensor torch ops; the HIP kernels are checked against THIS file.
"""
import math

import torch
import torch.nn.functional as F

TWO_PI = 2.0 * math.pi


def rgb_to_hsv(img, eps=1e-6):
    """img (..., 3, H, W) in [0,1] -> h in [0, 2 pi), s, v"""
    maxc, _ = img.max(-3)
    mask = img == maxc.unsqueeze(-3)
    _, idx = ((mask.cumsum(-3) == 1) & mask).max(-3)            # first channel that attains the maximum
    minc = img.min(-3)[0]
    v = maxc
    delta = maxc - minc
    s = delta / (v + eps)
    delta = torch.where(delta == 0, torch.ones_like(delta), delta)
    rc, gc, bc = (maxc.unsqueeze(-3) - img).unbind(-3)
    h = torch.stack([bc - gc, 2.0 * delta + rc - bc, 4.0 * delta + gc - rc], dim=-3)
    h = torch.gather(h, -3, idx.unsqueeze(-3)).squeeze(-3)
    h = (h / delta / 6.0) % 1.0
    return TWO_PI * h, s, v


def hsv_to_rgb(h, s, v):
    h = h / TWO_PI
    hi = torch.floor(h * 6) % 6
    f = ((h * 6) % 6) - hi
    p = v * (1 - s)
    q = v * (1 - f * s)
    t = v * (1 - (1 - f) * s)
    hi = hi.long()
    table = torch.stack((v, q, p, p, t, v, t, v, v, q, p, p, p, p, t, v, v, q), dim=-3)
    idx = torch.stack([hi, hi + 6, hi + 12], dim=-3)
    return torch.gather(table, -3, idx)


def jitter_op(img, op, brightness, contrast, saturation, hue):
    if op == 0:
        return (img + (brightness - 1.0)).clamp(0, 1)
    if op == 1:
        return (img * contrast).clamp(0, 1)
    h, s, v = rgb_to_hsv(img)
    if op == 2:
        s = (s * saturation).clamp(0, 1)
    else:
        h = torch.fmod(h + hue, TWO_PI)
    return hsv_to_rgb(h, s, v)


def gaussian_taps(ksize, sigma):
    d = torch.arange(ksize, dtype=torch.float32) - ksize // 2
    g = torch.exp(-d * d / (2.0 * sigma * sigma))
    return g / g.sum()


def gaussian_blur(frames, ksize, sigma):
    """frames (N, C, H, W); separable ksize x ksize Gaussian, reflect border"""
    g = gaussian_taps(ksize, sigma)
    C = frames.shape[1]
    r = ksize // 2
    x = F.pad(frames, (r, r, r, r), mode='reflect')
    x = F.conv2d(x, g.view(1, 1, 1, ksize).repeat(C, 1, 1, 1), groups=C)
    return F.conv2d(x, g.view(1, 1, ksize, 1).repeat(C, 1, 1, 1), groups=C)


def color_aug(x, params, blur_ksize=0):
    """x (B,3,T,H,W) fp32 in [0,1]; params (B,16) rows as in include/mscl_hip.h (mscl_color_aug)."""
    out = x.clone()
    for b in range(x.shape[0]):
        P = params[b].tolist()
        img = x[b].permute(1, 0, 2, 3)                          # (T,3,H,W)
        if P[0]:
            for op in P[1:5]:
                img = jitter_op(img, int(op), P[5], P[6], P[7], P[8])
        if P[9]:
            y = 0.299 * img[:, 0] + 0.587 * img[:, 1] + 0.114 * img[:, 2]
            img = y.unsqueeze(1).expand(-1, 3, -1, -1)
        if blur_ksize and P[10] > 0:
            img = gaussian_blur(img.contiguous(), blur_ksize, P[10])
        out[b] = img.permute(1, 0, 2, 3)
    return out
