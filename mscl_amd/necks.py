"""Necks of the MSCL recognizers on HIP kernels (registry names and constructor kwargs of the reference).

ref: mmaction/models/necks/base.py:9-24 (BaseMoCo), :136-175 (TPNMoCo); necks/fpn_video.py:43-136 (TPNSingle);
necks/fpn.py:130-152,188-203 (FPN); necks/sepc.py:16-54,57-135 (SEPC / PConv3D).
"""
import torch
import torch.nn as nn

from .nn import Conv3dHip, conv_bias, pool, upsample
from .registry import NECKS


def global_pool(x):
    """AdaptiveAvgPool3d((1,1,1)) + Flatten on an NDHWC map -> (N, C) fp32 (necks/base.py:17-21)."""
    n, t, h, w, c = x.shape
    return pool(x, n, t * h * w)


@NECKS.register_module()
class BaseMoCo(nn.Module):
    def forward(self, feats, target=None, levels=None):
        return (global_pool(feats[-1]), feats), dict()

    def init_weights(self):
        pass


class _ConvModule(nn.Module):
    """mmcv ConvModule with norm_cfg=None, act_cfg=None == conv + bias under `.conv` (fpn.py:131-149)."""

    def __init__(self, cin, cout, kernel, pad):
        super().__init__()
        self.conv = Conv3dHip(cin, cout, kernel, 1, pad, bias=True)


class FPNHip(nn.Module):
    def __init__(self, in_channels, out_channels, kernel=(1, 3, 3)):
        super().__init__()
        pad = tuple((k - 1) // 2 for k in kernel)
        self.lateral_convs = nn.ModuleList(_ConvModule(c, out_channels, 1, 0) for c in in_channels)
        self.fpn_convs = nn.ModuleList(_ConvModule(out_channels, out_channels, kernel, pad) for _ in in_channels)

    def forward(self, feats, levels=None):
        """top-down: lat[i-1] += nearest_up(lat[i]) (fpn.py:193-203); the add rides the lateral conv's
        epilogue instead of a separate pass.  `levels`: the outputs somebody reads (None = all); the others are None."""
        n = len(feats)
        lat = [None] * n
        lat[n - 1] = conv_bias(self.lateral_convs[n - 1].conv, feats[n - 1])
        for i in range(n - 2, -1, -1):
            up = upsample(lat[i + 1], feats[i].shape[1:4], trilinear=False)
            lat[i] = conv_bias(self.lateral_convs[i].conv, feats[i], addend=up)
        return [conv_bias(self.fpn_convs[i].conv, lat[i]) if levels is None or i in levels else None for i in range(n)]


class PConv3DHip(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.Pconv = nn.ModuleList([
            Conv3dHip(cin, cout, 3, 1, 1, bias=True),
            Conv3dHip(cin, cout, 3, 1, 1, bias=True),
            Conv3dHip(cin, cout, 3, stride, 1, bias=True),
        ])

    def forward(self, xs, levels=None):
        """y_l = relu(P1(x_l) [+ P2(x_{l-1})] [+ trilinear_up(P0(x_{l+1}))])  (sepc.py:118-135).
        Sums and the ReLU ride conv epilogues.  `levels`: output levels to compute (None = all)."""
        L = len(xs)
        out = []
        for l in range(L):
            if levels is not None and l not in levels:
                out.append(None)
                continue
            acc = None
            if l < L - 1:
                acc = upsample(conv_bias(self.Pconv[0], xs[l + 1]), xs[l].shape[1:4], trilinear=True)
            if l > 0:
                acc = conv_bias(self.Pconv[1], xs[l], addend=acc)
                y = conv_bias(self.Pconv[2], xs[l - 1], addend=acc, relu=True)
            else:
                y = conv_bias(self.Pconv[1], xs[l], addend=acc, relu=True)
            out.append(y)
        return out


class SEPCHip(nn.Module):
    def __init__(self, in_channels=[256] * 3, out_channels=256, stride=(2, 1, 1), iBN=False, Pconv_num=2):
        super().__init__()
        if iBN:
            raise NotImplementedError('iBN=True is not used by the MSCL configs (mscl_r18_cosm_lr2e-2.py:23)')
        self.in_channels = in_channels
        self.Pconvs = nn.ModuleList(PConv3DHip(in_channels[i], out_channels, stride) for i in range(Pconv_num))

    def forward(self, xs, levels=None):
        """levels: output levels the caller consumes (None = all).  Only the LAST PConv3D can drop levels: every level of an
        earlier one feeds its successor (sepc.py:118-135).  The skipped outputs are the ones that receive no gradient in the
        reference (`Pconvs.1.Pconv.2` has grad None), so parameter gradients are unchanged."""
        assert len(xs) == len(self.in_channels)
        wanted = self.wanted(levels)
        for i, p in enumerate(self.Pconvs):
            xs = p(xs, levels=wanted[i + 1])
        return xs

    def wanted(self, levels):
        """[the input levels read, the levels PConv3D 0 has to produce, ..., the levels the last one has to produce] (None = all)"""
        L, n = len(self.in_channels), len(self.Pconvs)
        # what each PConv3D has to produce: the last one the caller's levels; an earlier one every level its successor READS -- output
        # l of a PConv3D reads inputs l (P1), l + 1 (P0) and l - 1 (P2).  (Round 5: the first of the two used to compute all three
        # levels although the second, asked for level 0 alone, reads levels 0 and 1: two convs per step whose result nothing read
        # and whose backward never ran.)
        wanted = [None] * (n + 1)
        wanted[n] = None if levels is None else tuple(sorted(levels))
        for i in range(n - 1, -1, -1):
            nxt = wanted[i + 1]
            wanted[i] = None if nxt is None else tuple(sorted({m for l in nxt for m in (l - 1, l, l + 1) if 0 <= m < L}))
        return wanted


class TPNSingleHip(nn.Module):
    def __init__(self, in_channels, out_channels, fpn_cfg, temporal_modulation_cfg, sepc_cfg, reverse_st=False):
        super().__init__()
        assert isinstance(in_channels, list) and isinstance(out_channels, int)
        if temporal_modulation_cfg is not None or reverse_st:
            raise NotImplementedError('temporal modulation / reverse_st are not used by the MSCL configs')
        self.num_tpn_stages = len(in_channels)
        self.fpn = FPNHip(in_channels, out_channels, kernel=tuple(fpn_cfg.get('fpn_kerne_size', (1, 3, 3))))
        self.sepc = SEPCHip(**sepc_cfg) if sepc_cfg is not None else None
        self.init_weights()

    def init_weights(self):
        """every Conv3d under the neck ends up xavier-uniform with zero bias: TPNSingle.init_weights
        (fpn_video.py:97-108) runs after, and overwrites, PConv3D's normal(0, 0.01)."""
        for m in self.modules():
            if isinstance(m, Conv3dHip):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, feats, levels=None):
        feats = list(feats[-self.num_tpn_stages:])
        if self.sepc is not None:
            return self.sepc(self.fpn(feats, levels=self.sepc.wanted(levels)[0]), levels=levels)
        return self.fpn(feats, levels=levels)


@NECKS.register_module()
class TPNMoCo(nn.Module):
    def __init__(self, in_channels, out_channels,
                 fpn_cfg=dict(fpn_kerne_size=(1, 3, 3), conv_cfg=dict(type='Conv3d')),
                 temporal_modulation_cfg=None, sepc_cfg=None, reverse_st=False, emb_from_bkb=True):
        super().__init__()
        self.tpn = TPNSingleHip(list(in_channels), out_channels, fpn_cfg, temporal_modulation_cfg,
                                dict(sepc_cfg) if sepc_cfg is not None else None, reverse_st=reverse_st)
        self.emb_from_bkb = emb_from_bkb

    def init_weights(self):
        self.tpn.init_weights()

    def forward(self, feats, target=None, levels=None):
        """ref: necks/base.py:167-175.  `levels` (an MI355X-side argument, not in the reference): the pyramid levels the
        caller reads -- () skips the pyramid altogether when the embedding comes from the backbone (`emb_from_bkb`), which
        is the key branch of MSCL: its `k_mlvl` is never consumed (recognizers/moco.py:535-545, mscl.py:225-277), 145 GFLOP
        per step at B=8 that change no output.  Levels not asked for come back as None."""
        n = self.tpn.num_tpn_stages
        if levels is not None:
            levels = tuple(sorted({l % n for l in levels}))
        if self.emb_from_bkb:
            emb = global_pool(feats[-1])
            if levels is not None and len(levels) == 0:
                return (emb, [None] * n), {}
            outs = self.tpn(feats, levels=levels)
        else:
            outs = self.tpn(feats, levels=None if levels is None else tuple(sorted(set(levels) | {n - 1})))
            emb = global_pool(outs[-1])
        return (emb, outs), {}
