"""Oracle trunks and necks (fp32, plain torch.nn, any device).  TEST INFRASTRUCTURE ONLY.

Parameter names equal the reference's state-dict names so the closed-form fill and checkpoints map
1:1.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _conv_bn(cin, cout, kernel, stride, pad, relu):
    mods = [nn.Conv3d(cin, cout, kernel, stride, pad, bias=False), nn.BatchNorm3d(cout)]
    if relu:
        mods.append(nn.ReLU(inplace=True))
    return nn.Sequential(*mods)


class ResidualUnit(nn.Module):
    """BasicBlock: relu(bn(conv2(relu(bn(conv1 x)))) + shortcut(x)).
    ref: mmaction/models/backbones/r3d.py:95-127 (RGB twin of torchvision BasicBlock),
         mmaction/models/backbones/fastonly.py:104-136 (flow)."""

    def __init__(self, cin, cout, kernel, stride, pad, shortcut_stride=None):
        super().__init__()
        self.conv1 = _conv_bn(cin, cout, kernel, stride, pad, relu=True)
        self.conv2 = _conv_bn(cout, cout, kernel, 1, pad, relu=False)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if shortcut_stride is not None:
            self.downsample = _conv_bn(cin, cout, 1, shortcut_stride, 0, relu=False)

    def forward(self, x):
        y = self.conv2(self.conv1(x))
        sc = x if self.downsample is None else self.downsample(x)
        return self.relu(y + sc)


class VideoResNet18(nn.Module):
    """Two-block-per-stage video ResNet returning the four stage maps.

    kind='rgb' : torchvision r3d_18 == r3d.py:176-184 (stem (3,7,7) s(1,2,2)), r3d.py:16-39
                 (3x3x3 convs, stride (s,s,s)), r3d.py:260-296 (_make_layer; 1x1x1 shortcut with
                 the conv builder's downsample stride), widths 64..512.
    kind='flow': fastonly.py:185-193 (stem (1,7,7) s(2,2,2) p(0,3,3), 16 ch), fastonly.py:61-80
                 ((1,3,3) convs, stride (1,s,s); shortcut stride (1,s,s)), fastonly.py:238-326,
                 r2d_18 at fastonly.py:399-408, widths 16..128.
    forward == the patched multi-level forward, recognizers/moco.py:12-24.
    """

    def __init__(self, kind):
        super().__init__()
        if kind == 'rgb':
            base, kernel, pad = 64, (3, 3, 3), (1, 1, 1)
            self.stem = _conv_bn(3, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3), relu=True)
            st = lambda s: (s, s, s)
        elif kind == 'flow':
            base, kernel, pad = 16, (1, 3, 3), (0, 1, 1)
            self.stem = _conv_bn(3, 16, (1, 7, 7), (2, 2, 2), (0, 3, 3), relu=True)
            st = lambda s: (1, s, s)
        else:
            raise ValueError(kind)
        cin = base
        for li, mult in enumerate((1, 2, 4, 8), start=1):
            cout, s = base * mult, (1 if li == 1 else 2)
            first = ResidualUnit(cin, cout, kernel, st(s), pad,
                                 shortcut_stride=st(s) if (s != 1 or cin != cout) else None)
            setattr(self, f'layer{li}', nn.Sequential(first, ResidualUnit(cout, cout, kernel, 1, pad)))
            cin = cout
        # the reference swaps `fc` for nn.Identity (base_moco.py:90-91,100-101): no parameters
        self.avgpool = nn.AdaptiveAvgPool3d((1, 1, 1))
        self.fc = nn.Identity()

    def forward(self, x):
        x = self.stem(x)
        outs = []
        for li in range(1, 5):
            x = getattr(self, f'layer{li}')(x)
            outs.append(x)
        return outs


class _BiasConv(nn.Module):
    """mmcv ConvModule with norm_cfg=None, act_cfg=None: a conv with bias under `.conv`
    (call sites necks/fpn.py:131-149)."""

    def __init__(self, cin, cout, kernel, pad):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, kernel, padding=pad)

    def forward(self, x):
        return self.conv(x)


class FPN(nn.Module):
    """ref: necks/fpn.py:130-152 (1x1x1 laterals, (1,3,3) output convs, both with bias, no norm, no
    activation) and necks/fpn.py:188-203 (top-down: nearest upsample to the finer level's size, add)."""

    def __init__(self, in_channels, out_channels, kernel=(1, 3, 3)):
        super().__init__()
        pad = tuple((k - 1) // 2 for k in kernel)
        self.lateral_convs = nn.ModuleList(_BiasConv(c, out_channels, 1, 0) for c in in_channels)
        self.fpn_convs = nn.ModuleList(_BiasConv(out_channels, out_channels, kernel, pad) for _ in in_channels)

    def forward(self, feats):
        lat = [m(f) for m, f in zip(self.lateral_convs, feats)]
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode='nearest')
        return [m(x) for m, x in zip(self.fpn_convs, lat)]


class PConv3D(nn.Module):
    """ref: necks/sepc.py:57-135.  Three 3x3x3 convs with bias shared over pyramid levels:
    y_l = P1(x_l) [+ P2(x_{l-1}), stride (2,2,2)] [+ trilinear_up(P0(x_{l+1}))], then ReLU."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.Pconv = nn.ModuleList([
            nn.Conv3d(cin, cout, 3, padding=1),
            nn.Conv3d(cin, cout, 3, padding=1),
            nn.Conv3d(cin, cout, 3, padding=1, stride=stride),
        ])

    def forward(self, xs):
        out = []
        for l, x in enumerate(xs):
            y = self.Pconv[1](x)
            if l > 0:
                y = y + self.Pconv[2](xs[l - 1])
            if l < len(xs) - 1:
                y = y + F.interpolate(self.Pconv[0](xs[l + 1]), size=list(y.shape[2:]), mode='trilinear')
            out.append(F.relu(y))
        return out


class SEPC(nn.Module):
    """ref: necks/sepc.py:16-54 (iBN=False path)."""

    def __init__(self, in_channels, out_channels, stride, iBN=False, Pconv_num=2):
        super().__init__()
        assert not iBN, 'iBN is not used by mscl_r18 (mscl_r18_cosm_lr2e-2.py:23)'
        self.Pconvs = nn.ModuleList(PConv3D(in_channels[i], out_channels, stride) for i in range(Pconv_num))

    def forward(self, xs):
        for p in self.Pconvs:
            xs = p(xs)
        return xs


class _TPN(nn.Module):
    """ref: necks/fpn_video.py:43-136 with temporal_modulation_cfg=None, reverse_st=False."""

    def __init__(self, in_channels, out_channels, sepc_cfg):
        super().__init__()
        self.n = len(in_channels)
        self.fpn = FPN(in_channels, out_channels)
        self.sepc = SEPC(**sepc_cfg) if sepc_cfg is not None else None

    def forward(self, feats):
        outs = self.fpn(feats[-self.n:])
        return self.sepc(outs) if self.sepc is not None else outs


class TPNMoCo(nn.Module):
    """ref: necks/base.py:136-175: x_emb = global-avg-pool(layer4) (emb_from_bkb=True),
    multi-level output = SEPC(FPN(layers 2..4))."""

    def __init__(self, in_channels, out_channels, sepc_cfg=None):
        super().__init__()
        self.tpn = _TPN(in_channels, out_channels, sepc_cfg)

    def forward(self, feats):
        emb = F.adaptive_avg_pool3d(feats[-1], 1).flatten(1)
        return emb, self.tpn(feats)


class BaseMoCo(nn.Module):
    """ref: necks/base.py:9-24: global average pool of the last map; maps passed through."""

    def forward(self, feats):
        return F.adaptive_avg_pool3d(feats[-1], 1).flatten(1), feats
