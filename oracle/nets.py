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


class _ConvModuleBN(nn.Module):
    """mmcv ConvModule with norm_cfg=BN3d: bias-free conv under `.conv`, BatchNorm3d under `.bn`, optional ReLU
    (call sites backbones/resnet3d.py:262-296,448-459, resnet3d_slowfast.py:157-166)."""

    def __init__(self, cin, cout, kernel, stride, pad, relu):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, kernel, stride, pad, bias=False)
        self.bn = nn.BatchNorm3d(cout)
        self.relu = relu

    def forward(self, x):
        x = self.bn(self.conv(x))
        return F.relu(x) if self.relu else x


class BottleneckUnit(nn.Module):
    """relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1 x))))))) + shortcut(x)), the stride on conv2 ('pytorch' style).
    flavour 'mmcv': mmaction Bottleneck3d (backbones/resnet3d.py:162-330) -- conv1 is 3x1x1 when the block is inflated
                    ('3x1x1' style) else 1x1x1, conv2 1x3x3, conv3 1x1x1, sub-modules named conv / bn;
    flavour 'tv'  : the flow trunk's Bottleneck (backbones/fastonly.py:137-183) -- conv1 1x1x1, conv2 Conv3DNoTemporal
                    (fastonly.py:61-80), sub-modules are nn.Sequential (names 0 / 1)."""

    def __init__(self, cin, planes, stride, downsample, flavour, inflate=False):
        super().__init__()
        cout = 4 * planes
        s3 = (1, stride, stride)
        if flavour == 'mmcv':
            k1, p1 = ((3, 1, 1), (1, 0, 0)) if inflate else ((1, 1, 1), (0, 0, 0))
            self.conv1 = _ConvModuleBN(cin, planes, k1, 1, p1, True)
            self.conv2 = _ConvModuleBN(planes, planes, (1, 3, 3), s3, (0, 1, 1), True)
            self.conv3 = _ConvModuleBN(planes, cout, 1, 1, 0, False)
            self.downsample = _ConvModuleBN(cin, cout, 1, s3, 0, False) if downsample else None
        else:
            self.conv1 = _conv_bn(cin, planes, 1, 1, 0, relu=True)
            self.conv2 = _conv_bn(planes, planes, (1, 3, 3), s3, (0, 1, 1), relu=True)
            self.conv3 = _conv_bn(planes, cout, 1, 1, 0, relu=False)
            self.downsample = _conv_bn(cin, cout, 1, s3, 0, relu=False) if downsample else None

    def forward(self, x):
        y = self.conv3(self.conv2(self.conv1(x)))
        sc = x if self.downsample is None else self.downsample(x)
        return F.relu(y + sc)


def _bottleneck_stage(cin, planes, blocks, stride, flavour, inflate):
    units = [BottleneckUnit(cin, planes, stride, stride != 1 or cin != 4 * planes, flavour, inflate)]
    units += [BottleneckUnit(4 * planes, planes, 1, False, flavour, inflate) for _ in range(blocks - 1)]
    return nn.Sequential(*units)


class SlowOnly50(nn.Module):
    """ResNet3dSlowOnly depth 50 as configs/recognition/moco/mscl_r50_cosm_lr3e-2.py:16-26 builds it, returning the four
    stage maps (out_indices (0,1,2,3)).  ref: backbones/resnet3d_slowonly.py:15-52 (inflate (0,0,1,1), no pool2, lateral off),
    backbones/resnet3d.py:448-467 (stem conv (5,7,7) / stride (2,2,2) / pad (2,3,3) + BN + ReLU, max-pool (1,3,3) / (1,2,2) /
    (0,1,1)), :407-415 (depth 50 = Bottleneck3d x (3,4,6,3)), resnet3d_slowfast.py:89-204 (make_res_layer: a 1x1x1 shortcut
    conv + BN whenever stride != 1 or the width changes), resnet3d.py:845-860 (forward)."""

    def __init__(self, conv1_kernel=(5, 7, 7), conv1_stride_t=2, pool1_stride_t=1):
        super().__init__()
        pad = tuple((k - 1) // 2 for k in conv1_kernel)
        self.conv1 = _ConvModuleBN(3, 64, conv1_kernel, (conv1_stride_t, 2, 2), pad, True)
        self.maxpool = nn.MaxPool3d((1, 3, 3), (pool1_stride_t, 2, 2), (0, 1, 1))
        cin = 64
        for li, (planes, blocks, stride, inflate) in enumerate(((64, 3, 1, False), (128, 4, 2, False), (256, 6, 2, True), (512, 3, 2, True)), 1):
            setattr(self, f'layer{li}', _bottleneck_stage(cin, planes, blocks, stride, 'mmcv', inflate))
            cin = 4 * planes

    def forward(self, x):
        x = self.maxpool(self.conv1(x))
        outs = []
        for li in range(1, 5):
            x = getattr(self, f'layer{li}')(x)
            outs.append(x)
        return outs


class FlowR2D50(nn.Module):
    """resnet_flow.r2d_50: fastonly.py:431-441 (Bottleneck x (3,4,6,3), Conv3DNoTemporal), :238-262 (inplanes 8, widths
    8..64 x 4), :222-235 (BottleneckStem: conv (1,7,7) / (2,2,2) / (0,3,3) to 8 channels + BN + ReLU + max-pool (1,3,3) /
    (1,2,2) / (0,1,1)), :291-310 (_make_layer); forward == the patched multi-level forward (recognizers/moco.py:12-24)."""

    def __init__(self):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv3d(3, 8, (1, 7, 7), (2, 2, 2), (0, 3, 3), bias=False), nn.BatchNorm3d(8), nn.ReLU(inplace=True),
                                  nn.MaxPool3d((1, 3, 3), (1, 2, 2), (0, 1, 1)))
        cin = 8
        for li, (planes, blocks, stride) in enumerate(((8, 3, 1), (16, 4, 2), (32, 6, 2), (64, 3, 2)), 1):
            setattr(self, f'layer{li}', _bottleneck_stage(cin, planes, blocks, stride, 'tv', False))
            cin = 4 * planes
        self.avgpool = nn.AdaptiveAvgPool3d((1, 1, 1))
        self.fc = nn.Identity()

    def forward(self, x):
        x = self.stem(x)
        outs = []
        for li in range(1, 5):
            x = getattr(self, f'layer{li}')(x)
            outs.append(x)
        return outs


def build_trunk(kind):
    """'rgb' / 'flow': the mscl_r18 pair; 'rgb50' / 'flow50': the mscl_r50 pair"""
    if kind in ('rgb', 'flow'):
        return VideoResNet18(kind)
    return SlowOnly50() if kind == 'rgb50' else FlowR2D50()


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
