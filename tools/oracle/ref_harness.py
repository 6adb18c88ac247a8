"""Import the reference's MSCL hot-path *files* under a stub harness.

THIS RUNS ONLY IN THE DEVELOPMENT CONTAINER, where /root/reference exists.  Nothing here is
imported by the product, by tests marked gpu, by smoke() or by bench.py.  Its one job is to let
tools/oracle/make_golden.py run the reference's own Python on CPU so that oracle/ can be pinned
against it and small golden vectors can be written to tests/golden/.

Why a harness at all: the reference package cannot be imported as shipped (SURVEY.md §0 fact 4:
mmcv / torchvision / kornia are absent, moco_head_v2.py imports a symbol that does not exist,
`.cuda()` and torch.distributed are unconditional).  The hot-path files themselves are fine, so we
register *package shells* (so no reference __init__.py runs) and import the real files one by one.

Third-party pieces that are NOT under /root/reference and therefore are restated here, not
imported (parity for them is "unpinned by the reference"):
  * mmcv.utils.Registry, mmcv.cnn.ConvModule / xavier_init / constant_init / normal_init
    (mmcv-full 1.3.6..1.4.0, mmaction/__init__.py:7-14)
  * torchvision.models.video.r3d_18 -> the reference's own vendored twin
    mmaction/models/backbones/r3d.py (identical parameter names / shapes)
"""
import importlib
import os
import sys
import types
from unittest.mock import MagicMock

import torch
import torch.nn as nn

REF = '/root/reference'


# --------------------------------------------------------------------------- mmcv stand-ins
class Registry:
    def __init__(self, name, parent=None, **kw):
        self.name = name
        self.parent = parent
        self._module_dict = parent._module_dict if parent is not None else {}

    def __contains__(self, key):
        return key in self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def _register(self, cls, name=None):
        self._module_dict[name or cls.__name__] = cls
        return cls

    def register_module(self, name=None, force=False, module=None):
        if isinstance(name, type):            # used bare: @X.register_module
            return self._register(name)
        if module is not None:
            return self._register(module, name)
        return lambda cls: self._register(cls, name)

    def build(self, cfg, default_args=None):
        args = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        typ = args.pop('type')
        cls = self._module_dict[typ] if isinstance(typ, str) else typ
        return cls(**args)


def build_from_cfg(cfg, registry, default_args=None):
    return registry.build(cfg, default_args)


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    if getattr(module, 'weight', None) is not None:
        if distribution == 'uniform':
            nn.init.xavier_uniform_(module.weight, gain=gain)
        else:
            nn.init.xavier_normal_(module.weight, gain=gain)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if getattr(module, 'weight', None) is not None:
        nn.init.constant_(module.weight, val)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def normal_init(module, mean=0, std=1, bias=0):
    if getattr(module, 'weight', None) is not None:
        nn.init.normal_(module.weight, mean, std)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


class ConvModule(nn.Module):
    """mmcv.cnn.ConvModule (mmcv-full 1.3.x, absent here), order conv -> norm -> act.  TPNMoCo uses the norm_cfg=None path
    (fpn.py:131-149: conv with bias); the ResNet3d family (backbones/resnet3d.py:262-296,448-459) uses norm_cfg=BN3d, where
    mmcv names the norm layer `bn`, drops the conv bias (bias='auto') and initialises conv by kaiming_init / norm to 1."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'),
                 inplace=True, **kw):
        super().__init__()
        self.with_norm = norm_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        typ = (conv_cfg or dict(type='Conv2d'))['type']
        conv_cls = {'Conv3d': nn.Conv3d, 'Conv2d': nn.Conv2d, 'Conv1d': nn.Conv1d}[typ]
        self.conv = conv_cls(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                             dilation=dilation, groups=groups, bias=bias)
        if self.with_norm:
            assert norm_cfg['type'] == 'BN3d' and typ == 'Conv3d', norm_cfg
            self.bn = nn.BatchNorm3d(out_channels)
            for p in self.bn.parameters():
                p.requires_grad = norm_cfg.get('requires_grad', True)
        self.activate = nn.ReLU(inplace=(act_cfg or {}).get('inplace', inplace)) if act_cfg is not None else None
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')
        if bias:
            nn.init.constant_(self.conv.bias, 0)

    @property
    def norm(self):
        return self.bn

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = self.bn(x)
        if self.activate is not None:
            x = self.activate(x)
        return x


def kaiming_init(module, a=0, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
    if getattr(module, 'weight', None) is not None:
        (nn.init.kaiming_uniform_ if distribution == 'uniform' else nn.init.kaiming_normal_)(
            module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def build_activation_layer(cfg):
    cfg = dict(cfg)
    assert cfg.pop('type') == 'ReLU', cfg
    return nn.ReLU(**cfg)


def auto_fp16(*a, **k):
    def deco(fn):
        return fn
    return deco


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _shell(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


_installed = {}


def install():
    """Install stubs + shells, import the reference hot-path files; idempotent."""
    if _installed:
        return _installed
    assert os.path.isdir(REF), 'reference not mounted: this tool only runs in the dev container'
    if REF not in sys.path:
        sys.path.insert(0, REF)

    mm_models = Registry('models')
    _mod('mmcv')
    _mod('mmcv.cnn', MODELS=mm_models, ConvModule=ConvModule, xavier_init=xavier_init,
         constant_init=constant_init, normal_init=normal_init, kaiming_init=kaiming_init,
         NonLocal3d=None, build_activation_layer=build_activation_layer)
    _mod('mmcv.utils', Registry=Registry, _BatchNorm=nn.modules.batchnorm._BatchNorm,
         build_from_cfg=build_from_cfg, get_logger=lambda *a, **k: MagicMock(), print_log=lambda *a, **k: None)
    _mod('mmcv.runner', auto_fp16=auto_fp16, _load_checkpoint=None, load_state_dict=None, load_checkpoint=None)
    _mod('torchvision')
    _mod('torchvision.models')
    _mod('torchvision.models.utils', load_state_dict_from_url=None)
    tvv = _mod('torchvision.models.video')
    sys.modules['torchvision'].models = sys.modules['torchvision.models']
    sys.modules['torchvision.models'].video = tvv
    sys.modules['torchvision.models'].utils = sys.modules['torchvision.models.utils']

    mma = _shell('mmaction', f'{REF}/mmaction')
    _shell('mmaction.models', f'{REF}/mmaction/models')
    for sub in ('recognizers', 'heads', 'necks', 'backbones', 'losses', 'common'):
        _shell(f'mmaction.models.{sub}', f'{REF}/mmaction/models/{sub}')
    core = _shell('mmaction.core', f'{REF}/mmaction/core')
    _shell('mmaction.core.evaluation', f'{REF}/mmaction/core/evaluation')
    utils = _shell('mmaction.utils', f'{REF}/mmaction/utils')
    utils.import_module_error_func = lambda name: (lambda fn: fn)
    utils.get_root_logger = lambda *a, **k: MagicMock()
    mma.utils = utils
    for n in ('tools', 'tools.RAFT', 'tools.RAFT.core', 'tools.RAFT.core.utils'):
        _shell(n, f'{REF}/' + n.replace('.', '/'))

    imp = importlib.import_module
    builder = imp('mmaction.models.builder')
    acc = imp('mmaction.core.evaluation.accuracy')
    core.top_k_accuracy = acc.top_k_accuracy
    core.bbox_overlaps = None
    imp('mmaction.models.losses.cross_entropy_loss')
    necks = imp('mmaction.models.necks.base')
    r3d = imp('mmaction.models.backbones.r3d')

    def r3d_18(**kw):
        return r3d.R3D(block='BasicBlock', conv_makers='Conv3DSimple', layers=[2, 2, 2, 2],
                       stem='BasicStem', **kw)
    tvv.r3d_18 = r3d_18
    moco = imp('mmaction.models.recognizers.moco')
    mscl = imp('mmaction.models.recognizers.mscl')
    imp('mmaction.models.heads.moco_head')
    imp('mmaction.models.heads.local_cl_head')
    mscl.forward = moco.forward            # missing symbol
    imp('mmaction.models.heads.moco_head_v2')
    fastonly = imp('mmaction.models.backbones.fastonly')
    # resnet3d_slowonly.py:2 has a stray, unused `from turtle import forward` (an editor auto-import); the stdlib module needs
    # tkinter, which this image lacks, so an empty shell satisfies the import
    _mod('turtle', forward=None)
    imp('mmaction.models.backbones.resnet3d')
    imp('mmaction.models.backbones.resnet3d_slowfast')
    slowonly = imp('mmaction.models.backbones.resnet3d_slowonly')      # registers ResNet3dSlowOnly (mscl_r50 RGB trunk)

    # deterministic aug used on both sides (SURVEY.md §8c "OracleAug"): RGB -> ImageNet normalise
    # (ssl_aug_v2.py:66-68); flow views are fed already visualised, flow_normalizer is Identity
    # because normalize_flow=False (ssl_aug_v2.py:88).
    class OracleAug:
        def __init__(self, **kw):
            self.mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1, 1)
            self.std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1, 1)

        def __call__(self, im_q, im_k, aux):
            return (im_q - self.mean) / self.std, (im_k - self.mean) / self.std, aux

    class IdentityAug:
        def __init__(self, **kw):
            pass

        def __call__(self, clips):
            return clips
    builder.SSL_AUGS.register_module(module=OracleAug, name='OracleAug')
    builder.SSL_AUGS.register_module(module=IdentityAug, name='IdentityAug')

    torch.Tensor.cuda = lambda self, *a, **k: self
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('gloo', rank=0, world_size=1)

    _installed.update(builder=builder, moco=moco, mscl=mscl, r3d=r3d, fastonly=fastonly,
                      necks=necks, accuracy=acc, slowonly=slowonly)
    return _installed


def load_ref_cfg(name='mscl_r18_cosm_lr2e-2.py'):
    """exec the reference config text unchanged (only `_base_` is ignored)."""
    ns = {}
    with open(f'{REF}/configs/recognition/moco/{name}') as f:
        exec(compile(f.read(), name, 'exec'), ns)
    return ns


def build_ref_model(num_frames=8, K=65536, cfg_name='mscl_r18_cosm_lr2e-2.py'):
    """Unmodified reference `model` dict, except: deterministic aug, sup_head.t from num_frames
    (the config derives it the same way, mscl_r18_cosm_lr2e-2.py:47), optional small K."""
    h = install()
    cfg = load_ref_cfg(cfg_name)
    model = cfg['model']
    model['aug'] = dict(type='OracleAug')
    model['sup_head']['t'] = num_frames // 2
    model['recognizer']['K'] = K
    model['recognizer_flow']['K'] = K
    m = h['builder'].build_recognizer(model)
    return m, cfg
