"""Registry with the surface the reference gets from mmcv (`Registry('models')`, `register_module()`,
`build(cfg)`), restated because mmcv is not a dependency here.

ref: mmaction/models/builder.py:9-16 -- ONE shared registry aliased as BACKBONES / NECKS / HEADS /
RECOGNIZERS / LOSSES / LOCALIZERS plus a separate SSL_AUGS; construction pops `type` and passes the
remaining keys as keyword arguments (builder.py:29-97).
"""


class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def __contains__(self, key):
        return key in self._modules

    def __len__(self):
        return len(self._modules)

    def get(self, key):
        return self._modules.get(key)

    def _add(self, cls, name=None, force=False):
        key = name or cls.__name__
        if key in self._modules and not force:
            raise KeyError(f'{key} is already registered in {self.name}')
        self._modules[key] = cls
        return cls

    def register_module(self, name=None, force=False, module=None):
        """usable bare (`@R.register_module`), called (`@R.register_module()`), or as a function."""
        if isinstance(name, type):
            return self._add(name)
        if module is not None:
            return self._add(module, name, force)
        return lambda cls: self._add(cls, name, force)

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict) or 'type' not in cfg:
            raise TypeError(f'cfg must be a dict with a "type" key, got {cfg!r}')
        args = dict(cfg)
        for k, v in (default_args or {}).items():
            args.setdefault(k, v)
        typ = args.pop('type')
        if isinstance(typ, str):
            cls = self.get(typ)
            if cls is None:
                raise KeyError(f'{typ} is not in the {self.name} registry')
        else:
            cls = typ
        return cls(**args)


MODELS = Registry('models')
BACKBONES = NECKS = HEADS = RECOGNIZERS = LOSSES = LOCALIZERS = MODELS
SSL_AUGS = Registry('ssl_augs')


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_ssl_aug(cfg):
    return SSL_AUGS.build(cfg)


def build_recognizer(cfg, train_cfg=None, test_cfg=None):
    return RECOGNIZERS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_model(cfg, train_cfg=None, test_cfg=None):
    """ref: builder.py:63-87 (only the recognizer branch exists on this path)."""
    if cfg.get('type') not in RECOGNIZERS:
        raise ValueError(f"{cfg.get('type')} is not registered in RECOGNIZERS")
    return build_recognizer(cfg, train_cfg, test_cfg)
