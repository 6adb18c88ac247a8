"""Python-file configs with `_base_` inheritance and attribute access (the part of mmcv.Config the
reference's launch path uses: tools/train.py:80-84)."""
import os


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, (list, tuple)):
        return type(v)(_wrap(x) for x in v)
    return v


def _merge(base, new):
    out = dict(base)
    for k, v in new.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            out[k] = {kk: vv for kk, vv in v.items() if kk != '_delete_'} if isinstance(v, dict) else v
    return out


def _load(path):
    ns = {}
    with open(path) as f:
        exec(compile(f.read(), path, 'exec'), ns)
    cfg = {k: v for k, v in ns.items() if not k.startswith('__') and not callable(v) and not isinstance(v, type(os))}
    bases = cfg.pop('_base_', [])
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        merged = _merge(merged, _load(os.path.normpath(os.path.join(os.path.dirname(path), b))))
    return _merge(merged, cfg)


class Config(ConfigDict):
    @staticmethod
    def fromfile(path):
        cfg = Config(_wrap(_load(os.path.abspath(path))))
        dict.__setitem__(cfg, 'filename', os.path.abspath(path))
        return cfg

    def merge_from_dict(self, options):
        """`--cfg-options a.b=c` style overrides."""
        for key, val in options.items():
            d = self
            parts = key.split('.')
            for p in parts[:-1]:
                d = d.setdefault(p, ConfigDict())
            d[parts[-1]] = _wrap(val)
