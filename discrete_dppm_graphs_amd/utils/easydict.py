"""Attribute-access dict with the subset of easydict.EasyDict the reference uses (utils/arg_helper.py:13,45;
ppgn_simple.py:283).  The real `easydict` package is used when it is installed."""
try:
    from easydict import EasyDict  # noqa: F401
except ImportError:
    class EasyDict(dict):
        def __init__(self, d=None, **kwargs):
            super().__init__()
            d = dict(d or {}, **kwargs)
            for k, v in d.items():
                setattr(self, k, v)

        def __setattr__(self, name, value):
            if isinstance(value, (list, tuple)):
                value = type(value)(EasyDict(x) if isinstance(x, dict) and not isinstance(x, EasyDict) else x for x in value)
            elif isinstance(value, dict) and not isinstance(value, EasyDict):
                value = EasyDict(value)
            super().__setitem__(name, value)

        __setitem__ = __setattr__

        def __getattr__(self, name):
            try:
                return self[name]
            except KeyError:
                raise AttributeError(name)

        def update(self, e=None, **f):
            d = dict(e or {}, **f)
            for k, v in d.items():
                setattr(self, k, v)
