"""Model registry with the interface of the reference's lib/model_zoo/common/get_model.py (:10-111).

Public surface kept for drop-in use by the evaluator:
  * ``@register(name, version)``       -- class decorator, fills the process-wide table;
  * ``get_model()``                    -- returns that table object; calling it with a config
                                          (``cfg.type`` / ``cfg.args`` / optional ``cfg.pretrained``,
                                          attribute- or key-style) builds the network;
  * ``get_model().model`` / ``.version`` / ``.get_version(name)`` -- lookups used by callers;
  * ``load_state_dict`` / ``save_state_dict`` / ``preprocess_model_args`` -- helpers of the same names.
"""
import copy

import torch

from .utils import get_total_param, get_total_param_sum, get_unit

_WRAPPERS = (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)


def _field(cfg, key, default=None):
    """cfg may be an EasyDict-like object or a plain dict."""
    return cfg.get(key, default) if isinstance(cfg, dict) else getattr(cfg, key, default)


def load_state_dict(net, model_path):
    """Checkpoint values overwrite the freshly initialised ones, absent keys keep theirs
    (the reference merges before loading, get_model.py:10-22).  ``net`` may be a dict of networks
    with a matching dict of paths."""
    if isinstance(net, dict):
        for name in net:
            load_state_dict(net[name], model_path[name])
        return
    state = dict(net.state_dict())
    state.update(torch.load(model_path, map_location='cpu'))
    net.load_state_dict(state)


def save_state_dict(net, path):
    """Saves the bare module's tensors (unwraps DataParallel / DistributedDataParallel)."""
    torch.save((net.module if isinstance(net, _WRAPPERS) else net).state_dict(), path)


def preprocess_model_args(args):
    """Resolves the two indirections a config may carry: activation units given as strings and a
    nested ``backbone`` config that is itself a registered model."""
    out = copy.deepcopy(args if isinstance(args, dict) else dict(args))
    units = out.get('layer_units')
    if units is not None:
        out['layer_units'] = [get_unit()(spec) for spec in units]
    if 'backbone' in out:
        out['backbone'] = get_model()(out['backbone'])
    return out


class _Registry:
    """name -> class (``model``) and name -> version string (``version``)."""

    def __init__(self):
        self.model, self.version, self.verbose = {}, {}, False

    def register(self, model, name, version='x'):
        self.model[name], self.version[name] = model, version

    def get_version(self, name):
        return self.version[name]

    def _ensure_loaded(self, kind):
        # classes register themselves when their defining module is imported
        if kind not in self.model:
            from .. import comodgan, shgan, stylegan  # noqa: F401

    def __call__(self, cfg):
        if cfg is None:
            return None
        kind = _field(cfg, 'type')
        self._ensure_loaded(kind)
        net = self.model[kind](**preprocess_model_args(_field(cfg, 'args', {})))
        ckpt = _field(cfg, 'pretrained')
        if ckpt is not None:
            load_state_dict(net, ckpt)
        if self.verbose:
            print(f'Load {kind} with total {get_total_param(net)} parameters, {get_total_param_sum(net):3f} parameter sum.')
        return net


_TABLE = _Registry()


def get_model():
    """The process-wide registry (the reference spells this as a singleton class of the same name)."""
    return _TABLE


def register(name, version='x'):
    def _decorate(cls):
        _TABLE.register(cls, name, version)
        return cls
    return _decorate
