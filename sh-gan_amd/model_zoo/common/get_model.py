"""Model registry (interface of lib/model_zoo/common/get_model.py:10-111): ``@register(name, version)``
classes are built from a config object with ``.type`` / ``.args`` (attribute or key access) by
``get_model()(cfg)``; optional ``pretrained`` checkpoint is merged into the state dict."""
import copy

import torch

from .utils import get_total_param, get_total_param_sum, get_unit, singleton


def _cfg_get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def load_state_dict(net, model_path):
    """Update-then-load: keys missing from the checkpoint keep their current values (get_model.py:10-22)."""
    if isinstance(net, dict):
        for key, sub in net.items():
            load_state_dict(sub, model_path[key])
        return
    merged = net.state_dict()
    merged.update(torch.load(model_path, map_location=torch.device('cpu')))
    net.load_state_dict(merged)


def save_state_dict(net, path):
    if isinstance(net, (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)):
        net = net.module
    torch.save(net.state_dict(), path)


def preprocess_model_args(args):
    args = copy.deepcopy(dict(args) if not isinstance(args, dict) else args)
    if 'layer_units' in args:
        args['layer_units'] = [get_unit()(u) for u in args['layer_units']]
    if 'backbone' in args:
        args['backbone'] = get_model()(args['backbone'])
    return args


@singleton
class get_model(object):
    def __init__(self):
        self.model = {}
        self.version = {}
        self.verbose = False

    def register(self, model, name, version='x'):
        self.model[name] = model
        self.version[name] = version

    def __call__(self, cfg):
        if cfg is None:
            return None
        kind = _cfg_get(cfg, 'type')
        if kind not in self.model:   # registration happens on import of the defining module
            from .. import comodgan, shgan, stylegan  # noqa: F401
        net = self.model[kind](**preprocess_model_args(_cfg_get(cfg, 'args', {})))
        pretrained = _cfg_get(cfg, 'pretrained', None)
        if pretrained is not None:
            load_state_dict(net, pretrained)
        if self.verbose:
            print('Load {} with total {} parameters, {:3f} parameter sum.'.format(
                kind, get_total_param(net), get_total_param_sum(net)))
        return net

    def get_version(self, name):
        return self.version[name]


def register(name, version='x'):
    def wrapper(cls):
        get_model().register(cls, name, version)
        return cls
    return wrapper
