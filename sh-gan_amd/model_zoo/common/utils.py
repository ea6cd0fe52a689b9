"""Unit registry and the ``lrelu_agc`` activation (interface of lib/model_zoo/common/utils.py:39-143).

``get_unit()('lrelu_agc(alpha=0.2, gain=sqrt_2, clamp=256)')`` returns a constructor whose instance is a
plain callable ``act(x, gain=1)``; on a HIP tensor it runs the fused bias_act kernel."""
import functools
import math

import torch
import torch.nn as nn

from ... import kernels

_SINGLETONS = {}


def singleton(cls):
    def instance(*args, **kwargs):
        if cls not in _SINGLETONS:
            _SINGLETONS[cls] = cls(*args, **kwargs)
        return _SINGLETONS[cls]
    return instance


def str2value(text):
    """'3' -> 3, '0.2' -> 0.2, 'true' -> True, anything else stays a stripped string."""
    text = text.strip()
    for cast in (int, float):
        try:
            return cast(text)
        except ValueError:
            pass
    low = text.lower()
    if low in ('true', 'false') and text in ('True', 'true', 'False', 'false'):
        return low == 'true'
    return text


def _split_top_level(argstr):
    """Split 'a=1, b=(2,3), c=[4,5]' on commas that are not inside brackets."""
    parts, depth, cur = [], 0, ''
    for ch in argstr:
        if ch in '([':
            depth += 1
        elif ch in ')]':
            depth -= 1
        if ch == ',' and depth == 0:
            parts.append(cur)
            cur = ''
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return parts


@singleton
class get_unit(object):
    """String -> layer/activation constructor, e.g. 'relu', 'lrelu_agc(alpha=0.2, gain=sqrt_2)'."""

    def __init__(self):
        self.unit = {}
        self.register('none', None)
        for name, cls in (('conv', nn.Conv2d), ('bn', nn.BatchNorm2d), ('relu', nn.ReLU), ('relu6', nn.ReLU6),
                          ('lrelu', nn.LeakyReLU), ('dropout', nn.Dropout), ('dropout2d', nn.Dropout2d)):
            self.register(name, cls)

    def register(self, name, unitf):
        self.unit[name] = unitf

    def __call__(self, name):
        if name is None:
            return None
        head, _, rest = name.partition('(')
        ctor = self.unit[head.strip()]
        argstr = rest.rstrip()
        if argstr.endswith(')'):
            argstr = argstr[:-1]
        if not argstr.strip():
            return ctor
        kwargs = {}
        for item in _split_top_level(argstr):
            key, _, val = item.partition('=')
            val = val.strip()
            if val[:1] == '(' and val[-1:] == ')':
                kwargs[key.strip()] = tuple(str2value(v) for v in val[1:-1].split(','))
            elif val[:1] == '[' and val[-1:] == ']':
                kwargs[key.strip()] = [str2value(v) for v in val[1:-1].split(',')]
            else:
                kwargs[key.strip()] = str2value(val)
        return functools.partial(ctor, **kwargs)


def register(name):
    def wrapper(cls):
        get_unit().register(name, cls)
        return cls
    return wrapper


@register('lrelu_agc')
class lrelu_agc(object):
    """leaky-relu(alpha) -> * (gain * call_gain) -> clamp(+-clamp * call_gain); a callable, not a Module."""

    def __init__(self, alpha=0.1, gain=1, clamp=None):
        self.alpha = alpha
        self.gain = math.sqrt(2) if gain == 'sqrt_2' else gain
        self.clamp = clamp
        self.repr = 'lrelu_agc(alpha={}, gain={}, clamp={})'.format(alpha, gain, clamp)

    def __call__(self, x, gain=1):
        if torch.is_grad_enabled() and x.requires_grad:
            from ..stylegan_utils import grad_ops
            return grad_ops.bias_act(x, None, act=True, gain=gain, alpha=self.alpha, act_gain=self.gain, clamp=self.clamp)
        shape = x.shape
        x4 = x.reshape(shape[0], -1, 1, 1) if x.ndim != 4 else x
        y = kernels.bias_act(x4, act=True, gain=gain, alpha=self.alpha, act_gain=self.gain, clamp=self.clamp)
        return y.reshape(shape)

    def fused_args(self, gain=1):
        """(alpha, act_gain, clamp) consumed by the conv / FIR epilogues."""
        return dict(alpha=self.alpha, act_gain=self.gain, clamp=self.clamp)

    def __repr__(self):
        return self.repr


def get_total_param(net):
    return sum(p.numel() for p in net.parameters())


def get_total_param_sum(net):
    return float(sum(p.detach().double().sum().item() for p in net.parameters()))
