"""``conv2d`` / ``conv_transpose2d`` with the call signature of
lib/model_zoo/stylegan_utils/conv2d_gradfix.py:35-43 -- forward semantics only (the reference's
custom autograd path is active solely for torch 1.7-1.9 training).  Both run on the fp32-MFMA
implicit-GEMM kernel; supported geometry is what the generator needs: 3x3 / 1x1 kernels, stride 1
or 2, dilation 1, symmetric padding, and the stride-2 / padding-0 transposed form."""
import torch

from ... import kernels

enabled = False                      # kept for interface compatibility (conv2d_gradfix.py:22)
weight_gradients_disabled = False


def _one(v):
    if isinstance(v, (list, tuple)):
        if len(set(v)) != 1:
            raise NotImplementedError(f'anisotropic stride/padding/dilation {v} is not supported')
        return int(v[0])
    return int(v)


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    stride, padding, dilation = _one(stride), _one(padding), _one(dilation)
    if dilation != 1 or stride not in (1, 2):
        raise NotImplementedError('conv2d: only dilation 1 and stride 1/2 are implemented in HIP')
    n, c, h, w = input.shape
    pw = kernels.conv_weight_prep(weight, groups=groups)
    x = input.reshape(n * groups, c // groups, h, w)
    y = kernels.conv2d(x, pw, mode=kernels.MODE_SAME if stride == 1 else kernels.MODE_DOWN2, pad=padding,
                       bias=(bias if groups == 1 else None))
    y = y.reshape(n, -1, *y.shape[2:])
    if bias is not None and groups != 1:
        y = kernels.bias_act(y, bias=bias, act=False)
    return y


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    stride, padding, dilation = _one(stride), _one(padding), _one(dilation)
    if stride != 2 or dilation != 1 or _one(output_padding) != 0 or tuple(weight.shape[2:]) != (3, 3):
        raise NotImplementedError('conv_transpose2d: only the stride-2 3x3 form is implemented in HIP')
    n, c, h, w = input.shape
    ci_g, co_g = weight.shape[0] // groups, weight.shape[1]
    # torch layout [Cin, Cout/g, kh, kw] -> per group [Cout/g, Cin/g, kh, kw]
    wg = weight.reshape(groups, ci_g, co_g, 3, 3).transpose(1, 2).reshape(groups * co_g, ci_g, 3, 3).contiguous()
    pw = kernels.conv_weight_prep(wg, groups=groups)
    y = kernels.conv2d(input.reshape(n * groups, ci_g, h, w), pw, mode=kernels.MODE_UP2T, bias=(bias if groups == 1 else None))
    y = y.reshape(n, -1, *y.shape[2:])
    if padding:
        y = y[:, :, padding:y.shape[2] - padding, padding:y.shape[3] - padding].contiguous()
    if bias is not None and groups != 1:
        y = kernels.bias_act(y, bias=bias, act=False)
    return y
