"""CPU: the plain-C oracle (oracle/native_oracle.c) against the reference-generated golden vectors and the
torch oracle -- a second, independent formulation of upfirdn2d (gather, as in the reference's CUDA kernel)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, rel_err
from oracle import shgan_oracle as orc


@pytest.fixture(scope='module')
def clib():
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])
    lib = ctypes.CDLL(os.path.join(ROOT, 'oracle', '_build', 'liboracle.so'))
    lib.orc_upfirdn2d_f32.restype = ctypes.c_int
    lib.orc_sampler_indices.restype = ctypes.c_int
    return lib


def fp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_c_upfirdn2d_against_reference_golden(clib):
    g = load_golden('upfirdn2d')
    for name in g['names']:
        f = g[f'{name}__f']
        if f.size == 0 or f.ndim != 2:
            continue                      # identity / separable cases go through the Python helpers only
        x = np.ascontiguousarray(g[f'{name}__x'])
        upx, upy, dnx, dny, px0, px1, py0, py1, flip = [int(v) for v in g[f'{name}__cfg']]
        ref = g[f'{name}__y']
        y = np.zeros(ref.shape, np.float32)
        n, c, h, w = x.shape
        rc = clib.orc_upfirdn2d_f32(fp(x), fp(np.ascontiguousarray(f)), fp(y), n, c, h, w, f.shape[0], f.shape[1], upx, upy, dnx, dny,
                                    px0, px1, py0, py1, flip, ctypes.c_float(float(g[f'{name}__gain'])))
        assert rc == 0 and rel_err(y, ref) < 2e-6, name


def test_c_composite_and_sampler(clib):
    x, z, real_u8, mask = orc.synthetic_batch(2, 64, 8, seed=9)
    img = torch.from_numpy((np.random.RandomState(1).standard_normal((2, 3, 64, 64)) * 0.8).astype(np.float32))
    ref = orc.composite_u8(x, img).numpy()
    out = np.zeros_like(ref)
    clib.orc_composite_u8(fp(np.ascontiguousarray(x.numpy())), fp(np.ascontiguousarray(img.numpy())), fp(out), 2, 64 * 64)
    assert np.array_equal(out, ref)
    g = load_golden('integer_paths')
    buf = (ctypes.c_int64 * 4096)()
    for row in g['sampler_rows']:
        n_items, world, rank = [int(v) for v in row[:3]]
        cnt = clib.orc_sampler_indices(n_items, world, rank, 1, buf, 4096)
        assert [int(buf[i]) for i in range(cnt)] == [int(v) for v in row[3:]]
