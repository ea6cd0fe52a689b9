"""GPU parity tests (run with -m gpu on an MI355X): every HIP kernel, called through the C-ABI via the
host ops that mirror the reference's Python op API, against (a) golden vectors produced by the reference
and (b) the CPU oracle on fresh seeded inputs.  Tolerance: BASELINE.json north_star = 1e-3 relative fp32;
the ops land at 1e-6..1e-5 and the tests assert a tighter 1e-4 (1e-5 where the arithmetic is identical);
integer / index outputs are bit-exact."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'
TOL = 1e-4


@pytest.fixture(scope='module')
def mods():
    import shgan_amd  # noqa: F401
    from shgan_amd import kernels
    from shgan_amd.model_zoo import stylegan
    from shgan_amd.model_zoo.common import utils
    from shgan_amd.model_zoo.stylegan_utils import conv2d_gradfix, conv2d_resample, fma, upfirdn2d
    from oracle import shgan_oracle as orc
    assert torch.cuda.is_available()
    return dict(kernels=kernels, stylegan=stylegan, utils=utils, c2r=conv2d_resample, ufd=upfirdn2d, fma=fma,
                gradfix=conv2d_gradfix, orc=orc)


def g(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def c(a):
    return a.detach().cpu().numpy()


def test_device_is_gfx950(mods):
    import ctypes
    from shgan_amd import _lib
    buf = ctypes.create_string_buffer(128)
    cus = _lib.get_lib().shg_device_info(0, buf, 128)
    assert cus > 0 and buf.value.decode().startswith('gfx950'), (cus, buf.value)


def test_upfirdn2d_golden(mods):
    gd = load_golden('upfirdn2d')
    ufd = mods['ufd']
    for name in gd['names']:
        f = gd[f'{name}__f']
        f = g(f) if f.size else None
        upx, upy, dnx, dny, px0, px1, py0, py1, flip = [int(v) for v in gd[f'{name}__cfg']]
        y = ufd.upfirdn2d(g(gd[f'{name}__x']), f, up=[upx, upy], down=[dnx, dny], padding=[px0, px1, py0, py1],
                          flip_filter=bool(flip), gain=float(gd[f'{name}__gain']))
        ref = gd[f'{name}__y']
        assert tuple(y.shape) == ref.shape, name
        assert rel_err(c(y), ref) < 1e-5, name
    x, f = g(gd['helpers__x']), g(gd['helpers__f'])
    assert rel_err(c(ufd.upsample2d(x, f)), gd['helpers__up']) < 1e-5
    assert rel_err(c(ufd.downsample2d(x, f)), gd['helpers__down']) < 1e-5
    assert rel_err(c(ufd.filter2d(x, f)), gd['helpers__filt']) < 1e-5


@pytest.mark.parametrize('shape,pad,gain', [((2, 5, 65, 65), [1, 1, 1, 1], 4.0), ((3, 7, 64, 64), [2, 2, 2, 2], 1.0),
                                            ((1, 3, 130, 67), [2, 1, 1, 2], 1.0), ((2, 2, 5, 5), [1, 1, 1, 1], 4.0)])
def test_fir_fast_path_vs_oracle(mods, shape, pad, gain):
    orc, ufd = mods['orc'], mods['ufd']
    rs = np.random.RandomState(1)
    x = torch.from_numpy(rs.standard_normal(shape).astype(np.float32))
    f = orc.setup_filter([1, 3, 3, 1])
    ref = orc.upfirdn2d(x, f, padding=pad, gain=gain)
    y = ufd.upfirdn2d(x.to(DEV), f.to(DEV), padding=pad, gain=gain)
    assert rel_err(c(y), ref.numpy()) < 1e-5
    fa = torch.from_numpy(rs.standard_normal((4, 4)).astype(np.float32))      # asymmetric: catches flips
    for flip in (False, True):
        ref = orc.upfirdn2d(x, fa, padding=pad, gain=gain, flip_filter=flip)
        y = ufd.upfirdn2d(x.to(DEV), fa.to(DEV), padding=pad, gain=gain, flip_filter=flip)
        assert rel_err(c(y), ref.numpy()) < 1e-5


@pytest.mark.parametrize('shape', [(2, 3, 512, 512), (1, 5, 256, 256), (3, 2, 128, 128), (2, 7, 64, 64), (5, 3, 32, 32), (3, 5, 16, 16),
                                   (1, 9, 8, 8), (2, 3, 64, 128)])
def test_fir_march_pad2_vs_oracle_and_tiled_kernel(mods, shape):
    """Row-marching pad-2 FIR (csrc/fir_march.h, separable filters): plain rows and polyphase planes, against the oracle
    (upfirdn2d.py:98-138 semantics) and against the tiled kernel it replaces; asymmetric separable taps catch flips; a filter
    that is not an outer product must keep the general kernel."""
    orc, ufd, K = mods['orc'], mods['ufd'], mods['kernels']
    rs = np.random.RandomState(7)
    x = torch.from_numpy(rs.standard_normal(shape).astype(np.float32))
    n, ch, h, w = shape
    lib = K._lib.get_lib()
    assert lib.shg_fir_pad2_sep_supported(h, w, 0) == 1
    fa = torch.from_numpy(np.outer([0.5, 1.5, -0.7, 0.2], [0.3, 1.0, 2.0, -0.4]).astype(np.float32))
    for f in (orc.setup_filter([1, 3, 3, 1]), fa):
        assert K.sep_taps(f.to(DEV)) is not None
        for flip in (False, True):
            ref = orc.upfirdn2d(x, f, padding=[2, 2, 2, 2], gain=1.5, flip_filter=flip).numpy()
            fd = f.to(DEV)
            y = c(ufd.upfirdn2d(x.to(DEV), fd, padding=[2, 2, 2, 2], gain=1.5, flip_filter=flip))
            assert rel_err(y, ref) < 1e-5
            K.FIR_MARCH = False
            try:
                y_old = c(ufd.upfirdn2d(x.to(DEV), f.to(DEV), padding=[2, 2, 2, 2], gain=1.5, flip_filter=flip))
            finally:
                K.FIR_MARCH = True
            assert rel_err(y, y_old) < 1e-6
            if h % 2 == 0 and w % 2 == 0:
                assert lib.shg_fir_pad2_sep_supported(h, w, (w // 2 + 1 + 3) // 4 * 4) == 1
                for pp in ((w // 2 + 1 + 3) // 4 * 4, (w // 2 + 1 + 31) // 32 * 32):
                    xp = torch.full((4, n, ch, h // 2 + 1, pp), float('nan'), device=DEV)
                    xd = x.to(DEV)                 # (kept alive across the raw C call)
                    K.check(lib.shg_fir_pad2_sep_f32(K._ptr(xd), K.sep_taps(fd), K._ptr(xp), n, ch, h, w, pp, int(flip), 1.5,
                                                     None), 'fir_pad2_sep')
                    full = np.zeros((n, ch, 2 * (h // 2 + 1), 2 * pp), np.float32)
                    full[:, :, :h + 1, :w + 1] = ref
                    got = c(xp)
                    for a in range(2):
                        for b in range(2):
                            assert np.array_equal(np.isnan(got[a * 2 + b]), np.zeros_like(got[a * 2 + b], bool))
                            assert np.abs(got[a * 2 + b] - full[:, :, a::2, b::2]).max() < 1e-5 * max(1.0, np.abs(ref).max())
                            pad = full[:, :, a::2, b::2] == 0
                            edge = np.zeros_like(pad); edge[:, :, :, (w + 1 - b + 1) // 2:] = True; edge[:, :, (h + 1 - a + 1) // 2:, :] = True
                            assert (got[a * 2 + b][edge] == 0).all()           # beyond the filtered image: exact zeros
    fn = torch.from_numpy(rs.standard_normal((4, 4)).astype(np.float32))
    assert K.sep_taps(fn.to(DEV)) is None
    ref = orc.upfirdn2d(x, fn, padding=[2, 2, 2, 2]).numpy()
    assert rel_err(c(ufd.upfirdn2d(x.to(DEV), fn.to(DEV), padding=[2, 2, 2, 2])), ref) < 1e-5


@pytest.mark.parametrize('shape', [(2, 3, 512, 512), (1, 5, 256, 256), (3, 2, 128, 128), (2, 7, 64, 64), (5, 3, 32, 32), (3, 5, 16, 16),
                                   (2, 9, 8, 8), (2, 3, 64, 128)])
def test_fir_march_x2_resampling_vs_oracle_and_generic_kernel(mods, shape):
    """Row-marching down=2 (padding 1) and up=2 (padding [2,1,2,1]) FIRs of the training rows: oracle (upfirdn2d.py:98-138) and the
    generic gather kernel, symmetric and asymmetric separable taps, both flips, a gain."""
    orc, ufd, K = mods['orc'], mods['ufd'], mods['kernels']
    rs = np.random.RandomState(11)
    n, ch, h, w = shape
    lib = K._lib.get_lib()
    fa = torch.from_numpy(np.outer([0.5, 1.5, -0.7, 0.2], [0.3, 1.0, 2.0, -0.4]).astype(np.float32))
    for up, xs in ((1, (n, ch, h, w)), (2, (n, ch, h // 2, w // 2))):
        if not lib.shg_fir_resample2_sep_supported(xs[2], xs[3], up):
            assert up == 2 and xs[3] < 4
            continue
        x = torch.from_numpy(rs.standard_normal(xs).astype(np.float32))
        kw = dict(down=2, padding=[1, 1, 1, 1]) if up == 1 else dict(up=2, padding=[2, 1, 2, 1])
        for f in (orc.setup_filter([1, 3, 3, 1]), fa):
            for flip in (False, True):
                ref = orc.upfirdn2d(x, f, gain=1.7, flip_filter=flip, **kw).numpy()
                y = c(ufd.upfirdn2d(x.to(DEV), f.to(DEV), gain=1.7, flip_filter=flip, **kw))
                assert y.shape == ref.shape
                assert rel_err(y, ref) < 1e-5
                K.FIR_MARCH = False
                try:
                    y_old = c(ufd.upfirdn2d(x.to(DEV), f.to(DEV), gain=1.7, flip_filter=flip, **kw))
                finally:
                    K.FIR_MARCH = True
                assert rel_err(y, y_old) < 2e-6


def test_fir_fused_epilogue_vs_oracle(mods):
    orc, k = mods['orc'], mods['kernels']
    rs = np.random.RandomState(2)
    n, ch, r = 3, 6, 16
    mid = torch.from_numpy(rs.standard_normal((n, ch, 2 * r + 1, 2 * r + 1)).astype(np.float32))
    f = orc.setup_filter([1, 3, 3, 1])
    scale = torch.from_numpy(rs.rand(n, ch).astype(np.float32) + 0.5)
    bias = torch.from_numpy(rs.standard_normal(ch).astype(np.float32))
    res = torch.from_numpy(rs.standard_normal((n, ch, 2 * r, 2 * r)).astype(np.float32))
    for noise in (torch.from_numpy(rs.standard_normal((2 * r, 2 * r)).astype(np.float32)),
                  torch.from_numpy(rs.standard_normal((n, 1, 2 * r, 2 * r)).astype(np.float32)), None):
        ref = orc.upfirdn2d(mid, f, padding=[1, 1, 1, 1], gain=4.0) * scale[:, :, None, None]
        if noise is not None:
            ref = ref + noise * 0.3
        ref = orc.lrelu_agc(ref + bias.view(1, -1, 1, 1), gain=0.7) + res
        ep = dict(scale=scale.reshape(-1).to(DEV), bias=bias.to(DEV), noise=None if noise is None else noise.to(DEV),
                  noise_strength=0.3, residual=res.to(DEV), act=True, gain=0.7)
        y = k.upfirdn2d(mid.to(DEV), f.to(DEV), padx0=1, padx1=1, pady0=1, pady1=1, gain=4.0, epilogue=ep)
        assert rel_err(c(y), ref.numpy()) < 1e-5


def test_conv2d_resample_golden(mods):
    gd = load_golden('conv2d_resample')
    f4 = g(gd['f'])
    for name in gd['names']:
        up, down, pad, groups, flipw, hasf = [int(v) for v in gd[f'{name}__cfg']]
        y = mods['c2r'].conv2d_resample(x=g(gd[f'{name}__x']), w=g(gd[f'{name}__w']), f=(f4 if hasf else None), up=up,
                                        down=down, padding=pad, groups=groups, flip_weight=bool(flipw))
        assert tuple(y.shape) == gd[f'{name}__y'].shape, name
        assert rel_err(c(y), gd[f'{name}__y']) < 1e-5, name


CONV_CASES = [
    # n, ci, co, h, w, k, mode(0 same,1 down2,2 up2T), pad
    (2, 6, 5, 12, 12, 3, 0, 1), (1, 64, 64, 32, 32, 3, 0, 1), (2, 70, 130, 20, 36, 3, 0, 1), (3, 8, 200, 8, 8, 3, 0, 1),
    (5, 16, 16, 4, 4, 3, 0, 1), (2, 33, 65, 17, 17, 3, 1, 0), (2, 16, 24, 33, 33, 3, 1, 0), (2, 12, 20, 8, 8, 3, 2, 0),
    (1, 64, 128, 16, 16, 3, 2, 0), (3, 9, 7, 5, 7, 3, 2, 0), (2, 64, 384, 66, 32, 1, 0, 0), (2, 10, 3, 9, 9, 1, 0, 0),
    (1, 128, 64, 64, 64, 3, 0, 1),
    # small spatial grids with deep K -> split-K path (partials + fused reduce/epilogue)
    (16, 256, 256, 4, 4, 3, 0, 1), (16, 512, 512, 8, 8, 3, 0, 1), (4, 192, 130, 16, 16, 3, 0, 1), (8, 128, 128, 17, 17, 3, 1, 0),
    # all-phase transposed conv at sizes that cross tile borders
    (2, 40, 70, 33, 35, 3, 2, 0), (16, 64, 64, 4, 4, 3, 2, 0),
]


@pytest.mark.parametrize('n,ci,co,h,w,k,mode,pad', CONV_CASES)
def test_mfma_conv_vs_torch_cpu(mods, n, ci, co, h, w, k, mode, pad):
    """Transpose-detecting check of the MFMA fragment layout: random asymmetric weights, odd sizes."""
    import torch.nn.functional as F
    kk = mods['kernels']
    rs = np.random.RandomState(n * 1000 + ci + co)
    x = torch.from_numpy(rs.standard_normal((n, ci, h, w)).astype(np.float32))
    wt = torch.from_numpy(rs.standard_normal((co, ci, k, k)).astype(np.float32))
    if mode == 0:
        ref = F.conv2d(x, wt, padding=pad)
    elif mode == 1:
        ref = F.conv2d(x, wt, stride=2, padding=pad)
    else:
        ref = F.conv_transpose2d(x, wt.transpose(0, 1), stride=2)
    pw = kk.conv_weight_prep(wt.to(DEV))
    y = kk.conv2d(x.to(DEV), pw, mode=mode, pad=pad)
    assert tuple(y.shape) == tuple(ref.shape)
    assert rel_err(c(y), ref.numpy()) < 2e-5


@pytest.mark.parametrize('n,ci,co,h,w', [(8, 64, 128, 128, 128), (8, 128, 64, 128, 128), (16, 256, 512, 64, 64), (2, 512, 512, 128, 64)])
def test_conv1x1_gemm_form_vs_torch_cpu(mods, n, ci, co, h, w):
    """The 1x1 layers on the GEMM kernel of round 6 (conv1x1_gemm_kernel: both operands by LDS-DMA; taken for whole 128 x 128 or 64 x 256
    tiles that fill the chip -- the critic's skip branches and their input gradients): plain, with the whole store-pass tail (per-sample
    output scale, bias, lrelu_agc, gain, skip), and bit-identical to the tap-list kernel's arithmetic order is NOT claimed (another K order):
    both against torch CPU."""
    import torch.nn.functional as F
    kk, orc = mods['kernels'], mods['orc']
    rs = np.random.RandomState(n + ci + co + h)
    x = torch.from_numpy(rs.standard_normal((n, ci, h, w)).astype(np.float32))
    wt = torch.from_numpy(rs.standard_normal((co, ci, 1, 1)).astype(np.float32))
    s_out = torch.from_numpy(rs.rand(n, co).astype(np.float32) + 0.5)
    bias = torch.from_numpy(rs.standard_normal(co).astype(np.float32))
    res = torch.from_numpy(rs.standard_normal((n, co, h, w)).astype(np.float32))
    pw = kk.conv_weight_prep(wt.to(DEV), gain=0.1)
    y = kk.conv2d(x.to(DEV), pw, mode=0, pad=0, gain=0.5)
    assert rel_err(c(y), (F.conv2d(x, wt * 0.1) * 0.5).numpy()) < 2e-5
    y = kk.conv2d(x.to(DEV), pw, mode=0, pad=0, out_scale=s_out.to(DEV), bias=bias.to(DEV), act=True, gain=0.5, residual=res.to(DEV))
    ref = orc.lrelu_agc(F.conv2d(x, wt * 0.1) * s_out[:, :, None, None] + bias.view(1, -1, 1, 1), gain=0.5) + res
    assert rel_err(c(y), ref.numpy()) < 2e-5
    # a view that is not 16-byte aligned falls back to the tap-list kernel: same answer
    xo = torch.zeros(x.numel() + 1, device=DEV)[1:].view_as(x).copy_(x.to(DEV))
    assert xo.data_ptr() % 16 != 0
    assert rel_err(c(kk.conv2d(xo, pw, mode=0, pad=0, gain=0.5)), (F.conv2d(x, wt * 0.1) * 0.5).numpy()) < 2e-5
    ro = torch.zeros(res.numel() + 1, device=DEV)[1:].view_as(res).copy_(res.to(DEV))          # ... and so does an unaligned skip tensor
    y = kk.conv2d(x.to(DEV), pw, mode=0, pad=0, gain=0.5, residual=ro)
    assert rel_err(c(y), (F.conv2d(x, wt * 0.1) * 0.5 + res).numpy()) < 2e-5


WINO_CASES = [
    # n, ci, co, h, w : ragged channel counts (I % 8, O % 64), odd extents, several tiles per image, one-chunk K
    (2, 64, 64, 32, 32), (1, 13, 70, 33, 36), (3, 8, 3, 40, 64), (2, 72, 130, 35, 68), (1, 128, 64, 64, 96), (2, 5, 5, 32, 44),
    (2, 64, 64, 16, 16), (1, 24, 70, 20, 24), (3, 16, 130, 17, 28),      # 16 x 16 pixel tiles (images narrower than 32)
    (1, 16, 16, 32, 33),      # W % 4 != 0: the wrapper must fall back to the direct kernel
    (4, 512, 512, 16, 16), (2, 256, 200, 20, 24), (1, 384, 64, 16, 28),      # small grids: the launch splits along the input channels (8 / 4 / 4 slices)
]


@pytest.mark.parametrize('n,ci,co,h,w', WINO_CASES)
def test_wino_conv_vs_torch_cpu_and_direct(mods, n, ci, co, h, w):
    """Winograd F(2x2,3x3) kernel (shg_conv2d_wino_f32) vs torch CPU conv2d and vs the direct MFMA kernel, with the
    whole fused tail (styles, demodulation coefficient, per-sample noise, bias, lrelu_agc, skip) and true-convolution
    (flipped) weights."""
    import torch.nn.functional as F
    kk, orc = mods['kernels'], mods['orc']
    rs = np.random.RandomState(n * 100 + ci + co + h)
    x = torch.from_numpy(rs.standard_normal((n, ci, h, w)).astype(np.float32))
    wt = torch.from_numpy(rs.standard_normal((co, ci, 3, 3)).astype(np.float32))
    s_in = torch.from_numpy(rs.rand(n, ci).astype(np.float32) + 0.5)
    s_out = torch.from_numpy(rs.rand(n, co).astype(np.float32) + 0.5)
    bias = torch.from_numpy(rs.standard_normal(co).astype(np.float32))
    noise = torch.from_numpy(rs.standard_normal((n, 1, h, w)).astype(np.float32))
    res = torch.from_numpy(rs.standard_normal((n, co, h, w)).astype(np.float32))
    for flip in (False, True):
        wref = wt.flip([2, 3]) if flip else wt
        ref = F.conv2d(x * s_in[:, :, None, None], wref * 0.1, padding=1) * s_out[:, :, None, None] + noise * 0.25
        ref = orc.lrelu_agc(ref + bias.view(1, -1, 1, 1), gain=0.5) + res
        args = dict(mode=0, pad=1, in_scale=s_in.to(DEV), out_scale=s_out.to(DEV), bias=bias.to(DEV), noise=noise.to(DEV),
                    noise_strength=0.25, act=True, gain=0.5, residual=res.to(DEV))
        old = kk.WINO, kk.WINO4
        try:
            kk.WINO, kk.WINO4 = True, False
            pw = kk.conv_weight_prep(wt.to(DEV), gain=0.1, flip=flip)
            y = kk.conv2d(x.to(DEV), pw, **args)
            assert (pw.wu is not None) == (w % 4 == 0), 'Winograd dispatch: taken iff the 16-byte window DMA applies'
            assert pw.wu4 is None
            kk.WINO = False
            yd = kk.conv2d(x.to(DEV), kk.conv_weight_prep(wt.to(DEV), gain=0.1, flip=flip), **args)
        finally:
            kk.WINO, kk.WINO4 = old
        assert rel_err(c(y), ref.numpy()) < 2e-5
        assert rel_err(c(y), c(yd)) < 2e-5
    # plain form: no fused operands, linear output
    old = kk.WINO, kk.WINO4
    try:
        kk.WINO, kk.WINO4 = True, False
        y = kk.conv2d(x.to(DEV), kk.conv_weight_prep(wt.to(DEV)), mode=0, pad=1)
    finally:
        kk.WINO, kk.WINO4 = old
    assert rel_err(c(y), F.conv2d(x, wt, padding=1).numpy()) < 2e-5


WINO4_CASES = [
    # n, ci, co, h, w : ragged channel counts (I % 8, O % 64, O % 32), extents that are no multiple of the 16 x 32 tile or of 4
    (2, 64, 64, 32, 32), (1, 13, 70, 33, 36), (3, 8, 3, 40, 64), (2, 72, 130, 35, 68), (1, 128, 64, 64, 96), (2, 5, 5, 32, 44),
    (1, 512, 96, 48, 32),     # 64 K-chunks: the style table spans a whole wave
    (1, 16, 16, 32, 33), (2, 16, 16, 16, 32),      # W % 4 != 0 -> direct kernel; fewer than WINO4_MIN rows -> F(2x2,3x3)
    (1, 24, 70, 21, 132), (2, 16, 64, 32, 128), (1, 8, 8, 40, 200),      # W >= 128: the 8 x 64 tile shape, ragged in both directions
    (1, 16, 70, 18, 260), (2, 8, 64, 16, 256),                           # W >= 256: the 4 x 128 tile shape
    (4, 512, 512, 32, 32), (2, 256, 100, 36, 64), (1, 200, 64, 32, 32),  # small grids: split along the input channels (ragged last slice: 25 chunks)
]


@pytest.mark.parametrize('n,ci,co,h,w', WINO4_CASES)
def test_wino4_conv_vs_torch_cpu_and_direct(mods, n, ci, co, h, w):
    """Winograd F(4x4,3x3) kernel (shg_conv2d_wino4_f32) vs torch CPU conv2d (fp64) and vs the direct MFMA kernel, with the
    whole fused tail and flipped weights.  Tolerance 1e-4 of the output range: the 4x4 transform's constants cost about one
    decimal digit against the direct form (measured 3e-6..1.2e-5)."""
    import torch.nn.functional as F
    kk, orc = mods['kernels'], mods['orc']
    rs = np.random.RandomState(n * 100 + ci + co + h)
    x = torch.from_numpy(rs.standard_normal((n, ci, h, w)).astype(np.float32))
    wt = torch.from_numpy(rs.standard_normal((co, ci, 3, 3)).astype(np.float32))
    s_in = torch.from_numpy(rs.rand(n, ci).astype(np.float32) + 0.5)
    s_out = torch.from_numpy(rs.rand(n, co).astype(np.float32) + 0.5)
    bias = torch.from_numpy(rs.standard_normal(co).astype(np.float32))
    noise = torch.from_numpy(rs.standard_normal((n, 1, h, w)).astype(np.float32))
    res = torch.from_numpy(rs.standard_normal((n, co, h, w)).astype(np.float32))
    served = w % 4 == 0 and h >= kk.WINO4_MIN
    for flip in (False, True):
        wref = wt.flip([2, 3]) if flip else wt
        ref = F.conv2d((x * s_in[:, :, None, None]).double(), wref.double() * 0.1, padding=1) * s_out[:, :, None, None].double() + noise.double() * 0.25
        ref = orc.lrelu_agc((ref + bias.view(1, -1, 1, 1).double()).float(), gain=0.5) + res
        args = dict(mode=0, pad=1, in_scale=s_in.to(DEV), out_scale=s_out.to(DEV), bias=bias.to(DEV), noise=noise.to(DEV),
                    noise_strength=0.25, act=True, gain=0.5, residual=res.to(DEV))
        old = kk.WINO, kk.WINO4
        try:
            kk.WINO, kk.WINO4 = True, True
            pw = kk.conv_weight_prep(wt.to(DEV), gain=0.1, flip=flip)
            y = kk.conv2d(x.to(DEV), pw, **args)
            assert (pw.wu4 is not None) == served, 'F(4x4) dispatch'
            kk.WINO = False
            yd = kk.conv2d(x.to(DEV), kk.conv_weight_prep(wt.to(DEV), gain=0.1, flip=flip), **args)
        finally:
            kk.WINO, kk.WINO4 = old
        e_ref, e_dir = rel_err(c(y), ref.numpy()), rel_err(c(y), c(yd))
        print(f'wino4 n{n} i{ci} o{co} {h}x{w} flip{int(flip)}: vs fp64 {e_ref:.2e} vs direct {e_dir:.2e}')
        assert e_ref < 1e-4 and e_dir < 1e-4
        if served:
            assert not torch.equal(y, yd)
    # plain form: no fused operands, linear output; the batch handled image by image gives the same bits
    old = kk.WINO, kk.WINO4
    try:
        kk.WINO, kk.WINO4 = True, True
        pw = kk.conv_weight_prep(wt.to(DEV))
        y = kk.conv2d(x.to(DEV), pw, mode=0, pad=1)
        y0 = kk.conv2d(x[:1].to(DEV), pw, mode=0, pad=1)
    finally:
        kk.WINO, kk.WINO4 = old
    assert rel_err(c(y), F.conv2d(x.double(), wt.double(), padding=1).numpy()) < 1e-4
    from shgan_amd import _lib
    wsb = _lib.get_lib().shg_conv2d_wino4_workspace_bytes
    if served and (wsb(n, ci, co, pw.op, h, w) or wsb(1, ci, co, pw.op, h, w)):
        # small grids split along the input channels, and the number of slices follows the batch: same sums in another order
        assert rel_err(c(y[:1]), c(y0)) < 1e-5
    else:
        assert torch.equal(y[:1], y0)


def test_mfma_conv_fused_epilogue(mods):
    import torch.nn.functional as F
    kk, orc = mods['kernels'], mods['orc']
    rs = np.random.RandomState(5)
    n, ci, co, r = 3, 24, 40, 16
    x = torch.from_numpy(rs.standard_normal((n, ci, r, r)).astype(np.float32))
    wt = torch.from_numpy(rs.standard_normal((co, ci, 3, 3)).astype(np.float32))
    s_in = torch.from_numpy(rs.rand(n, ci).astype(np.float32) + 0.5)
    s_out = torch.from_numpy(rs.rand(n, co).astype(np.float32) + 0.5)
    bias = torch.from_numpy(rs.standard_normal(co).astype(np.float32))
    noise = torch.from_numpy(rs.standard_normal((n, 1, r, r)).astype(np.float32))
    res = torch.from_numpy(rs.standard_normal((n, co, r, r)).astype(np.float32))
    ref = F.conv2d(x * s_in[:, :, None, None], wt * 0.1, padding=1) * s_out[:, :, None, None] + noise * 0.25
    ref = orc.lrelu_agc(ref + bias.view(1, -1, 1, 1), gain=0.5) + res
    pw = kk.conv_weight_prep(wt.to(DEV), gain=0.1)
    y = kk.conv2d(x.to(DEV), pw, mode=0, pad=1, in_scale=s_in.to(DEV), out_scale=s_out.to(DEV), bias=bias.to(DEV),
                  noise=noise.to(DEV), noise_strength=0.25, act=True, gain=0.5, residual=res.to(DEV))
    assert rel_err(c(y), ref.numpy()) < 2e-5


def test_up_layer_planar_path_vs_oracle(mods):
    """convT (all phases, planar) -> FIR-from-planes with the fused tail == oracle convT -> FIR -> noise/bias/act/skip."""
    kk, orc = mods['kernels'], mods['orc']
    rs = np.random.RandomState(21)
    for n, ci, co, r in [(2, 12, 20, 8), (3, 70, 66, 16), (1, 32, 64, 33)]:
        x = torch.from_numpy(rs.standard_normal((n, ci, r, r + 1)).astype(np.float32))
        w = torch.from_numpy(rs.standard_normal((co, ci, 3, 3)).astype(np.float32))
        f = torch.from_numpy(rs.rand(4, 4).astype(np.float32))                      # asymmetric filter
        s_in = torch.from_numpy(rs.rand(n, ci).astype(np.float32) + 0.5)
        s_out = torch.from_numpy(rs.rand(n, co).astype(np.float32) + 0.5)
        bias = torch.from_numpy(rs.standard_normal(co).astype(np.float32))
        noise = torch.from_numpy(rs.standard_normal((n, 1, 2 * r, 2 * r + 2)).astype(np.float32))
        res = torch.from_numpy(rs.standard_normal((n, co, 2 * r, 2 * r + 2)).astype(np.float32))
        ref = orc.conv2d_resample(x * s_in[:, :, None, None], w, f=f, up=2, padding=1, flip_weight=False)
        ref = orc.lrelu_agc(ref * s_out[:, :, None, None] + noise * 0.4 + bias.view(1, -1, 1, 1), gain=0.8) + res
        pw = kk.conv_weight_prep(w.to(DEV))
        mid = kk.conv2d(x.to(DEV), pw, mode=kk.MODE_UP2T, in_scale=s_in.to(DEV), planar=True)
        y = kk.upfir_planar(mid, f.to(DEV), scale=s_out.reshape(-1).to(DEV), bias=bias.to(DEV), noise=noise.to(DEV),
                            noise_strength=0.4, residual=res.to(DEV), act=True, gain=0.8)
        assert tuple(y.shape) == tuple(ref.shape)
        assert rel_err(c(y), ref.numpy()) < 2e-5, (n, ci, co, r)


@pytest.mark.parametrize('n,co,h,w', [(2, 5, 256, 256), (1, 9, 128, 128), (3, 7, 64, 64), (2, 6, 32, 32), (5, 4, 16, 16), (3, 3, 8, 8),
                                       (2, 5, 4, 4), (2, 3, 16, 64)])
def test_upfir_march_vs_tiled_kernel_and_reference_math(mods, n, co, h, w):
    """Row-marching FIR-from-phase-planes (csrc/fir_march.h): separable filters, every plane-width class, all tail operands and
    subsets of them, against the tiled kernel (validated against the oracle above) and against a float64 restatement of
    conv2d_resample.py:133-138 + stylegan.py:295-304 on the interleaved planes."""
    kk = mods['kernels']
    rs = np.random.RandomState(33)
    mid = torch.from_numpy(rs.standard_normal((4, n, co, h + 1, w + 1)).astype(np.float32))
    scale = torch.from_numpy(rs.rand(n * co).astype(np.float32) + 0.5)
    bias = torch.from_numpy(rs.standard_normal(co).astype(np.float32))
    noise = torch.from_numpy(rs.standard_normal((n, 1, 2 * h, 2 * w)).astype(np.float32))
    res = torch.from_numpy(rs.standard_normal((n, co, 2 * h, 2 * w)).astype(np.float32))
    assert kk._lib.get_lib().shg_upfir_planar_sep_supported(h, w) == 1
    fsep = torch.from_numpy(np.outer([0.5, 1.5, -0.7, 0.2], [0.3, 1.0, 2.0, -0.4]).astype(np.float32))
    f1331 = mods['orc'].setup_filter([1, 3, 3, 1])
    # float64 reference from the interleaved (2H+1) x (2W+1) image
    full = np.zeros((n, co, 2 * h + 4, 2 * w + 4))
    m = mid.numpy().astype(np.float64)
    for a in range(2):
        for b in range(2):
            full[:, :, 1 + a:1 + a + 2 * (h + 1 - a):2, 1 + b:1 + b + 2 * (w + 1 - b):2] = m[a * 2 + b][:, :, :h + 1 - a, :w + 1 - b]
    for f, flip, kw in [(f1331, False, dict(scale=scale, bias=bias, noise=noise, residual=res, act=True)),
                        (fsep, False, dict(scale=scale, bias=bias, noise=noise, residual=res, act=True)),
                        (fsep, True, dict(scale=scale, noise=noise[0, 0], act=False)),
                        (fsep, False, dict(bias=bias, residual=res, act=True))]:
        args = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in kw.items()}
        y = c(kk.upfir_planar(mid.to(DEV), f.to(DEV), noise_strength=0.4, gain=0.8, flip=flip, **args))
        kk.FIR_MARCH = False
        try:
            y_old = c(kk.upfir_planar(mid.to(DEV), f.to(DEV), noise_strength=0.4, gain=0.8, flip=flip, **args))
        finally:
            kk.FIR_MARCH = True
        assert rel_err(y, y_old) < 2e-6
        fk = f.numpy().astype(np.float64) * 4.0
        fk = fk if flip else fk[::-1, ::-1]
        v = np.zeros((n, co, 2 * h, 2 * w))
        for ky in range(4):
            for kx in range(4):
                v += fk[ky, kx] * full[:, :, ky:ky + 2 * h, kx:kx + 2 * w]
        if 'scale' in kw:
            v = v * scale.numpy().astype(np.float64).reshape(n, co, 1, 1)
        if 'noise' in kw:
            nz = kw['noise'].numpy().astype(np.float64)
            v = v + (nz if nz.ndim == 4 else nz[None, None]) * 0.4
        if 'bias' in kw:
            v = v + bias.numpy().astype(np.float64).reshape(1, co, 1, 1)
        if kw['act']:
            v = np.clip(np.where(v < 0, 0.2 * v, v) * np.sqrt(2) * 0.8, -256 * 0.8, 256 * 0.8)
        if 'residual' in kw:
            v = v + res.numpy().astype(np.float64)
        assert rel_err(y, v) < 2e-6


def test_modulated_conv2d_golden(mods):
    gd = load_golden('modulated_conv2d')
    f4 = g(gd['f'])
    for name in gd['names']:
        up, demod, fused, k = [int(v) for v in gd[f'{name}__cfg']]
        y = mods['stylegan'].modulated_conv2d(
            x=g(gd[f'{name}__x']), weight=g(gd[f'{name}__w']), styles=g(gd[f'{name}__s']), noise=g(gd[f'{name}__noise']),
            up=up, padding=k // 2, resample_filter=(f4 if up > 1 else None), demodulate=bool(demod), flip_weight=(up == 1),
            fused_modconv=bool(fused))
        assert rel_err(c(y), gd[f'{name}__y']) < 2e-5, name


def test_small_ops_golden(mods):
    gd = load_golden('small_ops')
    utils, kk = mods['utils'], mods['kernels']
    x = g(gd['lrelu__x'])
    act = utils.get_unit()('lrelu_agc(alpha=0.2, gain=sqrt_2, clamp=256)')()
    assert np.array_equal(c(act(x.clone())), gd['lrelu__y_gain1'])                 # bit-exact
    assert np.array_equal(c(act(x.clone(), gain=np.sqrt(0.5))), gd['lrelu__y_gain_sqrt_half'])
    act2 = utils.get_unit()('lrelu_agc(alpha=0.1, gain=1)')()
    assert np.array_equal(c(act2(x.clone())), gd['lrelu__y_noclamp'])
    for tag in ('mapping', 'affine', 'fc'):
        lr, use_act = gd[f'dense_{tag}__cfg']
        w = g(gd[f'dense_{tag}__w'])
        y = kk.dense(g(gd[f'dense_{tag}__x']), w, g(gd[f'dense_{tag}__b']), wgain=float(lr) / np.sqrt(w.shape[1]),
                     bgain=float(lr), act=bool(use_act))
        assert rel_err(c(y), gd[f'dense_{tag}__y']) < 1e-5, tag
    y = mods['fma'].fma(g(gd['fma__a']), g(gd['fma__b']), g(gd['fma__c']))
    assert rel_err(c(y), gd['fma__y']) < 1e-6


def test_dense_large_and_normalize(mods):
    kk, orc = mods['kernels'], mods['orc']
    rs = np.random.RandomState(9)
    for n, kdim, o in [(16, 8192, 1024), (32, 1024, 8192), (33, 1536, 512), (1, 7, 3)]:
        x = torch.from_numpy(rs.standard_normal((n, kdim)).astype(np.float32))
        w = torch.from_numpy(rs.standard_normal((o, kdim)).astype(np.float32))
        b = torch.from_numpy(rs.standard_normal(o).astype(np.float32))
        ref = orc.dense(x, w, b, act=True)
        y = kk.dense(x.to(DEV), w.to(DEV), b.to(DEV), wgain=1 / np.sqrt(kdim), act=True)
        assert rel_err(c(y), ref.numpy()) < 1e-5
    z = torch.from_numpy(rs.standard_normal((5, 512)).astype(np.float32))
    ref = z * (z.square().mean(1, keepdim=True) + 1e-8).rsqrt()
    assert rel_err(c(kk.normalize_2nd_moment(z.to(DEV))), ref.numpy()) < 1e-6


def test_grouped_style_kernels(mods):
    """shg_dense_grouped_f32 / shg_modconv_style_prep_grouped_f32 (all style affines of a pass in one launch, input
    cat([w_i, x_global]) read from its two sources, strided output rows) == the per-layer calls == torch CPU."""
    kk = mods['kernels']
    rs = np.random.RandomState(21)
    for n in (3, 16, 19):                      # 19 > DENSE_MAXN: second batch slab
        ws = torch.from_numpy(rs.standard_normal((n, 5, 48)).astype(np.float32)).to(DEV)
        w0 = torch.from_numpy(rs.standard_normal((n, 80)).astype(np.float32)).to(DEV)
        dims = [(64, 96, True), (33, 70, True), (128, 3, False), (7, 200, True)]      # (I, O, demod)
        raw = torch.empty((n, sum(d[0] for d in dims)), device=DEV)
        d_items, p_items, refs = [], [], []
        off = 0
        for li, (i_n, o_n, demod) in enumerate(dims):
            aw = torch.from_numpy(rs.standard_normal((i_n, 128)).astype(np.float32)).to(DEV)
            ab = torch.from_numpy(rs.standard_normal(i_n).astype(np.float32)).to(DEV)
            cw = torch.from_numpy(rs.standard_normal((o_n, i_n, 3, 3)).astype(np.float32)).to(DEV)
            pw = kk.conv_weight_prep(cw, demod=True) if demod else None
            st = raw[:, off:off + i_n]
            off += i_n
            d_items.append(dict(x1=ws[:, li, :], x2=w0, w=aw, b=ab, y=st, wgain=0.3, bgain=1.5))
            s = torch.empty((n, i_n), device=DEV)
            d = torch.empty((n, o_n), device=DEV) if demod else None
            p_items.append(dict(styles=st, pw=pw, demod=demod, pre_gain=1.0 if demod else 0.25, s=s, d=d))
            x = torch.cat([ws[:, li, :], w0], 1)
            st_ref = kk.dense(x, aw, ab, wgain=0.3, bgain=1.5)
            s_ref, d_ref = kk.modconv_style_prep(st_ref, pw, demod=demod, pre_gain=1.0 if demod else 0.25)
            st_cpu = (x.cpu() @ (aw.cpu() * 0.3).t()) + ab.cpu() * 1.5
            refs.append((st, st_ref, st_cpu, s, s_ref, d, d_ref))
        kk.dense_grouped(d_items)
        kk.modconv_style_prep_grouped(p_items)
        for st, st_ref, st_cpu, s, s_ref, d, d_ref in refs:
            assert rel_err(c(st), st_cpu.numpy()) < 1e-5
            assert rel_err(c(st), c(st_ref)) < 1e-6          # (lane partial sums are grouped differently: not bit-equal)
            assert rel_err(c(s), c(s_ref)) < 1e-5
            if d is not None:
                assert rel_err(c(d), c(d_ref)) < 1e-5
    with pytest.raises(RuntimeError):
        kk.dense_grouped([dict(x1=ws[:, 0, :], x2=w0, w=torch.zeros((4, 7), device=DEV), b=None, y=torch.zeros((n, 4), device=DEV))])


def test_shu_golden(mods):
    """rFFT2 / heterogeneous filter / Gaussian split / irFFT2 kernels vs the reference's SHU (N=2)."""
    from shgan_amd.model_zoo import shgan
    gd = load_golden('shu')
    orc = mods['orc']
    sd = orc.init_state_dict(256, seed=int(gd['shu__seed']), ch_base=2048, ch_max=32, w_dim=64, z_dim=64, w0_dim=128,
                             bias_std=0.2)
    shu = shgan.SHU(32, 32, [2, 3], 'piecewise_linear', input_res=64, lowest_res=4, tail_sigma_mult=3)
    shu.load_state_dict({k[len('encoder.shu.'):]: v for k, v in sd.items() if k.startswith('encoder.shu.')}, strict=True)
    shu = shu.to(DEV).eval()
    x = g(gd['shu__x'])
    out = shu(x)
    for r in (4, 8, 16, 32, 64):
        assert rel_err(c(out[r]), gd[f'shu__y{r}']) < TOL, r
    # the same unit on bicubic band weights (shu_df_type='bicubic')
    shu_b = shgan.SHU(32, 32, [2, 3], 'bicubic', input_res=64, lowest_res=4, tail_sigma_mult=3)
    shu_b.load_state_dict(shu.state_dict(), strict=True)
    out_b = shu_b.to(DEV).eval()(x)
    for r in (4, 8, 16, 32, 64):
        assert rel_err(c(out_b[r]), gd[f'shu_bicubic__y{r}']) < TOL, r
        assert rel_err(c(out_b[r]), gd[f'shu__y{r}']) > 1e-3              # (and it is a different filter)
    # spectrum alone against torch.fft on the CPU
    t = c(mods['kernels'].shu_rfft2_shift(x))
    sp = torch.fft.rfftn(torch.from_numpy(gd['shu__x']), dim=(2, 3), norm='forward')
    sp = torch.cat([sp[:, :, 33:], sp[:, :, :33]], dim=2)
    assert rel_err(t[:, :32], sp.real.numpy()) < 1e-5 and rel_err(t[:, 32:], sp.imag.numpy()) < 1e-5
    # fused accumulate form == separate add
    feats = {r: torch.zeros(2, 40, r, r, device=DEV) for r in (4, 8, 16, 32, 64)}
    shu.forward_accumulate(x, feats)
    for r in (4, 8, 16, 32, 64):
        assert rel_err(c(feats[r][:, 8:]), gd[f'shu__y{r}']) < TOL and float(feats[r][:, :8].abs().max()) == 0.0


def test_composite_u8_bit_exact(mods):
    orc, kk = mods['orc'], mods['kernels']
    x, z, real_u8, mask = orc.synthetic_batch(3, 64, 8, seed=3)
    rs = np.random.RandomState(4)
    img = torch.from_numpy((rs.standard_normal((3, 3, 64, 64)) * 0.7).astype(np.float32))
    ref = orc.composite_u8(x, img)
    out = kk.composite_u8(x.to(DEV), img.to(DEV))
    assert out.dtype == torch.uint8 and np.array_equal(c(out), ref.numpy())
    m = mask.astype(bool)
    assert np.array_equal(np.where(m, c(out), 0), np.where(m, real_u8, 0))     # known pixels == the real image


def test_assemble_input_bit_exact(mods):
    """x = cat([mask-0.5, real*mask]) (shgan_default.py:267-274): same fp32 operations as torch -> bit-identical."""
    kk = mods['kernels']
    rs = np.random.RandomState(3)
    for n, h, w in [(3, 16, 20), (2, 64, 64), (1, 2, 2)]:
        real = torch.from_numpy(rs.uniform(-1, 1, (n, 3, h, w)).astype(np.float32))
        mask = torch.from_numpy((rs.rand(n, 1, h, w) < 0.6).astype(np.float32))
        ref = torch.cat([mask - 0.5, real * mask], dim=1)
        x = kk.assemble_input(real.to(DEV), mask.to(DEV))
        assert np.array_equal(c(x), ref.numpy())
    with pytest.raises(RuntimeError):
        kk.assemble_input(torch.zeros(1, 3, 3, 3, device=DEV), torch.zeros(1, 1, 3, 3, device=DEV))     # H*W % 4 != 0


def test_conv2d_gradfix_surface(mods):
    import torch.nn.functional as F
    rs = np.random.RandomState(11)
    x = torch.from_numpy(rs.standard_normal((2, 6, 10, 10)).astype(np.float32))
    w = torch.from_numpy(rs.standard_normal((8, 6, 3, 3)).astype(np.float32))
    b = torch.from_numpy(rs.standard_normal(8).astype(np.float32))
    y = mods['gradfix'].conv2d(x.to(DEV), w.to(DEV), b.to(DEV), padding=1)
    assert rel_err(c(y), F.conv2d(x, w, b, padding=1).numpy()) < 2e-5
    wt = torch.from_numpy(rs.standard_normal((6, 8, 3, 3)).astype(np.float32))
    y = mods['gradfix'].conv_transpose2d(x.to(DEV), wt.to(DEV), stride=2)
    assert rel_err(c(y), F.conv_transpose2d(x, wt, stride=2).numpy()) < 2e-5


def test_fir_march_at_baseline_sizes_vs_tiled_kernels(mods):
    """The marching FIR kernels at the layer sizes of BASELINE config 3 (512 x 512, batch 16: 64 channels at 512^2 ... 512 at 64^2)
    against the tiled kernels they replace (which the oracle tests above pin), plus the size-independent checks the domain offers:
    a constant image filters to the filter's DC gain away from the border, and the polyphase planes re-interleave to the plain result."""
    K, ufd, orc = mods['kernels'], mods['ufd'], mods['orc']
    f = orc.setup_filter([1, 3, 3, 1]).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(5)
    for ch, r in ((64, 512), (128, 256), (256, 128), (512, 64)):
        x = torch.randn(16, ch, r, r, device=DEV, generator=g)
        y = ufd.upfirdn2d(x, f, padding=[2, 2, 2, 2])
        K.FIR_MARCH = False
        try:
            y_old = ufd.upfirdn2d(x, f, padding=[2, 2, 2, 2])
        finally:
            K.FIR_MARCH = True
        assert float((y - y_old).abs().max()) < 2e-6 * float(y_old.abs().max())
        # planes of fir_conv_down2's first half == the plain result, re-interleaved
        lib = K._lib.get_lib()
        pp = (r // 2 + 1 + 31) // 32 * 32 if r % 256 == 0 else (r // 2 + 1 + 3) // 4 * 4
        xp = torch.empty((4, 16, ch, r // 2 + 1, pp), device=DEV)
        K.check(lib.shg_fir_pad2_sep_f32(K._ptr(x), K.sep_taps(f), K._ptr(xp), 16, ch, r, r, pp, 0, 1.0, None), 'fir_pad2_sep')
        for a in range(2):
            for b in range(2):
                d = xp[a * 2 + b][:, :, :(r + 2 - a) // 2, :(r + 2 - b) // 2] - y[:, :, a::2, b::2]      # (two kernels: FMA contraction differs)
                assert float(d.abs().max()) < 1e-6 * float(y.abs().max())
        assert float(xp[1][:, :, :, r // 2:].abs().max()) == 0.0 and float(xp[2][:, :, r // 2:].abs().max()) == 0.0
        del xp, y, y_old
        # FIR-from-planes with the whole tail, low-resolution r/2 -> r
        h = r // 2
        mid = torch.randn(4, 16, ch, h + 1, h + 1, device=DEV, generator=g)
        kw = dict(scale=torch.rand(16 * ch, device=DEV, generator=g) + 0.5, bias=torch.randn(ch, device=DEV, generator=g),
                  noise=torch.randn(16, 1, r, r, device=DEV, generator=g), residual=torch.randn(16, ch, r, r, device=DEV, generator=g))
        z = K.upfir_planar(mid, f, noise_strength=0.3, act=True, **kw)
        K.FIR_MARCH = False
        try:
            z_old = K.upfir_planar(mid, f, noise_strength=0.3, act=True, **kw)
        finally:
            K.FIR_MARCH = True
        assert float((z - z_old).abs().max()) < 2e-6 * float(z_old.abs().max())
        del mid, kw, z, z_old, x
    ones = torch.ones(2, 3, 256, 256, device=DEV)
    yc = ufd.upfirdn2d(ones, f, padding=[2, 2, 2, 2])
    assert float((yc[:, :, 2:-2, 2:-2] - 1.0).abs().max()) < 1e-6            # [1,3,3,1] x [1,3,3,1] / 64 sums to 1
    yd = ufd.upfirdn2d(ones, f, down=2, padding=[1, 1, 1, 1])
    yu = ufd.upfirdn2d(ones[:, :, :128, :128], f, up=2, padding=[2, 1, 2, 1], gain=4)
    assert float((yd[:, :, 1:-1, 1:-1] - 1.0).abs().max()) < 1e-6 and float((yu[:, :, 2:-2, 2:-2] - 1.0).abs().max()) < 1e-6


# ---- the native op over the plugin's whole operand range (upfirdn2d.cpp:38-59: any strides, float / double / half) -------------------------

_UFD_CASES = [
    # up, down, padding, flip, gain, filter shape
    (1, 1, [2, 2, 2, 2], False, 1.0, (4, 4)),
    (2, 1, [2, 1, 2, 1], False, 4.0, (4, 4)),
    (1, 2, [1, 1, 1, 1], True, 1.0, (4, 4)),
    ([2, 1], [1, 3], [3, 0, -1, 2], True, 0.75, (3, 5)),
    (3, 2, [0, 4, 2, 1], False, 2.5, (6, 2)),
]


@pytest.mark.parametrize('layout', ['f64_nchw', 'f64_cl', 'f32_cl', 'f16_nchw', 'f32_view', 'f64_view'])
def test_native_op_serves_every_dtype_and_layout_of_the_plugin(mods, layout):
    """``upfirdn2d_plugin.upfirdn2d`` accepts float32 / float16 / float64 tensors with any strides and returns y in x's suggested memory
    format (upfirdn2d.cpp:37-59).  The streaming kernels cover the two dense network layouts; ``shg_upfirdn2d_strided`` serves the rest in
    place: checked against the oracle's float64 evaluation of upfirdn2d.py:98-138 on the same (rounded) operands, incl. a non-contiguous
    FILTER view (the op passes the filter's strides, :48), asymmetric / negative padding and per-axis factors."""
    from shgan_amd.model_zoo.stylegan_utils import custom_ops
    plugin = custom_ops.get_plugin('upfirdn2d_plugin')
    orc = mods['orc']
    rs = np.random.RandomState(11)
    dt = {'f64': torch.float64, 'f32': torch.float32, 'f16': torch.float16}[layout[:3]]
    tol = {'f64': 1e-13, 'f32': 2e-6, 'f16': 1.5e-3}[layout[:3]]
    for up, down, pad, flip, gain, fs in _UFD_CASES:
        x = torch.from_numpy(rs.standard_normal((2, 6, 9, 11))).to(dt)
        if layout.endswith('_cl'):
            xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
        elif layout.endswith('_view'):
            big = torch.from_numpy(rs.standard_normal((2, 9, 12, 16))).to(dt).to(DEV)
            xd = big[:, 1:7, 2:11, 3:14:1]                       # a strided window of a larger tensor: neither layout is dense
            x = xd.cpu()
        else:
            xd = x.to(DEV)
        # a filter that is a transposed view: stride(0) = 1 (the op reads it through its strides)
        fT = torch.from_numpy(rs.standard_normal((fs[1], fs[0])).astype(np.float32)).to(DEV)
        f = fT.t()
        assert not f.is_contiguous() or 1 in fs
        upx, upy = (up, up) if isinstance(up, int) else up
        dnx, dny = (down, down) if isinstance(down, int) else down
        y = plugin.upfirdn2d(xd, f, upx, upy, dnx, dny, pad[0], pad[1], pad[2], pad[3], flip, gain)
        ref = orc.upfirdn2d(x.double(), f.cpu().contiguous().double(), up=[upx, upy], down=[dnx, dny], padding=pad, flip_filter=flip, gain=gain)
        assert y.dtype == dt and tuple(y.shape) == tuple(ref.shape), (layout, up, down)
        if layout.endswith('_cl'):
            assert y.is_contiguous(memory_format=torch.channels_last)                      # upfirdn2d.cpp:37 suggest_memory_format
        else:
            assert y.is_contiguous()
        e = rel_err(c(y.double()), ref.numpy())
        assert e < tol, (layout, up, down, pad, e)


def test_public_upfirdn2d_in_float64_passes_gradcheck_first_and_second_order(mods):
    """The public operator (upfirdn2d.py:141-192 semantics: its gradient is the same op with up / down exchanged and the filter flipped)
    on float64 tensors, now that the native op takes them: torch.autograd.gradcheck / gradgradcheck against finite differences."""
    ufd = mods['ufd']
    rs = np.random.RandomState(12)
    f = ufd.setup_filter([1, 3, 3, 1]).to(DEV)
    f1 = ufd.setup_filter([1, 2, 1], separable=True).to(DEV)
    with torch.enable_grad():
        for kw in (dict(up=2, padding=[2, 1, 2, 1], gain=4.0), dict(down=2, padding=[1, 1, 1, 1]), dict(padding=[2, 2, 2, 2], flip_filter=True)):
            x = torch.from_numpy(rs.standard_normal((1, 2, 5, 6))).to(DEV).requires_grad_(True)
            fn = lambda t: ufd.upfirdn2d(t, f, **kw)          # noqa: E731
            assert fn(x).dtype == torch.float64
            assert torch.autograd.gradcheck(fn, (x,), eps=1e-6, atol=1e-7, rtol=1e-6, nondet_tol=0.0)
            assert torch.autograd.gradgradcheck(fn, (x,), eps=1e-6, atol=1e-7, rtol=1e-6, nondet_tol=0.0)
        assert f1.ndim == 1                                   # separable: one pass per axis (upfirdn2d.py:164-168)
        x = torch.from_numpy(rs.standard_normal((1, 2, 5, 6))).to(DEV).requires_grad_(True)
        assert torch.autograd.gradcheck(lambda t: ufd.upfirdn2d(t, f1, up=2, padding=[1, 1, 1, 1]), (x,), eps=1e-6, atol=1e-7, rtol=1e-6)


def test_strided_native_op_rejects_bad_arguments(mods):
    k = mods['kernels']
    from shgan_amd import _lib
    f = torch.ones(2, 2, device=DEV)
    with pytest.raises(_lib.ShgError):
        k.upfirdn2d_strided(torch.zeros(1, 2, 4, 4, device=DEV, dtype=torch.int32), f)
    with pytest.raises(_lib.ShgError):
        k.upfirdn2d_strided(torch.zeros(1, 2, 4, 4, device=DEV, dtype=torch.float64), f.double())
    with pytest.raises(_lib.ShgError):
        k.upfirdn2d_strided(torch.zeros(1, 2, 1, 1, device=DEV, dtype=torch.float64), torch.ones(4, 4, device=DEV))         # empty output
    with pytest.raises(_lib.ShgError):
        k.upfirdn2d_strided(torch.zeros(1, 2, 4, 4, dtype=torch.float64), torch.ones(2, 2))                                  # CPU tensors
