"""Round-6 op-surface tests: the public ``fma`` under autograd (lib/model_zoo/stylegan_utils/fma.py:15-58:
``da = _unbroadcast(dout*b)``, ``db = _unbroadcast(dout*a)``, ``dc = _unbroadcast(dout)``) on the broadcasting HIP kernels, held
to finite differences in float64 and to torch's own autograd of ``a*b+c`` in float32; the prepared-weight cache's guard against
writes that do not move a parameter's version counter."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'

# the reference's own uses (stylegan.py:176-180: x[N,C,H,W] * d[N,C,1,1] + noise[N,1,H,W]) and the broadcasting corners
SHAPES = [
    ((2, 5, 6, 7), (2, 5, 1, 1), (2, 1, 6, 7)),
    ((2, 5, 6, 7), (2, 5, 1, 1), (6, 7)),
    ((3, 4), (3, 4), (3, 4)),
    ((4, 1, 3), (2, 1, 5, 1), (5, 3)),
    ((1,), (2, 3), ()),
    ((2, 3), (1,), (4, 2, 3)),            # c enlarges the result: a and b are both unbroadcast over the leading dimension
]


@pytest.fixture(scope='module')
def fma():
    import shgan_amd  # noqa: F401
    from shgan_amd.model_zoo.stylegan_utils import fma
    assert torch.cuda.is_available()
    return fma.fma


def _rand(rs, shape, dtype):
    return torch.from_numpy(np.asarray(rs.standard_normal(shape))).to(dtype).to(DEV)


@pytest.mark.parametrize('sa,sb,sc', SHAPES)
def test_fma_forward_and_first_order_vs_torch_autograd(fma, sa, sb, sc):
    rs = np.random.RandomState(3)
    with torch.enable_grad():
        ops = [_rand(rs, s, torch.float32).requires_grad_(True) for s in (sa, sb, sc)]
        y = fma(*ops)
        assert y.grad_fn is not None, 'the public op must be differentiable (fma.py:20-46)'
        ref_ops = [t.detach().cpu().double().requires_grad_(True) for t in ops]
        ref = ref_ops[0] * ref_ops[1] + ref_ops[2]
        assert tuple(y.shape) == tuple(ref.shape)
        assert rel_err(y.detach().cpu().numpy(), ref.detach().numpy()) < 1e-6
        go = _rand(rs, tuple(ref.shape), torch.float32)
        grads = torch.autograd.grad(y, ops, go)
        ref_grads = torch.autograd.grad(ref, ref_ops, go.cpu().double())
        for gr, rg, t in zip(grads, ref_grads, ops):
            assert gr.shape == t.shape
            assert rel_err(gr.cpu().numpy(), rg.numpy()) < 2e-6


def test_fma_strided_and_expanded_operands_are_read_in_place(fma):
    rs = np.random.RandomState(4)
    a = _rand(rs, (2, 8, 6, 10), torch.float32)[:, 2:7, :, ::2]            # channel slice + column stride
    b = _rand(rs, (2, 5, 1, 1), torch.float32).expand(2, 5, 6, 5)
    c = _rand(rs, (5, 6), torch.float32).t().contiguous().t()[None, None].transpose(2, 3)   # transposed view [1,1,6,5]
    y = fma(a, b, c)
    ref = torch.addcmul(c.cpu(), a.cpu(), b.cpu())
    assert y.is_contiguous() and rel_err(y.cpu().numpy(), ref.numpy()) < 1e-6


def test_fma_float64_gradcheck_first_and_second_order(fma):
    rs = np.random.RandomState(5)
    with torch.enable_grad():
        for sa, sb, sc in SHAPES[:4] + SHAPES[5:]:
            ops = tuple(_rand(rs, s, torch.float64).requires_grad_(True) for s in (sa, sb, sc))
            assert fma(*ops).dtype == torch.float64
            assert torch.autograd.gradcheck(fma, ops, eps=1e-6, atol=1e-7, rtol=1e-6, nondet_tol=0.0)
            assert torch.autograd.gradgradcheck(fma, ops, eps=1e-6, atol=1e-7, rtol=1e-6, nondet_tol=0.0)


def test_fma_large_reductions_both_mappings(fma):
    """The two mappings of shg_mul_reduce at a layer-sized operand: d [N,C,1,1] (the fastest dimension is reduced: a workgroup per
    output) and noise [N,1,H,W] / [H,W] (the fastest dimension is kept: a thread per output)."""
    rs = np.random.RandomState(6)
    with torch.enable_grad():
        x = _rand(rs, (3, 40, 64, 64), torch.float32).requires_grad_(True)
        d = _rand(rs, (3, 40, 1, 1), torch.float32).requires_grad_(True)
        for nshape in ((3, 1, 64, 64), (64, 64)):
            n = _rand(rs, nshape, torch.float32).requires_grad_(True)
            go = _rand(rs, (3, 40, 64, 64), torch.float32)
            gx, gd, gn = torch.autograd.grad(fma(x, d, n), (x, d, n), go)
            g64, x64, d64 = go.cpu().double(), x.detach().cpu().double(), d.detach().cpu().double()
            assert rel_err(gx.cpu().numpy(), (g64 * d64).numpy()) < 1e-6
            assert rel_err(gd.cpu().numpy(), (g64 * x64).sum((2, 3), keepdim=True).numpy()) < 1e-5
            ref_n = g64.sum(1, keepdim=True) if len(nshape) == 4 else g64.sum((0, 1))
            assert gn.shape == n.shape and rel_err(gn.cpu().numpy(), ref_n.numpy()) < 1e-5
            # determinism: fixed summation order
            gd2 = torch.autograd.grad(fma(x, d, n), d, go)[0]
            assert torch.equal(gd, gd2)


def test_fma_rejects_cpu_and_mixed_dtypes(fma):
    from shgan_amd import _lib
    a = torch.ones(2, 2, device=DEV)
    with pytest.raises(_lib.ShgError):
        fma(a.cpu(), a.cpu(), a.cpu())
    with pytest.raises(_lib.ShgError):
        fma(a, a.double(), a)
    with pytest.raises(_lib.ShgError):
        fma(a.half(), a.half(), a.half())


def test_param_cache_guard_notices_data_writes():
    """`_ParamCache` keys on the parameters' version counters; a write through `.data` does not move them.  The documented contract
    is `invalidate_param_caches()` after such a write; with the guard on, a stale hit raises instead of computing with old weights."""
    import shgan_amd  # noqa: F401
    from shgan_amd.model_zoo import stylegan
    layer = stylegan.conv2d_layer(8, 8, kernel_size=3, activation='lrelu_agc(gain=sqrt_2)').to(DEV).eval()
    x = torch.randn(1, 8, 16, 16, device=DEV)
    PC = stylegan._ParamCache
    old = PC.guard
    try:
        PC.guard = True
        y0 = layer(x).clone()
        with torch.no_grad():
            layer.weight.mul_(2.0)                     # a tracked write: the cache follows by itself
        y1 = layer(x).clone()
        assert not torch.equal(y0, y1)
        layer.weight.data.mul_(0.5)                    # an untracked write
        with pytest.raises(RuntimeError, match='stale prepared weights'):
            layer(x)
        stylegan.invalidate_param_caches()
        y2 = layer(x)
        assert rel_err(y2.cpu().numpy(), y0.cpu().numpy()) < 1e-6
        PC.guard = False
        layer.weight.data.mul_(2.0)                    # guard off: documented behaviour -- stale until invalidated
        assert torch.equal(layer(x), y2)
        stylegan.invalidate_param_caches()
        assert rel_err(layer(x).cpu().numpy(), y1.cpu().numpy()) < 1e-6
    finally:
        PC.guard = old
