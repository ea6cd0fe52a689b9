"""Floor of the Spectral Hint Unit's three launches: every stage of the product (csrc/shu.hip) beside a skeleton
kernel of the same grid / workgroup / LDS footprint / global load + store pattern that does no arithmetic (tools/micro/shu_floor.hip),
and beside an empty launch of the same grid.  The three stages run as the DEPENDENT chain they are in the encoder
(rfft2 -> spectral -> split + irfft2 accumulating into the skip features) at the bench shape: N = 16, 32 channels, 64 x 64.

  python tools/shu_floor.py [--build-only]      # build here (cross-compile), run on the GPU box -> profiles/r06_shu_floor.txt
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'tools', '_variants', 'libshu_floor.so')
SRC = os.path.join(ROOT, 'tools', 'micro', 'shu_floor.hip')


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-o', LIB, SRC])
    return LIB


def main():
    build()
    if '--build-only' in sys.argv:
        return
    import torch
    import shgan_amd  # noqa: F401
    from shgan_amd import configs, kernels
    dev = 'cuda:0'
    fl = ctypes.CDLL(LIB)
    N, C = 16, 32
    G = configs.seeded_init_(configs.build_generator(512), seed=0).eval().requires_grad_(False).to(dev)
    shu = G.encoder.shu
    x_full = torch.randn(N, 512, 64, 64, device=dev)
    x = x_full[:, -C:]
    feats = {r: torch.randn(N, 512, r, r, device=dev) for r in (4, 8, 16, 32, 64)}
    outs = [feats[r][:, 512 - C:] for r in (4, 8, 16, 32, 64)]
    gauss = [getattr(shu, f'_gauss{r}') for r in shu.reslist]
    w0p, b0, w1p = shu._packed()
    cw = shu._cw
    P = 64 * 33
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())                       # noqa: E731
    T = kernels.shu_rfft2_shift(x)
    S = kernels.shu_spectral(T, w0p, b0, w1p, cw)
    o_arr = (ctypes.c_void_p * 5)(*[o.data_ptr() for o in outs])
    s_arr = (ctypes.c_long * 5)(*[o.stride(0) for o in outs])

    def real_rfft2():
        kernels.shu_rfft2_shift(x)

    def real_spectral():
        kernels.shu_spectral(T, w0p, b0, w1p, cw)

    def real_split():
        kernels.shu_split_irfft2(S, None, gauss, outs, accumulate=True)

    def real_chain():
        t = kernels.shu_rfft2_shift(x)
        s = kernels.shu_spectral(t, w0p, b0, w1p, cw)
        kernels.shu_split_irfft2(s, None, gauss, outs, accumulate=True)

    def fl_rfft2():
        fl.floor_rfft2(vp(x), ctypes.c_long(x.stride(0)), vp(T), N, C, st)

    def fl_spectral():
        fl.floor_spectral(vp(T), vp(w0p), vp(w1p), vp(cw), vp(S), N, P, cw.shape[0], st)

    def fl_split():
        fl.floor_split(vp(S), o_arr, s_arr, N, C, st)

    def fl_chain():
        fl_rfft2(); fl_spectral(); fl_split()

    def empty(gx, gy):
        return lambda: fl.floor_empty(gx, gy, st)

    def time_us(fn, iters=400):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters * 1e3)
        return best

    rows = []
    for name, real, floor, grid in (('shu_rfft2', real_rfft2, fl_rfft2, (C, N)), ('shu_spectral', real_spectral, fl_spectral, (P // 64, N)),
                                    ('shu_irfft2', real_split, fl_split, (C, N)), ('chain of the three', real_chain, fl_chain, None)):
        r, f = time_us(real), time_us(floor)
        e = time_us(empty(*grid)) if grid else 3 * rows[0][3]
        rows.append((name, r, f, e))
    print('Spectral Hint Unit: product launch vs a no-arithmetic skeleton of the same launch vs an empty launch of the same grid')
    print(f'(N = {N}, {C} channels, 64 x 64; back-to-back launches on one stream, HIP events over 400 launches, best of 5; microseconds per launch)')
    print(f'{"stage":22s} {"product":>9s} {"skeleton":>9s} {"empty":>7s} {"product/skeleton":>17s}')
    for name, r, f, e in rows:
        print(f'{name:22s} {r:9.2f} {f:9.2f} {e:7.2f} {r / f:17.2f}')
    # the matrix work of each stage at the fp32-MFMA peak, for scale
    print('arithmetic at 157.3 TFLOP/s: rfft2 1.21 GFLOP (dense DFT as MFMA passes: 2 x 64^3 x 2 + 2 x 128 x 64 x 32 x 2 per plane) = 7.7 us; '
          'spectral 1.94 GFLOP = 12.3 us; split + irfft2 (64^2 level as MFMA) 1.2 GFLOP = 7.7 us')


if __name__ == '__main__':
    main()
