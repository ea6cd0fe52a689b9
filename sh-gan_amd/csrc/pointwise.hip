// HBM-bound pointwise / thin-channel kernels of the generator path (gfx950).
//   * bias_act      : x + bias -> lrelu_agc             (stylegan.py:232-238,298-304; common/utils.py:135-143)
//   * fromrgb 1x1   : conv 1x1 with tiny I (4) + bias + lrelu_agc   (stylegan.py:226-238 via comodgan.py:45-51)
//   * torgb         : modulated 1x1 conv to 3 channels, no demod, + bias, fused with the FIR-upsampled
//                     running RGB image  img = upsample2d(img_prev) + torgb(x)   (stylegan.py:325-337,
//                     comodgan.py:331-338, upfirdn2d.py:279-314)
//   * composite_u8  : x[:,1:4]*m + img*(1-m) -> *127.5+127.5 -> clamp -> uint8 (truncation)
//                     (lib/experiments/shgan_default.py:257-262)
//   * scale_channels / fma: the non-fused modconv pieces (stylegan.py:173-180, stylegan_utils/fma.py:15)
#include "shg_common.h"

// ---------------------------------------------------------------------------------------------
// y = act((x [* scale[n,c]]) + noise*strength + bias[c]) [+ residual]   over [N,C,HW]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bias_act_kernel(const float* x, float* y, const float* scale, const float* bias,
                                                       const float* noise, int noise_mode, float noise_strength,
                                                       const float* residual, int C, int HW, long total, int act, float alpha,
                                                       float gain, float clamp) {
    const long stride = (long)gridDim.x * 256;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
        const long nc = e / HW;
        const int pix = (int)(e - nc * HW);
        const int c = (int)(nc % C);
        const long n = nc / C;
        float v = x[e];
        if (scale) v *= scale[nc];
        if (noise_mode == 1) v += noise[pix] * noise_strength;
        else if (noise_mode == 2) v += noise[n * HW + pix] * noise_strength;
        if (bias) v += bias[c];
        v = act ? shg_lrelu_agc(v, alpha, gain, clamp) : v * gain;
        if (residual) v += residual[e];
        y[e] = v;
    }
}

// HW % 4 == 0 and < 2^31 float4 pieces: one float4 per lane and trip, 32-bit index arithmetic, the per-plane operands read once per float4.
// The scalar kernel above (three 64-bit divisions per ELEMENT) ran at 3.2 TB/s of read + write traffic on the 512^2 tensors of the training
// step and 2.1 with scale + noise; this one 5.6 / 4.9 (tools/pointwise_bench.py).  PU > 1 (several float4 requested before the first use)
// measured SLOWER: 4.5 / 4.3 / 3.8 TB/s at 2 / 4 / 8 -- the pass is not short of loads in flight.
#ifndef SHG_PW_PU
#define SHG_PW_PU 1
#endif
constexpr int PU = SHG_PW_PU;
__global__ __launch_bounds__(256) void bias_act_v4_kernel(const float4* x, float4* y, const float* scale, const float* bias, const float4* noise,
                                                          int noise_mode, float noise_strength, const float4* residual, unsigned C, unsigned HW4,
                                                          unsigned total4, int act, float alpha, float gain, float clamp) {
    const unsigned stride = gridDim.x * 256u;
    for (unsigned eb = blockIdx.x * 256u + threadIdx.x; eb < total4; eb += PU * stride) {          // (host: total4 + PU * stride < 2^32)
        float4 v[PU], r[PU], nz[PU];
        float sc[PU], bi[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const unsigned e = eb + u * stride < total4 ? eb + u * stride : total4 - 1;
            const unsigned nc = e / HW4, pix4 = e - nc * HW4, n = nc / C, c = nc - n * C;
            v[u] = x[e];
            r[u] = residual ? residual[e] : make_float4(0.f, 0.f, 0.f, 0.f);
            nz[u] = noise_mode == 1 ? noise[pix4] : (noise_mode == 2 ? noise[n * HW4 + pix4] : make_float4(0.f, 0.f, 0.f, 0.f));
            sc[u] = scale ? scale[nc] : 1.f;
            bi[u] = bias ? bias[c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            float4 o;
            float* op = &o.x;
            const float *vp = &v[u].x, *rp = &r[u].x, *np_ = &nz[u].x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float t = vp[k];
                if (scale) t *= sc[u];
                if (noise_mode) t += np_[k] * noise_strength;
                if (bias) t += bi[u];
                t = act ? shg_lrelu_agc(t, alpha, gain, clamp) : t * gain;
                if (residual) t += rp[k];
                op[k] = t;
            }
            if (eb + u * stride < total4) y[eb + u * stride] = o;
        }
    }
}

extern "C" int shg_bias_act_f32(const float* x, float* y, const float* scale, const float* bias, const float* noise,
                                int noise_mode, float noise_strength, const float* residual, int N, int C, int HW, int act,
                                float alpha, float gain, float clamp, void* stream) {
    SHG_CHECK_ARG(x && y, "bias_act: null pointer");
    SHG_CHECK_ARG(N >= 0 && C >= 1 && HW >= 1, "bias_act: bad shape");
    const long total = (long)N * C * HW;
    if (total == 0) return SHG_OK;
    const bool al16 = ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)noise | (uintptr_t)residual) & 15) == 0);
    if ((HW & 3) == 0 && al16 && total / 4 < (1L << 31)) {
        const long total4 = total / 4;
        int grid = shg_cdiv(total4, 256L * PU);
        if (grid > 256 * 16) grid = 256 * 16;
        hipLaunchKernelGGL(bias_act_v4_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (float4*)y, scale, bias,
                           (const float4*)noise, noise ? noise_mode : 0, noise_strength, (const float4*)residual, (unsigned)C, (unsigned)(HW / 4),
                           (unsigned)total4, act, alpha, gain, clamp);
        SHG_CHECK_LAUNCH();
        return SHG_OK;
    }
    int grid = shg_cdiv(total, 256);
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(bias_act_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, scale, bias, noise,
                       noise ? noise_mode : 0, noise_strength, residual, C, HW, total, act, alpha, gain, clamp);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// Backward of y = lrelu_agc(x + bias) (common/utils.py:135-143 under autograd: leaky_relu -> *gain -> clamp): with the
// OUTPUT y at hand, dx = g * (|y| < clamp ? (y > 0 ? gain : alpha*gain) : 0) -- gain, alpha > 0 keep the sign, and a clamped
// value has zero slope.  The bias gradient is the per-channel sum of dx (a reduction left to the caller).
__global__ __launch_bounds__(256) void bias_act_backward_kernel(const float* g, const float* y, float* dx, long total, int act,
                                                                 float alpha, float gain, float clamp) {
    const long stride = (long)gridDim.x * 256;
    const float gp = gain, gn = act ? alpha * gain : gain;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
        const float v = y[e];
        const float slope = (act && clamp >= 0.f && fabsf(v) >= clamp) ? 0.f : (v > 0.f ? gp : gn);
        dx[e] = g[e] * slope;
    }
}

// (one float per lane: three coalesced streams run at 5.5 TB/s as they are; a float4 version measured 10 % slower on the 512^2 tensors)
// dx = dL/dx of y = act(x + bias) given g = dL/dy and the forward OUTPUT y (same act / alpha / gain / clamp as the forward call)
extern "C" int shg_bias_act_backward_f32(const float* g, const float* y, float* dx, long total, int act, float alpha, float gain,
                                         float clamp, void* stream) {
    SHG_CHECK_ARG(g && y && dx, "bias_act_backward: null pointer");
    SHG_CHECK_ARG(total >= 0, "bias_act_backward: bad size");
    if (total == 0) return SHG_OK;
    int grid = shg_cdiv(total, 256);
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(bias_act_backward_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, y, dx, total, act, alpha, gain, clamp);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// ---------------------------------------------------------------------------------------------
// c = a*b + c' with NCHW broadcasting of b:[N,C,1,1] and c':[N,1,H,W] or [H,W]  (stylegan.py:176)
// handled by bias_act (scale = b, noise = c').  Plain elementwise fma for the generic op:
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fma_kernel(const float* a, const float* b, const float* c, float* y, long total) {
    const long stride = (long)gridDim.x * 256;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) y[e] = fmaf(a[e], b[e], c[e]);
}

extern "C" int shg_fma_f32(const float* a, const float* b, const float* c, float* y, long total, void* stream) {
    SHG_CHECK_ARG(a && b && c && y, "fma: null pointer");
    if (total <= 0) return SHG_OK;
    int grid = shg_cdiv(total, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(fma_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, c, y, total);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// ---------------------------------------------------------------------------------------------
// The generic op with NumPy broadcasting (stylegan_utils/fma.py:15-58): operands are addressed through element strides (0 on a
// broadcast dimension) so that nothing is materialised, and the gradient of a broadcast operand -- `_unbroadcast(dout * b)`,
// fma.py:40-58 -- is ONE product + reduction pass.  float32 and float64 (the public op is differentiable twice and is held to
// finite differences in float64).  Dimensions are collapsed by the caller; at most SHG_BC_DIMS remain.
// ---------------------------------------------------------------------------------------------
#define SHG_BC_DIMS 6
struct ShgBcast {
    int nd;
    long size[SHG_BC_DIMS];
    long sa[SHG_BC_DIMS], sb[SHG_BC_DIMS], sc[SHG_BC_DIMS];
};

template <typename T>
__global__ __launch_bounds__(256) void fma_bcast_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c,
                                                        T* __restrict__ y, long total, ShgBcast g) {
    const long stride = (long)gridDim.x * 256;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
        long r = e, ia = 0, ib = 0, ic = 0;
#pragma unroll
        for (int d = SHG_BC_DIMS - 1; d >= 0; --d) {
            if (d < g.nd) {
                const long q = r / g.size[d], i = r - q * g.size[d];
                r = q;
                ia += i * g.sa[d];
                ib += i * g.sb[d];
                ic += i * g.sc[d];
            }
        }
        const T p = a[ia] * b[ib];
        y[e] = c ? T(a[ia] * b[ib] + c[ic]) : p;
    }
}
// float32: one rounding, as torch.addcmul's fused form on the device
template <>
__global__ __launch_bounds__(256) void fma_bcast_kernel<float>(const float* __restrict__ a, const float* __restrict__ b,
                                                               const float* __restrict__ c, float* __restrict__ y, long total, ShgBcast g) {
    const long stride = (long)gridDim.x * 256;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
        long r = e, ia = 0, ib = 0, ic = 0;
#pragma unroll
        for (int d = SHG_BC_DIMS - 1; d >= 0; --d) {
            if (d < g.nd) {
                const long q = r / g.size[d], i = r - q * g.size[d];
                r = q;
                ia += i * g.sa[d];
                ib += i * g.sb[d];
                ic += i * g.sc[d];
            }
        }
        y[e] = c ? fmaf(a[ia], b[ib], c[ic]) : a[ia] * b[ib];
    }
}

static int shg_bcast_fill(ShgBcast& g, int nd, const long* shape, const long* sa, const long* sb, const long* sc, long& total) {
    if (nd < 0 || nd > SHG_BC_DIMS) return -1;
    g.nd = nd;
    total = 1;
    for (int d = 0; d < SHG_BC_DIMS; ++d) {
        g.size[d] = d < nd ? shape[d] : 1;
        g.sa[d] = d < nd && sa ? sa[d] : 0;
        g.sb[d] = d < nd && sb ? sb[d] : 0;
        g.sc[d] = d < nd && sc ? sc[d] : 0;
        if (g.size[d] < 0) return -1;
        total *= g.size[d];
    }
    return 0;
}

extern "C" int shg_fma_bcast(const void* a, const void* b, const void* c, void* y, int nd, const long* shape, const long* sa,
                             const long* sb, const long* sc, int f64, void* stream) {
    SHG_CHECK_ARG(a && b && y && (nd == 0 || (shape && sa && sb)) && (!c || nd == 0 || sc), "fma_bcast: null pointer");
    ShgBcast g;
    long total;
    SHG_CHECK_ARG(shg_bcast_fill(g, nd, shape, sa, sb, sc, total) == 0, "fma_bcast: at most %d (collapsed) dimensions, sizes >= 0", SHG_BC_DIMS);
    if (total == 0) return SHG_OK;
    int grid = shg_cdiv(total, 256);
    if (grid > 8192) grid = 8192;
    if (f64)
        hipLaunchKernelGGL(fma_bcast_kernel<double>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const double*)a, (const double*)b,
                           (const double*)c, (double*)y, total, g);
    else
        hipLaunchKernelGGL(fma_bcast_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)a, (const float*)b,
                           (const float*)c, (float*)y, total, g);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// out[kept index] = sum over the reduced index of g * b (b == nullptr: of g).  The caller orders the dimensions KEPT FIRST, REDUCED
// LAST (nk kept dimensions), so that `out` is the contiguous tensor of the kept sizes.  Two mappings, chosen by the caller from
// the strides: `inner_kept` -- the fastest-varying dimension of g is a kept one: one thread per output, neighbouring threads read
// neighbouring addresses, the reduction is a serial walk; otherwise one workgroup per output with its 256 lanes along the
// reduced index (coalesced when the fastest dimension is reduced) and a fixed-order tree in LDS.  Both orders are fixed: deterministic.
template <typename T>
__device__ __forceinline__ T mul_reduce_term(const T* __restrict__ g, const T* __restrict__ b, long r, long og, long ob, const ShgBcast& s, int nk) {
    long ig = og, ib = ob;
#pragma unroll
    for (int d = SHG_BC_DIMS - 1; d >= 0; --d) {
        if (d >= nk && d < s.nd) {
            const long q = r / s.size[d], i = r - q * s.size[d];
            r = q;
            ig += i * s.sa[d];
            ib += i * s.sb[d];
        }
    }
    return b ? g[ig] * b[ib] : g[ig];
}

template <typename T>
__global__ __launch_bounds__(256) void mul_reduce_kernel(const T* __restrict__ g, const T* __restrict__ b, T* __restrict__ out, long n_out,
                                                         long n_red, ShgBcast s, int nk, int inner_kept) {
    __shared__ T part[256];
    if (inner_kept) {
        const long o = (long)blockIdx.x * 256 + threadIdx.x;
        if (o >= n_out) return;
        long r = o, og = 0, ob = 0;
#pragma unroll
        for (int d = SHG_BC_DIMS - 1; d >= 0; --d) {
            if (d < nk) {
                const long q = r / s.size[d], i = r - q * s.size[d];
                r = q;
                og += i * s.sa[d];
                ob += i * s.sb[d];
            }
        }
        T acc = T(0);
        for (long k = 0; k < n_red; ++k) acc += mul_reduce_term(g, b, k, og, ob, s, nk);
        out[o] = acc;
        return;
    }
    for (long o = blockIdx.x; o < n_out; o += gridDim.x) {
        long r = o, og = 0, ob = 0;
#pragma unroll
        for (int d = SHG_BC_DIMS - 1; d >= 0; --d) {
            if (d < nk) {
                const long q = r / s.size[d], i = r - q * s.size[d];
                r = q;
                og += i * s.sa[d];
                ob += i * s.sb[d];
            }
        }
        T acc = T(0);
        for (long k = threadIdx.x; k < n_red; k += 256) acc += mul_reduce_term(g, b, k, og, ob, s, nk);
        part[threadIdx.x] = acc;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[o] = part[0];
        __syncthreads();
    }
}

extern "C" int shg_mul_reduce(const void* g, const void* b, void* out, int nd, int nk, const long* shape, const long* sg, const long* sb,
                              int inner_kept, int f64, void* stream) {
    SHG_CHECK_ARG(g && out && (nd == 0 || (shape && sg)) && (!b || nd == 0 || sb), "mul_reduce: null pointer");
    SHG_CHECK_ARG(nk >= 0 && nk <= nd, "mul_reduce: kept dimensions %d of %d", nk, nd);
    ShgBcast s;
    long total;
    SHG_CHECK_ARG(shg_bcast_fill(s, nd, shape, sg, sb, nullptr, total) == 0, "mul_reduce: at most %d (collapsed) dimensions, sizes >= 0", SHG_BC_DIMS);
    long n_out = 1, n_red = 1;
    for (int d = 0; d < nd; ++d) (d < nk ? n_out : n_red) *= shape[d];
    if (n_out == 0) return SHG_OK;
    long grid = inner_kept ? (n_out + 255) / 256 : (n_out < 65536 ? n_out : 65536);
    SHG_CHECK_ARG(grid <= 0x7fffffffL, "mul_reduce: too many outputs");
    if (f64)
        hipLaunchKernelGGL(mul_reduce_kernel<double>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const double*)g, (const double*)b,
                           (double*)out, n_out, n_red, s, nk, inner_kept);
    else
        hipLaunchKernelGGL(mul_reduce_kernel<float>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const float*)g, (const float*)b,
                           (float*)out, n_out, n_red, s, nk, inner_kept);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// ---------------------------------------------------------------------------------------------
// Thin 1x1 convolution: y[n,o,p] = act(sum_i w[o,i]*wgain * x[n,i,p] * s[n,i] + bias[o]) [+ base]
// for small I (fromrgb, I = 4) or small O (torgb, O = 3): one lane per 4 pixels, float4 I/O.
// `base_up` (optional) is the previous-resolution RGB image [N,O,H/2,W/2]; it is FIR-upsampled x2
// on the fly with the 4x4 filter `f` (pad [2,1,2,1], gain 4: upfirdn2d.py:305-314) and added.
// ---------------------------------------------------------------------------------------------
template <int MAXI>
__global__ __launch_bounds__(256) void conv1x1_small_i_kernel(const float* x, const float* w, const float* bias, float* y, int I,
                                                              int O, int HW, float wgain, int act, float alpha, float gain,
                                                              float clamp, int PB, int OC) {
    // fromrgb / the toRGB input gradient: a thread owns 4 consecutive pixels of one sample and walks output channels.  A workgroup =
    // PB pixel groups (a power of two <= 256) x 256 / PB channel slices of the OC channels blockIdx.z selects: low resolutions (3 -> 512
    // at 64^2 and below: 8-32 workgroups of 512 serial stores per thread, 230-250 us whatever the size) spread their channels over
    // the grid instead; the arithmetic per output is unchanged.
    const int n = blockIdx.y, pl = threadIdx.x & (PB - 1), osl = threadIdx.x / PB, S = 256 / PB;
    const long p4 = ((long)blockIdx.x * PB + pl) * 4;
    if (p4 >= HW) return;
    float4 xv[MAXI];
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
        xv[i] = i < I ? *reinterpret_cast<const float4*>(x + ((long)n * I + i) * HW + p4) : make_float4(0, 0, 0, 0);
    const int o_end = min(O, ((int)blockIdx.z + 1) * OC);
    for (int o = blockIdx.z * OC + osl; o < o_end; o += S) {
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            if (i < I) {
                const float wv = w[o * I + i] * wgain;
                acc.x += wv * xv[i].x; acc.y += wv * xv[i].y; acc.z += wv * xv[i].z; acc.w += wv * xv[i].w;
            }
        }
        const float b = bias ? bias[o] : 0.f;
        float r[4] = {acc.x + b, acc.y + b, acc.z + b, acc.w + b};
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = act ? shg_lrelu_agc(r[k], alpha, gain, clamp) : r[k] * gain;
        *reinterpret_cast<float4*>(y + ((long)n * O + o) * HW + p4) = make_float4(r[0], r[1], r[2], r[3]);
    }
}

extern "C" int shg_conv1x1_thin_in_f32(const float* x, const float* w, const float* bias, float* y, int N, int I, int O, int HW,
                                       float wgain, int act, float alpha, float gain, float clamp, void* stream) {
    SHG_CHECK_ARG(x && w && y, "conv1x1_thin_in: null pointer");
    SHG_CHECK_ARG(I >= 1 && I <= 8, "conv1x1_thin_in: I must be in 1..8 (got %d)", I);
    SHG_CHECK_ARG(HW % 4 == 0, "conv1x1_thin_in: H*W must be a multiple of 4");
    SHG_CHECK_ARG(N >= 1 && N <= 65535, "conv1x1_thin_in: bad N");
    const int P4 = HW / 4;
    int PB = 256;
    if (P4 < 256) { PB = 1; while (PB < P4) PB <<= 1; }
    const int gx = shg_cdiv(P4, PB), S = 256 / PB;
    long oz = shg_cdiv(2048, gx * N);                              // aim at >= 2048 workgroups; a workgroup takes at least one channel per slice
    if (oz < 1) oz = 1;
    int OC = shg_cdiv(O, (int)(oz > O ? O : oz));
    if (OC < S) OC = S;
    if (OC > O) OC = O;
    dim3 grid(gx, N, shg_cdiv(O, OC));
    hipLaunchKernelGGL((conv1x1_small_i_kernel<8>), grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, I, O, HW, wgain, act,
                       alpha, gain, clamp, PB, OC);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

template <int MAXO>
__global__ __launch_bounds__(256) void torgb_kernel(const float* x, const float* w, const float* styles, const float* bias,
                                                    const float* base_up, const float* f, float* y, int I, int O, int H, int W,
                                                    int S) {
    // ws[o][i] = w[o,i] * styles[n,i] staged in LDS.  A workgroup = 256/S pixels x S channel slices (S = 1 for large
    // images: one thread per pixel, coalesced rows; small images split the channel loop S ways so that the serial
    // chain per thread stays short) -- the slices are summed through LDS.
    extern __shared__ float ws[];   // [MAXO][I] + [S-1][256/S][MAXO] partial sums
    const int n = blockIdx.z;
    for (int k = threadIdx.x; k < O * I; k += 256) {
        const int i = k % I;
        ws[k] = w[k] * (styles ? styles[(long)n * I + i] : 1.f);
    }
    __syncthreads();
    const int HW = H * W;
    const int PPB = 256 / S;                         // pixels per workgroup
    const int pl = threadIdx.x % PPB, slice = threadIdx.x / PPB;
    const int pix = blockIdx.x * PPB + pl;
    const bool inside = pix < HW;
    float acc[MAXO];
#pragma unroll
    for (int o = 0; o < MAXO; ++o) acc[o] = 0.f;
    const int per = (I + S - 1) / S;
    const int i_lo = slice * per, i_hi = min(I, i_lo + per);
    if (inside) {
        const float* xp = x + (long)n * I * HW + pix;
        int i = i_lo;
        for (; i + 4 <= i_hi; i += 4) {
            const float x0 = xp[(long)i * HW], x1 = xp[(long)(i + 1) * HW], x2 = xp[(long)(i + 2) * HW], x3 = xp[(long)(i + 3) * HW];
#pragma unroll
            for (int o = 0; o < MAXO; ++o)
                if (o < O) acc[o] += ws[o * I + i] * x0 + ws[o * I + i + 1] * x1 + ws[o * I + i + 2] * x2 + ws[o * I + i + 3] * x3;
        }
        for (; i < i_hi; ++i) {
            const float x0 = xp[(long)i * HW];
#pragma unroll
            for (int o = 0; o < MAXO; ++o)
                if (o < O) acc[o] += ws[o * I + i] * x0;
        }
    }
    if (S > 1) {
        float* red = ws + MAXO * I;                  // [S-1][PPB][MAXO]
        if (slice > 0) {
#pragma unroll
            for (int o = 0; o < MAXO; ++o) red[((slice - 1) * PPB + pl) * MAXO + o] = acc[o];
        }
        __syncthreads();
        if (slice > 0) return;
        for (int sl = 0; sl < S - 1; ++sl)
#pragma unroll
            for (int o = 0; o < MAXO; ++o) acc[o] += red[(sl * PPB + pl) * MAXO + o];
    }
    if (!inside) return;
    const int oy = pix / W, ox = pix - oy * W;
#pragma unroll
    for (int o = 0; o < MAXO; ++o) {
        if (o >= O) break;
        float v = acc[o] + (bias ? bias[o] : 0.f);
        if (base_up) {
            // upsample2d(up=2, 4x4 filter, pad [2,1,2,1], gain 4): u = oy + ky - 2 must be even, iy = u/2
            const int h2 = H >> 1, w2 = W >> 1;
            const float* bp = base_up + ((long)n * O + o) * h2 * w2;
            float u = 0.f;
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
                const int uy = oy + ky - 2;
                if (uy < 0 || (uy & 1)) continue;
                const int iy = uy >> 1;
                if (iy >= h2) continue;
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    const int ux = ox + kx - 2;
                    if (ux < 0 || (ux & 1)) continue;
                    const int ix = ux >> 1;
                    if (ix >= w2) continue;
                    u += bp[iy * w2 + ix] * (f[(3 - ky) * 4 + (3 - kx)] * 4.f);
                }
            }
            v += u;
        }
        y[((long)n * O + o) * HW + pix] = v;
    }
}

// The same for W % 4 == 0 and 16-byte aligned tensors: one lane per FOUR horizontally adjacent pixels (16-byte loads, 8 channels =
// 8 KB per wave in flight, 16-byte stores); the modulated weights w[o,i]*styles[n,i] sit at wave-uniform addresses in LDS.
// S channel slices per pixel group for small images, summed through LDS like above.  O <= 3.
__global__ __launch_bounds__(256) void torgb4_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ styles, const float* __restrict__ bias,
                                                     const float* __restrict__ base_up, const float* __restrict__ f,
                                                     float* __restrict__ y, int I, int O, int H, int W, int S) {
    extern __shared__ float ws[];   // [3][I] + [S-1][256/S][12] partial sums
    const int n = blockIdx.z;
    for (int k = threadIdx.x; k < 3 * I; k += 256) {
        const int i = k % I;
        ws[k] = k < O * I ? w[k] * (styles ? styles[(long)n * I + i] : 1.f) : 0.f;
    }
    __syncthreads();
    const int HW = H * W, Q = HW >> 2;               // pixel quads per plane
    const int QPB = 256 / S;
    const int ql = threadIdx.x % QPB, slice = threadIdx.x / QPB;
    const int quad = blockIdx.x * QPB + ql;
    const bool inside = quad < Q;
    float4 acc[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int per = ((I + S - 1) / S + 7) & ~7;
    const int i_lo = slice * per, i_hi = min(I, i_lo + per);
    const float4* xp = reinterpret_cast<const float4*>(x + (long)n * I * HW) + (inside ? quad : 0);
    int i = i_lo;
    for (; i + 8 <= i_hi; i += 8) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = xp[(long)(i + k) * Q];
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float c = ws[o * I + i + k];
                acc[o].x += c * v[k].x; acc[o].y += c * v[k].y; acc[o].z += c * v[k].z; acc[o].w += c * v[k].w;
            }
    }
    for (; i < i_hi; ++i) {
        const float4 v = xp[(long)i * Q];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const float c = ws[o * I + i];
            acc[o].x += c * v.x; acc[o].y += c * v.y; acc[o].z += c * v.z; acc[o].w += c * v.w;
        }
    }
    if (S > 1) {
        float4* red = reinterpret_cast<float4*>(ws + 3 * I + ((3 * I) & 3 ? 4 - ((3 * I) & 3) : 0));   // [S-1][QPB][3]
        if (slice > 0) {
#pragma unroll
            for (int o = 0; o < 3; ++o) red[((slice - 1) * QPB + ql) * 3 + o] = acc[o];
        }
        __syncthreads();
        if (slice > 0) return;
        for (int sl = 0; sl < S - 1; ++sl)
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float4 r = red[(sl * QPB + ql) * 3 + o];
                acc[o].x += r.x; acc[o].y += r.y; acc[o].z += r.z; acc[o].w += r.w;
            }
    }
    if (!inside) return;
    const int pix = quad * 4, oy = pix / W, ox = pix - oy * W;
    for (int o = 0; o < O; ++o) {
        const float b = bias ? bias[o] : 0.f;
        float out[4] = {acc[o].x + b, acc[o].y + b, acc[o].z + b, acc[o].w + b};
        if (base_up) {
            // upsample2d(up=2, 4x4 filter, pad [2,1,2,1], gain 4): u = oy + ky - 2 must be even, iy = u/2 (same order of the
            // additions as torgb_kernel)
            const int h2 = H >> 1, w2 = W >> 1;
            const float* bp = base_up + ((long)n * O + o) * h2 * w2;
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                float u = 0.f;
#pragma unroll
                for (int ky = 0; ky < 4; ++ky) {
                    const int uy = oy + ky - 2;
                    if (uy < 0 || (uy & 1)) continue;
                    const int iy = uy >> 1;
                    if (iy >= h2) continue;
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) {
                        const int ux = ox + dx + kx - 2;
                        if (ux < 0 || (ux & 1)) continue;
                        const int ix = ux >> 1;
                        if (ix >= w2) continue;
                        u += bp[iy * w2 + ix] * (f[(3 - ky) * 4 + (3 - kx)] * 4.f);
                    }
                }
                out[dx] += u;
            }
        }
        *reinterpret_cast<float4*>(y + ((long)n * O + o) * HW + pix) = make_float4(out[0], out[1], out[2], out[3]);
    }
}

// y[n,o] = sum_i w[o,i]*styles[n,i]*x[n,i] + bias[o] (+ upsample2d(base_up, f)); O <= 4.
extern "C" int shg_torgb_f32(const float* x, const float* w, const float* styles, const float* bias, const float* base_up,
                             const float* f, float* y, int N, int I, int O, int H, int W, void* stream) {
    SHG_CHECK_ARG(x && w && y, "torgb: null pointer");
    SHG_CHECK_ARG(O >= 1 && O <= 4, "torgb: O must be in 1..4 (got %d)", O);
    SHG_CHECK_ARG(!base_up || (f && H % 2 == 0 && W % 2 == 0), "torgb: base_up needs a 4x4 filter and even H, W");
    SHG_CHECK_ARG(N >= 1 && N <= 65535, "torgb: bad N");
    SHG_CHECK_ARG((size_t)4 * I * 4 + 4096 <= 64 * 1024, "torgb: I too large");
    // channel slices per pixel: enough to put >= ~64k threads on the chip (small images are otherwise one long serial chain)
    // (measured at N = 16: 512^2 x 64ch 236 vs 257 us, 64^2 x 512ch 38 vs 63 us; in between -- 128^2 x 256ch -- the pixel-per-lane
    //  kernel's 16 waves per CU win, 59 vs 65 us)
    const long nquad = (long)N * (H * W / 4);
    if (O <= 3 && W % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && (nquad >= 262144 || nquad <= 16384)) {
        // quad kernel: channel slices until ~256k lanes are busy, at least 8 channels per slice and 16 quads per row piece
        int S4 = 1;
        while (S4 < 16 && (long)N * (H * W / 4) * S4 < 262144 && I / (S4 * 2) >= 8) S4 *= 2;
        const size_t lds = sizeof(float) * (((3 * I + 3) & ~3) + (size_t)(S4 - 1) * (256 / S4) * 12);
        if (lds <= 64 * 1024) {
            dim3 g4(shg_cdiv(H * W / 4, 256 / S4), 1, N);
            hipLaunchKernelGGL(torgb4_kernel, g4, dim3(256), lds, (hipStream_t)stream, x, w, styles, bias, base_up, f, y, I, O, H, W, S4);
            SHG_CHECK_LAUNCH();
            return SHG_OK;
        }
    }
    int S = 1;
    while (S < 16 && (long)N * H * W * S < 65536 && I / (S * 2) >= 8) S *= 2;
    dim3 grid(shg_cdiv(H * W, 256 / S), 1, N);
    hipLaunchKernelGGL((torgb_kernel<4>), grid, dim3(256), sizeof(float) * (4 * I + (S - 1) * (256 / S) * 4), (hipStream_t)stream, x, w,
                       styles, bias, base_up, f, y, I, O, H, W, S);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// ---------------------------------------------------------------------------------------------
// Phase planes of the transposed convolution -> image: y[nc][Y][X] = full[Y + lo][X + lo] (+ bias[c]), where
// full[Yf][Xf] = mid[(Yf&1)*2 + (Xf&1)][nc][Yf>>1][Xf>>1] is the (2H+1) x (2W+1) result of conv_transpose2d(stride 2) and positions
// outside it read as zero -- the crop / zero-extension `conv2d_gradfix` applies around a transposed convolution
// (`padding`, and the output_padding rule of conv2d_gradfix.py:96-105), fused with the interleave of the planes.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void planes_to_image_kernel(const float* __restrict__ mid, const float* __restrict__ bias,
                                                              float* __restrict__ y, int NC, int C, int H, int W, int lo, int OH, int OW) {
    const int X = blockIdx.x * 256 + threadIdx.x, Y = blockIdx.y, nc = blockIdx.z;
    if (X >= OW) return;
    const int Yf = Y + lo, Xf = X + lo;
    float v = 0.f;
    if (Yf >= 0 && Yf <= 2 * H && Xf >= 0 && Xf <= 2 * W)
        v = mid[(((long)((Yf & 1) * 2 + (Xf & 1)) * NC + nc) * (H + 1) + (Yf >> 1)) * (W + 1) + (Xf >> 1)];
    if (bias) v += bias[nc % C];
    y[((long)nc * OH + Y) * OW + X] = v;
}

// mid [4][N*C][H+1][W+1] (shg_conv2d_f32 mode 2 with out_mode 1, or shg_conv2d_up_poly_f32) -> y [N*C, OH, OW]; bias [C] or null
extern "C" int shg_planes_to_image_f32(const float* mid, const float* bias, float* y, int N, int C, int H, int W, int lo, int OH,
                                       int OW, void* stream) {
    SHG_CHECK_ARG(mid && y, "planes_to_image: null pointer");
    SHG_CHECK_ARG(N >= 1 && C >= 1 && H >= 1 && W >= 1 && OH >= 1 && OW >= 1, "planes_to_image: bad shape");
    SHG_CHECK_ARG((long)N * C <= 65535 * 1L && OH <= 65535, "planes_to_image: too many planes / rows for one launch");
    SHG_CHECK_ARG(4L * N * C * (H + 1) * (W + 1) <= 2147483647L && (long)N * C * OH * OW <= 2147483647L, "planes_to_image: tensor too large");
    hipLaunchKernelGGL(planes_to_image_kernel, dim3(shg_cdiv(OW, 256), OH, N * C), dim3(256), 0, (hipStream_t)stream, mid, bias, y,
                       N * C, C, H, W, lo, OH, OW);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}


// ---------------------------------------------------------------------------------------------
// First-order backward of the modulation tail y = A(t * d[n,c] + noise + bias[c]) of a float32 NCHW layer in ONE pass (training rows:
// stylegan.py:173,176-181 + :298-304 under autograd; the forward is bias_act_kernel with scale / noise / bias):
//   gz = gy * A'(y) from the saved output;  gt = gz * d;  per-(n,c) pixel sums of gz*t (-> d) and gz (-> bias);  per-pixel channel
//   sums of gz (-> noise).  A lane owns 4 adjacent pixels (16-byte accesses) and walks over ALL channels: the channel sum stays in its
//   registers, the pixel sums are reduced per wave (DPP) and per workgroup (LDS) into part[n][block][2][C] -- summed by the caller in
//   a fixed order (deterministic, no atomics).  HW % 4 == 0.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void modtail_backward_f32_kernel(const float* gy, const float* y, const float* t, const float* d, const float* u,
                                                                    const float* e, float* gt, float* part, float* gnoise, int C, int HW, int nblk,
                                                                    int act, float alpha, float gain, float clamp) {
    __shared__ float red[2][4][512];                                // per-wave sums of every channel: one barrier at the end (C <= 512)
    const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float gp = gain, gn = act ? alpha * gain : gain;
    const int p4 = (blockIdx.x * 256 + tid) * 4;                    // first of this lane's 4 pixels
    const bool ok = p4 < HW;
    float4 nsum = make_float4(0.f, 0.f, 0.f, 0.f);
    // small images: the channels are split over gridDim.z workgroups (each writes its own slice of the noise-gradient partials)
    const int cpc = (C + gridDim.z - 1) / gridDim.z, c_lo = blockIdx.z * cpc, c_hi = c_lo + cpc < C ? c_lo + cpc : C;
    for (int c = c_lo; c < c_hi; ++c) {
        const long off = ((long)n * C + c) * HW + p4;
        float s1 = 0.f, s0 = 0.f;
        if (ok) {
            const float4 g = *(const float4*)(gy + off), yv = *(const float4*)(y + off);
            float4 tv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t) tv = *(const float4*)(t + off);
            const float dd = d ? d[(long)n * C + c] : 1.f;
            float4 uv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u) uv = *(const float4*)(u + off);
            const float ee = (u && e) ? e[(long)n * C + c] : (u ? 1.f : 0.f);
            const float gq[4] = {g.x, g.y, g.z, g.w}, yq[4] = {yv.x, yv.y, yv.z, yv.w}, tq[4] = {tv.x, tv.y, tv.z, tv.w}, uq[4] = {uv.x, uv.y, uv.z, uv.w};
            float gz[4], o4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float slope = (act && clamp >= 0.f && fabsf(yq[q]) >= clamp) ? 0.f : ((yq[q] > 0.f || !act) ? gp : gn);
                gz[q] = gq[q] * slope;
                s1 += gz[q] * tq[q];
                s0 += gz[q];
                o4[q] = u ? __builtin_fmaf(uq[q] * slope, ee, gz[q] * dd) : gz[q] * dd;
            }
            nsum.x += gz[0]; nsum.y += gz[1]; nsum.z += gz[2]; nsum.w += gz[3];
            *(float4*)(gt + off) = make_float4(o4[0], o4[1], o4[2], o4[3]);
        }
        if (part) {                                                 // (uniform branch: whole waves take part in the shuffles)
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { s1 += __shfl_xor(s1, m, 64); s0 += __shfl_xor(s0, m, 64); }
            if (lane == 0) { red[0][wave][c - c_lo] = s1; red[1][wave][c - c_lo] = s0; }
        }
    }
    if (part) {
        __syncthreads();
        const int cn = c_hi - c_lo;
        for (int e = tid; e < 2 * cn; e += 256) {
            const int k = e / cn, c = e - k * cn;
            part[(((long)n * nblk + blockIdx.x) * 2 + k) * C + c_lo + c] = red[k][0][c] + red[k][1][c] + red[k][2][c] + red[k][3][c];
        }
    }
    if (gnoise && ok) *(float4*)(gnoise + ((long)blockIdx.z * gridDim.y + n) * HW + p4) = nsum;
}

extern "C" int shg_modtail_backward_f32_blocks(long HW) { return (int)((HW / 4 + 255) / 256); }
// channel slices the launch is split into (rows of the gnoise partial buffer): enough workgroups for the small images
extern "C" int shg_modtail_backward_f32_cslices(int N, int C, long HW) {
    const long wg = (long)shg_modtail_backward_f32_blocks(HW) * N;
    long z = (512 + wg - 1) / wg;
    if (z > C / 8) z = C / 8;
    return (int)(z < 1 ? 1 : z);
}

// gt = gy * A'(y) * d [+ u * A'(y) * e]; part [N][blocks][2][C] = per-workgroup pixel sums of gz*t and gz (NULL: skipped; t may be NULL); gnoise
// [cslices][N,HW] = channel sums of gz per channel slice (NULL: skipped; the caller adds the slices).  u [N,C,HW] / e [N,C] (both optional): the second
// product of the tail's double backward, d/dgy = A'(y) (ggt d + t ggd).  NCHW float32, HW % 4 == 0, 16-byte aligned tensors.
extern "C" int shg_modtail_backward_f32(const float* gy, const float* y, const float* t, const float* d, const float* u, const float* e, float* gt,
                                        float* part, float* gnoise, int N, int C, long HW, int act, float alpha, float gain, float clamp, void* stream) {
    SHG_CHECK_ARG(gy && y && gt && N >= 1 && C >= 1 && C <= 512 && HW >= 4 && (HW % 4) == 0, "modtail_backward_f32: bad arguments (C <= 512, HW a multiple of 4)");
    SHG_CHECK_ARG(((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(gt) |
                    reinterpret_cast<uintptr_t>(gnoise) | reinterpret_cast<uintptr_t>(u)) & 15) == 0, "modtail_backward_f32: tensors must be 16-byte aligned");
    const int nblk = shg_modtail_backward_f32_blocks(HW);
    hipLaunchKernelGGL(modtail_backward_f32_kernel, dim3(nblk, N, shg_modtail_backward_f32_cslices(N, C, HW)), dim3(256), 0, (hipStream_t)stream, gy, y,
                       t, d, u, e, gt, part, gnoise, C, (int)HW, nblk, act, alpha, gain, clamp);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// ---------------------------------------------------------------------------------------------
// y = half(x * gain) and y = float(x) * gain: `(weight * weight_gain).to(x.dtype)` of the fp16 layers (stylegan.py:228,236-238 with half
// activations) and its gradient as ONE launch each instead of a product and a cast (90 + 127 library launches per training step).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scale_cast_to_half_kernel(const float* x, _Float16* y, long n, float gain) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) y[e] = (_Float16)(x[e] * gain);
}
__global__ __launch_bounds__(256) void scale_cast_to_float_kernel(const _Float16* x, float* y, long n, float gain) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) y[e] = (float)x[e] * gain;
}

// to_half != 0: src float32 -> dst float16 = half(src * gain) (one rounding, as the tensor operators); else src float16 -> dst float32 = float(src) * gain
extern "C" int shg_scale_cast_f32_f16(const void* src, void* dst, long n, float gain, int to_half, void* stream) {
    SHG_CHECK_ARG(src && dst && n >= 0, "scale_cast: bad arguments");
    if (n == 0) return SHG_OK;
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (to_half) hipLaunchKernelGGL(scale_cast_to_half_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)src, (_Float16*)dst, n, gain);
    else hipLaunchKernelGGL(scale_cast_to_float_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)src, (float*)dst, n, gain);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// ---------------------------------------------------------------------------------------------
// out[n, :] = sum_b part[n, b, :]   (the per-workgroup partial sums of the tail-backward kernels -> per-sample sums, in block order:
// deterministic).  A library reduction of [8, 256, 2, 512] took 8 us, 129 of them per training step; here a workgroup owns 64 columns of
// one sample: 4 row groups x 64 columns, each thread adds its rows in order (four interleaved accumulators), the groups meet in LDS.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* part, float* out, int B, int K) {
    __shared__ float red[4][64];
    const int n = blockIdx.y, col = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (col < K) {
        const float* p = part + (long)n * B * K + col;
        int b = g;
        for (; b + 12 < B; b += 16) { a0 += p[(long)b * K]; a1 += p[(long)(b + 4) * K]; a2 += p[(long)(b + 8) * K]; a3 += p[(long)(b + 12) * K]; }
        for (; b < B; b += 4) a0 += p[(long)b * K];
    }
    red[g][threadIdx.x & 63] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (g == 0 && col < K) out[(long)n * K + col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

extern "C" int shg_sum_partials_f32(const float* part, float* out, int N, int B, int K, void* stream) {
    SHG_CHECK_ARG(part && out && N >= 1 && N <= 65535 && B >= 1 && K >= 1, "sum_partials: bad arguments");
    hipLaunchKernelGGL(sum_partials_kernel, dim3(shg_cdiv(K, 64), N), dim3(256), 0, (hipStream_t)stream, part, out, B, K);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// ---------------------------------------------------------------------------------------------
// x[n,i,:] *= s[n,i]   (non-fused modulation, stylegan.py:173)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scale_channels_kernel(const float* x, const float* s, float* y, int HW, long total) {
    const long stride = (long)gridDim.x * 256;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) y[e] = x[e] * s[e / HW];
}

extern "C" int shg_scale_channels_f32(const float* x, const float* s, float* y, int NC, int HW, void* stream) {
    SHG_CHECK_ARG(x && s && y, "scale_channels: null pointer");
    const long total = (long)NC * HW;
    if (total <= 0) return SHG_OK;
    int grid = shg_cdiv(total, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(scale_channels_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, s, y, HW, total);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// ---------------------------------------------------------------------------------------------
// eval composite -> uint8   (shgan_default.py:257-262); x4 = [N,4,R,R] (ch0 = mask-0.5, ch1..3 = rgb*mask)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void composite_u8_kernel(const float* x4, const float* img, uint8_t* out, int HW, long total4) {
    // one thread = 4 consecutive pixels of one (n, c) plane
    const long stride = (long)gridDim.x * 256;
    const int HW4 = HW >> 2;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total4; e += stride) {
        const long nc = e / HW4;
        const int p4 = (int)(e - nc * HW4) * 4;
        const long n = nc / 3;
        const int c = (int)(nc - n * 3);
        const float4 mk = *reinterpret_cast<const float4*>(x4 + (n * 4) * HW + p4);
        const float4 kn = *reinterpret_cast<const float4*>(x4 + (n * 4 + 1 + c) * HW + p4);
        const float4 gi = *reinterpret_cast<const float4*>(img + nc * HW + p4);
        const float m[4] = {mk.x + 0.5f, mk.y + 0.5f, mk.z + 0.5f, mk.w + 0.5f};
        const float k[4] = {kn.x, kn.y, kn.z, kn.w};
        const float g[4] = {gi.x, gi.y, gi.z, gi.w};
        uint32_t packed = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // every product and sum is rounded separately, exactly as the PyTorch expression evaluates
            // them op by op: fused multiply-adds would move values across the truncation boundary
            // (HIP's __fmul_rn/__fadd_rn are plain operators, so contraction is switched off here).
#pragma clang fp contract(off)
            const float a = k[j] * m[j];
            const float b = g[j] * (1.0f - m[j]);
            float v = a + b;
            v = v * 127.5f;
            v = v + 127.5f;
            v = fminf(fmaxf(v, 0.f), 255.f);
            packed |= ((uint32_t)(uint8_t)(int)v) << (8 * j);
        }
        *reinterpret_cast<uint32_t*>(out + nc * HW + p4) = packed;
    }
}

extern "C" int shg_composite_u8(const float* x4, const float* img, uint8_t* out, int N, int H, int W, void* stream) {
    SHG_CHECK_ARG(x4 && img && out, "composite_u8: null pointer");
    SHG_CHECK_ARG(N >= 0 && H >= 1 && W >= 1 && (H * W) % 4 == 0, "composite_u8: H*W must be a positive multiple of 4");
    const long total4 = (long)N * 3 * (H * W / 4);
    if (total4 == 0) return SHG_OK;
    int grid = shg_cdiv(total4, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(composite_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x4, img, out, H * W, total4);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// ------------------------------------------------------------------------------------------------
// Input assembly of the eval loop (lib/experiments/shgan_default.py:267-274):
//   x = cat([mask - 0.5, real * mask], dim=1)     real [N,3,H,W], mask [N,H,W] (0/1 floats) -> x [N,4,H,W]
// one pass instead of the reference's sub + mul + cat.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void assemble_input_kernel(const float* real, const float* mask, float* x, int HW4, long total4) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;          // one float4 of one image plane
    if (e >= total4) return;
    const int n = (int)(e / HW4), q = (int)(e - (long)n * HW4);
    const float4 m = reinterpret_cast<const float4*>(mask)[(long)n * HW4 + q];
    float4* xo = reinterpret_cast<float4*>(x) + (long)n * 4 * HW4 + q;
    xo[0] = make_float4(m.x - 0.5f, m.y - 0.5f, m.z - 0.5f, m.w - 0.5f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float4 r = reinterpret_cast<const float4*>(real)[((long)n * 3 + c) * HW4 + q];
        xo[(long)(c + 1) * HW4] = make_float4(r.x * m.x, r.y * m.y, r.z * m.z, r.w * m.w);
    }
}

extern "C" int shg_assemble_input_f32(const float* real, const float* mask, float* x, int N, int H, int W, void* stream) {
    SHG_CHECK_ARG(real && mask && x, "assemble_input: null pointer");
    SHG_CHECK_ARG(N >= 0 && H >= 1 && W >= 1 && (H * W) % 4 == 0, "assemble_input: H*W must be a positive multiple of 4");
    SHG_CHECK_ARG(((reinterpret_cast<uintptr_t>(real) | reinterpret_cast<uintptr_t>(mask) | reinterpret_cast<uintptr_t>(x)) & 15) == 0,
                  "assemble_input: buffers must be 16-byte aligned");
    if (N == 0) return SHG_OK;
    const long total4 = (long)N * (H * W / 4);
    hipLaunchKernelGGL(assemble_input_kernel, dim3(shg_cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream, real, mask, x, H * W / 4, total4);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// The same hand-off from the DECODED bytes (ds_ffhq.py:307-347: uint8 -> ToTensor /255 -> *2-1; shgan_default.py:267-274): the loader
// ships uint8 pixels (a quarter of the PCIe bytes of float images) and the value of every code comes from a 256-entry table the
// caller computed with the host formatter's own arithmetic, so x is bit-identical to the host route whatever that arithmetic rounds to.
__global__ __launch_bounds__(256) void assemble_input_u8_kernel(const uint8_t* real, const float* mask, const float* lut, float* x, int HW4,
                                                                long total4) {
    __shared__ float tab[256];
    tab[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const long e = (long)blockIdx.x * 256 + threadIdx.x;          // four pixels of one image plane
    if (e >= total4) return;
    const int n = (int)(e / HW4), q = (int)(e - (long)n * HW4);
    const float4 m = reinterpret_cast<const float4*>(mask)[(long)n * HW4 + q];
    float4* xo = reinterpret_cast<float4*>(x) + (long)n * 4 * HW4 + q;
    xo[0] = make_float4(m.x - 0.5f, m.y - 0.5f, m.z - 0.5f, m.w - 0.5f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uchar4 r = reinterpret_cast<const uchar4*>(real)[((long)n * 3 + c) * HW4 + q];
        xo[(long)(c + 1) * HW4] = make_float4(tab[r.x] * m.x, tab[r.y] * m.y, tab[r.z] * m.z, tab[r.w] * m.w);
    }
}

extern "C" int shg_assemble_input_u8(const uint8_t* real, const float* mask, const float* lut, float* x, int N, int H, int W, void* stream) {
    SHG_CHECK_ARG(real && mask && lut && x, "assemble_input_u8: null pointer");
    SHG_CHECK_ARG(N >= 0 && H >= 1 && W >= 1 && (H * W) % 4 == 0, "assemble_input_u8: H*W must be a positive multiple of 4");
    SHG_CHECK_ARG(((reinterpret_cast<uintptr_t>(mask) | reinterpret_cast<uintptr_t>(x)) & 15) == 0 && (reinterpret_cast<uintptr_t>(real) & 3) == 0,
                  "assemble_input_u8: mask / x must be 16-byte aligned, real 4-byte aligned");
    if (N == 0) return SHG_OK;
    const long total4 = (long)N * (H * W / 4);
    hipLaunchKernelGGL(assemble_input_u8_kernel, dim3(shg_cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream, real, mask, lut, x, H * W / 4, total4);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// ------------------------------------------------------------------------------------------------
// minibatch_std_layer (stylegan.py:686-704): x [N,C,H,W] -> y [N,C+F,H,W].  The batch is split into N/G groups of G
// samples (sample of group-slot g and group n = g*(N/G) + n), the channels into F sets of c = C/F; the statistic of
// (n, f) is the mean over (c, H, W) of the standard deviation over the G samples; it is appended as channel C+f of
// every sample of group n.  Launch 1: one workgroup per (n, f) reduces; launch 2 copies x and broadcasts the statistic.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mbstd_stat_kernel(const float* x, float* stat, int NG, int G, int F, int c, int HW, int C) {
    const int n = blockIdx.x / F, f = blockIdx.x % F;
    const int per = c * HW;
    const long sample = (long)C * HW;
    float acc = 0.f;
    for (int e = threadIdx.x; e < per; e += 256) {
        const float* px = x + (long)n * sample + (long)f * per + e;     // + g*NG*sample
        float mean = 0.f;
        for (int g = 0; g < G; ++g) mean += px[(long)g * NG * sample];
        mean /= (float)G;
        float var = 0.f;
        for (int g = 0; g < G; ++g) { const float d = px[(long)g * NG * sample] - mean; var += d * d; }
        acc += sqrtf(var / (float)G + 1e-8f);
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) stat[n * F + f] = red[0] / (float)per;
}

__global__ __launch_bounds__(256) void mbstd_write_kernel(const float* x, const float* stat, float* y, int NG, int F, int C, int HW, long total) {
    const long stride = (long)gridDim.x * 256;
    const int CF = C + F;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
        const long nc = e / HW;
        const int pix = (int)(e - nc * HW);
        const int s = (int)(nc / CF), ch = (int)(nc - (long)s * CF);
        y[e] = ch < C ? x[((long)s * C + ch) * HW + pix] : stat[(s % NG) * F + (ch - C)];
    }
}

extern "C" int shg_minibatch_std_f32(const float* x, float* y, float* stat, int N, int C, int H, int W, int G, int F, void* stream) {
    SHG_CHECK_ARG(x && y && stat, "minibatch_std: null pointer");
    SHG_CHECK_ARG(N >= 1 && C >= 1 && H >= 1 && W >= 1 && G >= 1 && F >= 1, "minibatch_std: bad shape");
    SHG_CHECK_ARG(N % G == 0, "minibatch_std: batch %d is not a multiple of the group size %d", N, G);
    SHG_CHECK_ARG(C % F == 0, "minibatch_std: %d channels do not split into %d sets", C, F);
    const int NG = N / G, HW = H * W;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mbstd_stat_kernel, dim3(NG * F), dim3(256), 0, s, x, stat, NG, G, F, C / F, HW, C);
    SHG_CHECK_LAUNCH();
    const long total = (long)N * (C + F) * HW;
    int grid = shg_cdiv(total, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(mbstd_write_kernel, dim3(grid), dim3(256), 0, s, x, stat, y, NG, F, C, HW, total);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}
