// Skinny fully-connected layers and the style/demodulation preparation (gfx950).
//   * dense            : y = act(x @ (W*wgain)^T + b*bgain)              (stylegan.py:87-98)
//   * normalize_2nd_moment                                               (stylegan.py:343-344)
//   * modconv_style_prep: s = styles * rsqrt(mean(styles^2)) (batch-global, stylegan.py:147),
//                         dcoef[n,o] = rsqrt(sum_i s[n,i]^2 * wsq[i,o] + 1e-8)   (stylegan.py:150-155)
// The batch is tiny (N = 16..32) and every weight is read exactly once, so these are HBM/L2
// streaming kernels: one wave per output feature walks a weight row with coalesced 256 B reads
// while the activations of the whole batch sit in LDS.
#include "shg_common.h"

#define DENSE_MAXN 32     // samples per pass (batch is processed in chunks of DENSE_MAXN)
#define DENSE_KC 512      // K chunk staged in LDS: DENSE_MAXN * DENSE_KC * 4 B = 64 KiB

__global__ __launch_bounds__(256) void dense_kernel(const float* x, const float* w, const float* b, float* y, int N, int K, int O,
                                                    int ldx, int ldy, float wgain, float bgain, int act, float alpha, float gain,
                                                    float clamp) {
    extern __shared__ float xs[];   // [nb][DENSE_KC]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.y * DENSE_MAXN;
    const int nb = min(DENSE_MAXN, N - n0);
    const int o = blockIdx.x * 4 + wave;
    float acc[DENSE_MAXN];
#pragma unroll
    for (int n = 0; n < DENSE_MAXN; ++n) acc[n] = 0.f;
    for (int k0 = 0; k0 < K; k0 += DENSE_KC) {
        const int kc = min(DENSE_KC, K - k0);
        __syncthreads();
        for (int e = threadIdx.x; e < nb * DENSE_KC; e += 256) {
            const int n = e / DENSE_KC, k = e - n * DENSE_KC;
            xs[e] = k < kc ? x[(long)(n0 + n) * ldx + k0 + k] : 0.f;
        }
        __syncthreads();
        if (o < O) {
            const float* wr = w + (long)o * K + k0;
            for (int k = lane; k < kc; k += 64) {
                const float wv = wr[k];
#pragma unroll
                for (int n = 0; n < DENSE_MAXN; ++n)
                    if (n < nb) acc[n] += wv * xs[n * DENSE_KC + k];
            }
        }
    }
    if (o >= O) return;
    const float bias = b ? b[o] * bgain : 0.f;
#pragma unroll
    for (int n = 0; n < DENSE_MAXN; ++n) {
        if (n >= nb) break;
        float v = acc[n];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) {
            v = v * wgain + bias;
            if (act) v = shg_lrelu_agc(v, alpha, gain, clamp);
            y[(long)(n0 + n) * ldy + o] = v;
        }
    }
}

// x: [N, K] with row pitch ldx; w: [O, K]; y: [N, O] with row pitch ldy.
extern "C" int shg_dense_f32(const float* x, const float* w, const float* b, float* y, int N, int K, int O, int ldx, int ldy,
                             float wgain, float bgain, int act, float alpha, float gain, float clamp, void* stream) {
    SHG_CHECK_ARG(x && w && y, "dense: null pointer");
    SHG_CHECK_ARG(N >= 1 && K >= 1 && O >= 1 && ldx >= K && ldy >= O, "dense: bad shape");
    dim3 grid(shg_cdiv(O, 4), shg_cdiv(N, DENSE_MAXN));
    const size_t lds = sizeof(float) * DENSE_MAXN * DENSE_KC;
    hipLaunchKernelGGL(dense_kernel, grid, dim3(256), lds, (hipStream_t)stream, x, w, b, y, N, K, O, ldx, ldy, wgain, bgain, act,
                       alpha, gain, clamp);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// y[n,:] = x[n,:] * rsqrt(mean(x[n,:]^2) + eps)
__global__ __launch_bounds__(256) void normalize_2nd_moment_kernel(const float* x, float* y, int K, float eps) {
    __shared__ float red[256];
    const int n = blockIdx.x;
    float acc = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) { const float v = x[(long)n * K + k]; acc += v * v; }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    const float r = rsqrtf(red[0] / (float)K + eps);
    for (int k = threadIdx.x; k < K; k += 256) y[(long)n * K + k] = x[(long)n * K + k] * r;
}

extern "C" int shg_normalize_2nd_moment_f32(const float* x, float* y, int N, int K, float eps, void* stream) {
    SHG_CHECK_ARG(x && y && N >= 1 && K >= 1, "normalize_2nd_moment: bad arguments");
    hipLaunchKernelGGL(normalize_2nd_moment_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, x, y, K, eps);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// One block per sample.  styles [N,I] (row pitch lds_), s_out [N,I], dcoef [N,O] (may be null when !demod).
__global__ __launch_bounds__(256) void modconv_style_prep_kernel(const float* styles, int ld, const float* wsq, float* s_out,
                                                                 float* dcoef, int N, int I, int O, int OP, int demod,
                                                                 float pre_gain) {
    extern __shared__ float s2[];   // [I] squared normalised styles
    __shared__ float red[256];
    const int n = blockIdx.x;
    float snorm = 1.f;
    if (demod) {
        float acc = 0.f;
        for (int e = threadIdx.x; e < N * I; e += 256) {
            const float v = styles[(long)(e / I) * ld + (e % I)] * pre_gain;
            acc += v * v;
        }
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        snorm = rsqrtf(red[0] / (float)(N * I));
    }
    for (int i = threadIdx.x; i < I; i += 256) {
        const float v = styles[(long)n * ld + i] * pre_gain * snorm;
        s_out[(long)n * I + i] = v;
        s2[i] = v * v;
    }
    if (!demod || !dcoef) return;
    __syncthreads();
    for (int o = threadIdx.x; o < O; o += 256) {
        float acc = 0.f;
        for (int i = 0; i < I; ++i) acc += s2[i] * wsq[(long)i * OP + o];
        dcoef[(long)n * O + o] = rsqrtf(acc + 1e-8f);
    }
}

extern "C" int shg_modconv_style_prep_f32(const float* styles, int ld, const float* wsq, float* s_out, float* dcoef, int N, int I,
                                          int O, int OP, int demod, float pre_gain, void* stream) {
    SHG_CHECK_ARG(styles && s_out, "style_prep: null pointer");
    SHG_CHECK_ARG(!demod || (wsq && dcoef), "style_prep: demodulation needs wsq and dcoef");
    SHG_CHECK_ARG(N >= 1 && I >= 1 && ld >= I, "style_prep: bad shape");
    SHG_CHECK_ARG((size_t)I * 4 <= 64 * 1024, "style_prep: I too large");
    hipLaunchKernelGGL(modconv_style_prep_kernel, dim3(N), dim3(256), sizeof(float) * I, (hipStream_t)stream, styles, ld, wsq,
                       s_out, dcoef, N, I, O, OP, demod, pre_gain);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}
