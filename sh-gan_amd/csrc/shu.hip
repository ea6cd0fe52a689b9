// Spectral Hint Unit kernels (gfx950): rFFT2 + row shift, and band-weighted sum + Gaussian split +
// row unshift + irFFT2 at five resolutions.  Reference: lib/model_zoo/shgan.py:312-336 (SHU.forward),
// :143-160 (heterogeneous_filter), :281-310 (Gaussian-split tables); index conventions in
// SURVEY.md appendix B.  The two 1x1 convolutions in between (64->64 +bias+ReLU, 64->384) run on the
// MFMA implicit-GEMM kernel (conv_mfma.hip) as single-tap convolutions.
//
// The transform is always 64x64 (shu_input_res = 64), so a plane fits in LDS many times over: one
// workgroup owns one (sample, channel) plane, keeps the whole spectrum in LDS and never goes back to
// HBM between the row pass, the column pass and the five inverse transforms.  At N = 64 a direct
// DFT with an LDS twiddle table (64 complex MACs per output) costs < 1 MFLOP per plane.
#include "shg_common.h"

#define SHU_N 64          // transform size (shu_input_res)
#define SHU_NH 33         // half spectrum width

__device__ __forceinline__ void shu_build_twiddles(float2* tw) {
    if (threadIdx.x < SHU_N) {
        float s, c;
        sincospif((float)threadIdx.x / 32.0f, &s, &c);   // angle = 2*pi*m/64
        tw[threadIdx.x] = make_float2(c, s);
    }
}

// x: channel planes [C][64][64] per sample at x + n*xbs.  T: [N, 2C, 64, 33]; ch c = Re, ch C+c = Im,
// rows shifted so DC sits on row 31 (shgan.py:313-319), scaled by 1/4096 (norm='forward').
__global__ __launch_bounds__(256) void shu_rfft2_shift_kernel(const float* x, long xbs, float* T, int C) {
    __shared__ float xs[SHU_N][SHU_N + 1];
    __shared__ float2 R[SHU_N][SHU_NH];
    __shared__ float2 tw[SHU_N];
    const int c = blockIdx.x, n = blockIdx.y;
    shu_build_twiddles(tw);
    const float* xp = x + (long)n * xbs + (long)c * SHU_N * SHU_N;
    for (int e = threadIdx.x; e < SHU_N * SHU_N; e += 256) xs[e >> 6][e & 63] = xp[e];
    __syncthreads();
    // real -> half-complex along w
    for (int e = threadIdx.x; e < SHU_N * SHU_NH; e += 256) {
        const int h = e / SHU_NH, k = e - h * SHU_NH;
        float re = 0.f, im = 0.f;
#pragma unroll 8
        for (int w = 0; w < SHU_N; ++w) {
            const float2 t = tw[(k * w) & 63];
            const float v = xs[h][w];
            re += v * t.x;
            im -= v * t.y;
        }
        R[h][k] = make_float2(re, im);
    }
    __syncthreads();
    // complex DFT along h, scale, shift rows, split re/im into channels
    const float sc = 1.0f / (SHU_N * SHU_N);
    for (int e = threadIdx.x; e < SHU_N * SHU_NH; e += 256) {
        const int u = e / SHU_NH, k = e - u * SHU_NH;
        float re = 0.f, im = 0.f;
#pragma unroll 8
        for (int h = 0; h < SHU_N; ++h) {
            const float2 t = tw[(u * h) & 63];
            const float2 v = R[h][k];
            re += v.x * t.x + v.y * t.y;     // (a+bi)(c - si)
            im += v.y * t.x - v.x * t.y;
        }
        const int r = (u + 31) & 63;
        T[(((long)n * 2 * C + c) * SHU_N + r) * SHU_NH + k] = re * sc;
        T[(((long)n * 2 * C + C + c) * SHU_N + r) * SHU_NH + k] = im * sc;
    }
}

extern "C" int shg_shu_rfft2_shift_f32(const float* x, long x_batch_stride, float* T, int N, int C, void* stream) {
    SHG_CHECK_ARG(x && T, "shu_rfft2: null pointer");
    SHG_CHECK_ARG(N >= 1 && N <= 65535 && C >= 1, "shu_rfft2: bad shape");
    hipLaunchKernelGGL(shu_rfft2_shift_kernel, dim3(C, N), dim3(256), 0, (hipStream_t)stream, x, x_batch_stride, T, C);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

struct ShuSplitParams {
    const float* Y;        // [N, 2C*B, 64, 33] when B > 1 (bands to reduce with cw) or [N, 2C, 64, 33] when B == 1
    const float* cw;       // [B, 64, 33]
    const float* gauss[5]; // level l: [r, r/2+1], r = 4 << l
    float* out[5];         // level l: planes [C][r][r] per sample at out[l] + n*obs[l]
    long obs[5];
    int C, B, accumulate;
};

__global__ __launch_bounds__(256) void shu_split_irfft2_kernel(const ShuSplitParams p) {
    __shared__ float2 S[SHU_N][SHU_NH];
    __shared__ float2 Z[SHU_N][SHU_NH];
    __shared__ float2 tw[SHU_N];
    const int c = blockIdx.x, n = blockIdx.y;
    const int C = p.C, B = p.B;
    shu_build_twiddles(tw);
    // heterogeneous band sum: flat conv-output channel = o*B + k  (shgan.py:157-160)
    const long plane = SHU_N * SHU_NH;
    const float* yre = p.Y + ((long)n * 2 * C + c) * B * plane;
    const float* yim = p.Y + ((long)n * 2 * C + C + c) * B * plane;
    for (int e = threadIdx.x; e < plane; e += 256) {
        float re = 0.f, im = 0.f;
        for (int k = 0; k < B; ++k) {
            const float wgt = B > 1 ? p.cw[k * plane + e] : 1.f;
            re += yre[k * plane + e] * wgt;
            im += yim[k * plane + e] * wgt;
        }
        S[e / SHU_NH][e % SHU_NH] = make_float2(re, im);
    }
    __syncthreads();
    for (int l = 0; l < 5; ++l) {
        const int r = 4 << l, rh = r / 2 + 1, tstep = SHU_N / r;
        if (!p.out[l]) continue;
        const float* g = p.gauss[l];
        // complex inverse DFT over rows of the cropped, weighted, un-shifted block (shgan.py:328-334)
        for (int e = threadIdx.x; e < r * rh; e += 256) {
            const int y = e / rh, w = e - y * rh;
            float re = 0.f, im = 0.f;
            for (int j = 0; j < r; ++j) {
                const int a = (j + r / 2 - 1) & (r - 1);          // row inside the crop
                const float2 v = S[SHU_N / 2 - r / 2 + a][w];
                const float gw = g[a * rh + w];
                const float2 t = tw[(j * y * tstep) & 63];
                re += gw * (v.x * t.x - v.y * t.y);               // (a+bi)(c + si)
                im += gw * (v.x * t.y + v.y * t.x);
            }
            Z[y][w] = make_float2(re, im);
        }
        __syncthreads();
        // half-complex -> real along x (imaginary parts of the DC and Nyquist bins are ignored, as in c2r)
        float* op = p.out[l] + (long)n * p.obs[l] + (long)c * r * r;
        for (int e = threadIdx.x; e < r * r; e += 256) {
            const int y = e / r, x = e - y * r;
            float v = Z[y][0].x + ((x & 1) ? -Z[y][r / 2].x : Z[y][r / 2].x);
            for (int w = 1; w < r / 2; ++w) {
                const float2 t = tw[(w * x * tstep) & 63];
                v += 2.f * (Z[y][w].x * t.x - Z[y][w].y * t.y);
            }
            op[e] = p.accumulate ? op[e] + v : v;
        }
        __syncthreads();
    }
}

// gauss / out / out_batch_stride: arrays of 5 entries for r = 4, 8, 16, 32, 64 (out[l] may be null to skip).
extern "C" int shg_shu_split_irfft2_f32(const float* Y, const float* cw, const float* const* gauss, float* const* out,
                                        const long* out_batch_stride, int N, int C, int bands, int accumulate, void* stream) {
    SHG_CHECK_ARG(Y && gauss && out && out_batch_stride, "shu_split: null pointer");
    SHG_CHECK_ARG(bands >= 1 && (bands == 1 || cw), "shu_split: cw required when bands > 1");
    SHG_CHECK_ARG(N >= 1 && N <= 65535 && C >= 1, "shu_split: bad shape");
    ShuSplitParams p;
    p.Y = Y; p.cw = cw; p.C = C; p.B = bands; p.accumulate = accumulate;
    for (int l = 0; l < 5; ++l) {
        p.gauss[l] = gauss[l]; p.out[l] = out[l]; p.obs[l] = out_batch_stride[l];
        SHG_CHECK_ARG(!out[l] || gauss[l], "shu_split: missing gaussian table for level %d", l);
    }
    hipLaunchKernelGGL(shu_split_irfft2_kernel, dim3(C, N), dim3(256), 0, (hipStream_t)stream, p);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}
