// Skinny fully-connected layers and the style/demodulation preparation (gfx950).
//   * dense            : y = act(x @ (W*wgain)^T + b*bgain)              (stylegan.py:87-98)
//   * normalize_2nd_moment                                               (stylegan.py:343-344)
//   * modconv_style_prep: s = styles * rsqrt(mean(styles^2)) (batch-global, stylegan.py:147),
//                         dcoef[n,o] = rsqrt(sum_i s[n,i]^2 * wsq[i,o] + 1e-8)   (stylegan.py:150-155)
// The batch is tiny (N = 16..32) and every weight is read exactly once, so these are HBM/L2
// streaming kernels: one wave per output feature walks a weight row with coalesced 256 B reads
// while the activations of the whole batch sit in LDS.
#include "shg_common.h"
#include "../../include/shgan_hip.h"

#define DENSE_MAXN 16     // samples per pass (larger batches are processed in slabs of DENSE_MAXN rows)

// One wave per output feature: lane l walks k = l, l+64, ... of the weight row (coalesced 256 B) and of the
// DENSE_MAXN activation rows (the whole activation matrix is a few hundred KB and stays in L2); all loads
// of an iteration are independent, so the compiler keeps 1 + N of them in flight per lane.
// MAXN = 4 / 8 / 16 samples per pass (the smallest that holds the batch; larger batches in slabs of 16): every activation load is unconditional --
// rows past the batch re-read the last row.  (A run-time `n < nb` guard around the loads made the batch-8 training calls 4x slower than the
// batch-16 inference calls: 57 vs 15 us.)
template <int MAXN>
__global__ __launch_bounds__(256) void dense_kernel(const float* x, const float* w, const float* b, float* y, int N, int K, int O,
                                                    int ldx, int ldy, float wgain, float bgain, int act, float alpha, float gain,
                                                    float clamp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int o = blockIdx.x * 4 + wave;
    if (o >= O) return;
    const int n0 = blockIdx.y * MAXN;
    const int nb = min(MAXN, N - n0);
    const float* wr = w + (long)o * K;
    const float* xr = x + (long)n0 * ldx;
    long roff[MAXN];
#pragma unroll
    for (int n = 0; n < MAXN; ++n) roff[n] = (long)(n < nb ? n : nb - 1) * ldx;
    float acc[MAXN];
#pragma unroll
    for (int n = 0; n < MAXN; ++n) acc[n] = 0.f;
#pragma unroll 2
    for (int k = lane; k < K; k += 64) {
        const float wv = wr[k];
#pragma unroll
        for (int n = 0; n < MAXN; ++n) acc[n] += wv * xr[roff[n] + k];
    }
    const float bias = b ? b[o] * bgain : 0.f;
#pragma unroll
    for (int n = 0; n < MAXN; ++n) {
        float v = acc[n];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0 && n < nb) {
            v = v * wgain + bias;
            v = act ? shg_lrelu_agc(v, alpha, gain, clamp) : v * gain;
            y[(long)(n0 + n) * ldy + o] = v;
        }
    }
}

// x: [N, K] with row pitch ldx; w: [O, K]; y: [N, O] with row pitch ldy.
extern "C" int shg_dense_f32(const float* x, const float* w, const float* b, float* y, int N, int K, int O, int ldx, int ldy,
                             float wgain, float bgain, int act, float alpha, float gain, float clamp, void* stream) {
    SHG_CHECK_ARG(x && w && y, "dense: null pointer");
    SHG_CHECK_ARG(N >= 1 && K >= 1 && O >= 1 && ldx >= K && ldy >= O, "dense: bad shape");
    if (N <= 4)
        hipLaunchKernelGGL(dense_kernel<4>, dim3(shg_cdiv(O, 4), 1), dim3(256), 0, (hipStream_t)stream, x, w, b, y, N, K, O, ldx, ldy, wgain, bgain,
                           act, alpha, gain, clamp);
    else if (N <= 8)
        hipLaunchKernelGGL(dense_kernel<8>, dim3(shg_cdiv(O, 4), 1), dim3(256), 0, (hipStream_t)stream, x, w, b, y, N, K, O, ldx, ldy, wgain, bgain,
                           act, alpha, gain, clamp);
    else
        hipLaunchKernelGGL(dense_kernel<DENSE_MAXN>, dim3(shg_cdiv(O, 4), shg_cdiv(N, DENSE_MAXN)), dim3(256), 0, (hipStream_t)stream, x, w, b, y, N, K,
                           O, ldx, ldy, wgain, bgain, act, alpha, gain, clamp);
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

// ---- the two other products of a dense layer's gradient (training rows: the transposed forms of stylegan.py:87-98 under autograd).
// out[N,K] = scale * a[N,M] @ b[M,K]: block = 64 output columns x MATW slices of M (one wave each), MATN batch rows in registers per
// pass; every b element is read once per batch slab, the a values are wave-uniform.  Partial sums of the waves meet in LDS.
#define MATN 8
#define MATW 16           // waves per block = slices of M (a 1536-long sum becomes 96 steps per wave: the kernel is latency-bound, not bandwidth-bound)
__global__ __launch_bounds__(64 * MATW) void matmul_nn_kernel(const float* a, const float* b, float* out, int N, int M, int K, int lda, int ldo,
                                                             float scale) {
    __shared__ float red[MATW][MATN][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (provably uniform: the a values become scalar loads)
    const int k = blockIdx.x * 64 + lane;
    const int n0 = blockIdx.y * MATN;
    const int nb = min(MATN, N - n0);
    const int kc = k < K ? k : K - 1;
    const int per = (M + MATW - 1) / MATW, m0 = wave * per, m1 = min(M, m0 + per);
    const float* ar[MATN];
#pragma unroll
    for (int n = 0; n < MATN; ++n) ar[n] = a + (long)(n0 + (n < nb ? n : nb - 1)) * lda;
    float acc[MATN];
#pragma unroll
    for (int n = 0; n < MATN; ++n) acc[n] = 0.f;
#pragma unroll 8
    for (int m = m0; m < m1; ++m) {
        const float bv = b[(long)m * K + kc];
#pragma unroll
        for (int n = 0; n < MATN; ++n) acc[n] += ar[n][m] * bv;
    }
#pragma unroll
    for (int n = 0; n < MATN; ++n) red[wave][n][lane] = acc[n];
    __syncthreads();
    for (int e = threadIdx.x; e < MATN * 64; e += 64 * MATW) {
        const int n = e >> 6, l = e & 63, kk = blockIdx.x * 64 + l;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < MATW; ++w) v += red[w][n][l];
        if (n < nb && kk < K) out[(long)(n0 + n) * ldo + kk] = scale * v;
    }
}

// out[M,K] = scale * a[N,M]^T @ b[N,K] (the weight gradient: a sum over the batch) and, optionally, colsum[m] = csum_scale * sum_n a[n,m]
// (the bias gradient of the same layer): a thread owns 4 rows x 1 column of `out`, walks the batch.
__global__ __launch_bounds__(256) void matmul_tn_kernel(const float* a, const float* b, float* out, float* colsum, int N, int M, int K, int lda,
                                                       int ldb, float scale, float csum_scale) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int m0 = blockIdx.y * 4;
    const int kc = k < K ? k : K - 1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, cs[4] = {0.f, 0.f, 0.f, 0.f};
    int mm[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) mm[r] = m0 + r < M ? m0 + r : M - 1;
#pragma unroll 4
    for (int n = 0; n < N; ++n) {
        const float bv = b[(long)n * ldb + kc];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float av = a[(long)n * lda + mm[r]];
            acc[r] += av * bv;
            cs[r] += av;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (m0 + r < M && k < K) out[(long)(m0 + r) * K + k] = scale * acc[r];
        if (colsum && m0 + r < M && k == 0) colsum[m0 + r] = csum_scale * cs[r];
    }
}

extern "C" int shg_matmul_nn_f32(const float* a, const float* b, float* out, int N, int M, int K, int lda, int ldo, float scale, void* stream) {
    SHG_CHECK_ARG(a && b && out, "matmul_nn: null pointer");
    SHG_CHECK_ARG(N >= 1 && M >= 1 && K >= 1 && lda >= M && ldo >= K, "matmul_nn: bad shape");
    hipLaunchKernelGGL(matmul_nn_kernel, dim3(shg_cdiv(K, 64), shg_cdiv(N, MATN)), dim3(64 * MATW), 0, (hipStream_t)stream, a, b, out, N, M, K, lda, ldo, scale);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

extern "C" int shg_matmul_tn_f32(const float* a, const float* b, float* out, float* colsum, int N, int M, int K, int lda, int ldb, float scale,
                                 float csum_scale, void* stream) {
    SHG_CHECK_ARG(a && b && out, "matmul_tn: null pointer");
    SHG_CHECK_ARG(N >= 1 && M >= 1 && K >= 1 && lda >= M && ldb >= K, "matmul_tn: bad shape");
    hipLaunchKernelGGL(matmul_tn_kernel, dim3(shg_cdiv(K, 256), shg_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, a, b, out, colsum, N, M, K, lda, ldb,
                       scale, csum_scale);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// ---- weight side of modulated_conv2d under autograd (stylegan.py:136-138,146,150-155): one launch instead of ~12 tensor ops forward and
// ~25 backward per layer.  Per output channel o (one workgroup):
//   w1 = w * c,  c = 1 / (sqrt(I K) max|w[o]|) when `prenorm` (the fp16 pre-normalisation), else w1 = w
//   wn = w1 * rsqrt(mean(w1^2)),      wsq[o,i] = sum_k wn[o,i,k]^2,      sfac[o] = wn / w   (saved for the backward pass)
// The pre-normalisation cancels in wn up to round-off (and exactly in its derivative), so the backward needs sfac only:
//   G = g_wn + 2 g_wsq[o,i] wn,      g_w = sfac * (G - wn * mean(G wn)).
__device__ __forceinline__ float dw_block_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float dw_block_max(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void demod_weight_kernel(const float* w, float* wn, float* wsq, float* sfac, int I, int K, int prenorm) {
    __shared__ float red[4];
    const int o = blockIdx.x, IK = I * K;
    const float* wo = w + (long)o * IK;
    float c = 1.f;
    if (prenorm) {
        float mx = 0.f;
        for (int e = threadIdx.x; e < IK; e += 256) mx = fmaxf(mx, fabsf(wo[e]));
        mx = dw_block_max(mx, red);
        c = 1.f / sqrtf((float)IK) / mx;
    }
    float ss = 0.f;
    for (int e = threadIdx.x; e < IK; e += 256) { const float v = wo[e] * c; ss += v * v; }
    ss = dw_block_sum(ss, red);
    const float r = rsqrtf(ss / (float)IK);
    for (int e = threadIdx.x; e < IK; e += 256) wn[(long)o * IK + e] = wo[e] * c * r;
    if (threadIdx.x == 0) sfac[o] = c * r;
    for (int i = threadIdx.x; i < I; i += 256) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) { const float v = wo[i * K + k] * c * r; acc += v * v; }
        wsq[(long)o * I + i] = acc;
    }
}

__global__ __launch_bounds__(256) void demod_weight_backward_kernel(const float* wn, const float* sfac, const float* gwn, const float* gwsq, float* gw,
                                                                   int I, int K) {
    __shared__ float red[4];
    const int o = blockIdx.x, IK = I * K;
    const float* u = wn + (long)o * IK;
    float dot = 0.f;
    for (int e = threadIdx.x; e < IK; e += 256) {
        const float uu = u[e];
        const float g = (gwn ? gwn[(long)o * IK + e] : 0.f) + (gwsq ? 2.f * gwsq[(long)o * I + e / K] * uu : 0.f);
        dot += g * uu;
    }
    dot = dw_block_sum(dot, red) / (float)IK;
    const float s = sfac[o];
    for (int e = threadIdx.x; e < IK; e += 256) {
        const float uu = u[e];
        const float g = (gwn ? gwn[(long)o * IK + e] : 0.f) + (gwsq ? 2.f * gwsq[(long)o * I + e / K] * uu : 0.f);
        gw[(long)o * IK + e] = s * (g - uu * dot);
    }
}

extern "C" int shg_demod_weight_f32(const float* w, float* wn, float* wsq, float* sfac, int O, int I, int K, int prenorm, void* stream) {
    SHG_CHECK_ARG(w && wn && wsq && sfac && O >= 1 && I >= 1 && K >= 1, "demod_weight: bad arguments");
    hipLaunchKernelGGL(demod_weight_kernel, dim3(O), dim3(256), 0, (hipStream_t)stream, w, wn, wsq, sfac, I, K, prenorm);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// gwn [O,I,K] and gwsq [O,I]: either may be NULL (no gradient arrived on that output)
extern "C" int shg_demod_weight_backward_f32(const float* wn, const float* sfac, const float* gwn, const float* gwsq, float* gw, int O, int I, int K,
                                             void* stream) {
    SHG_CHECK_ARG(wn && sfac && gw && (gwn || gwsq) && O >= 1 && I >= 1 && K >= 1, "demod_weight_backward: bad arguments");
    hipLaunchKernelGGL(demod_weight_backward_kernel, dim3(O), dim3(256), 0, (hipStream_t)stream, wn, sfac, gwn, gwsq, gw, I, K);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// ---- style side of modulated_conv2d under autograd (stylegan.py:138,147,155): sn = s1 * rsqrt(mean(s1^2)) with s1 = s / max_i |s| per
// row when `prenorm` (the fp16 pre-normalisation) else s, and dcoef[n][o] = rsqrt(sum_i sn[n][i]^2 wsq[o][i] + 1e-8).  ~11 tensor ops
// forward and ~25 backward per layer and pass in the composed form; here one launch forward, two backward.  All three are latency
// problems (a few hundred kFLOP): many small workgroups, every global load of a thread issued before the first use.
// Forward: grid = ceil(O / 8) workgroups (a wave owns 2 output channels); every workgroup recomputes the (tiny) batch-global statistic,
// workgroup 0 writes sn and aux = {M_n (row maxima, 1 without prenorm), r}.  LDS: q = sn^2 [N][I], then the workgroup's 8 wsq rows.
// The loops stay rolled: straight-line code that runs once costs more in instruction fetch than it saves (25 -> see MEASUREMENTS).
template <int NB>
__global__ __launch_bounds__(256) void style_factors_kernel(const float* s, const float* wsq, float* sn, float* d, float* aux, int N, int I, int O,
                                                            int prenorm) {
    extern __shared__ float q[];
    __shared__ float red[256];
    __shared__ float rowmax[NB];                  // row maxima (1 without prenorm)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* wl = q + N * I;                        // [8][I]
    // every global load of the kernel is requested up front: flat, independent (a row-by-row walk pays one memory latency per step)
#pragma unroll 8
    for (int e = tid; e < N * I; e += 256) q[e] = s[e];
    if (d) {
        const int rows = min(8, O - (int)blockIdx.x * 8);
        const float* wr = wsq + (long)blockIdx.x * 8 * I;
#pragma unroll 8
        for (int e = tid; e < rows * I; e += 256) wl[e] = wr[e];
    }
    __syncthreads();
    for (int n = wave; n < N; n += 4) {
        float mx = 0.f;
        if (prenorm)
            for (int i = lane; i < I; i += 64) mx = fmaxf(mx, fabsf(q[n * I + i]));
        for (int st = 32; st > 0; st >>= 1) mx = fmaxf(mx, __shfl_xor(mx, st));
        if (lane == 0) {
            rowmax[n] = prenorm ? mx : 1.f;
            if (blockIdx.x == 0) aux[n] = prenorm ? mx : 1.f;
        }
    }
    __syncthreads();
    float acc = 0.f;
    for (int n = wave; n < N; n += 4) {
        // (divided by, not multiplied with the inverse: the row's maximum must become exactly +-1 -- the backward pass finds it again as |sn| == r)
        const float m = rowmax[n];
        for (int i = lane; i < I; i += 64) {
            const float v = q[n * I + i] / m;
            q[n * I + i] = v;
            acc += v * v;
        }
    }
    red[tid] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    const float r = rsqrtf(red[0] / (float)(N * I));
    if (blockIdx.x == 0 && tid == 0) aux[N] = r;
    for (int e = tid; e < N * I; e += 256) {
        const float v = q[e] * r;
        if (blockIdx.x == 0) sn[e] = v;
        q[e] = v * v;
    }
    __syncthreads();
    if (!d) return;
#pragma unroll 1
    for (int k = 0; k < 2; ++k) {
        const int ol = wave * 2 + k, o = blockIdx.x * 8 + ol;
        if (o >= O) break;
        float a[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) a[n] = 0.f;
#pragma unroll 1
        for (int i = lane; i < I; i += 64) {
            const float wv = wl[ol * I + i];
#pragma unroll
            for (int n = 0; n < NB; ++n)
                if (n < N) a[n] += q[n * I + i] * wv;
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            float v = a[n];
            for (int st = 32; st > 0; st >>= 1) v += __shfl_xor(v, st);
            if (lane == 0 && n < N) d[(long)n * O + o] = rsqrtf(v + 1e-8f);
        }
    }
}

// Backward, pass 1: grid = (ceil(I / 64), ceil(O / 64)); workgroup (ib, ob) owns 64 style columns x 64 output channels, a wave 16 of the
// channels.  gt = -1/2 gd d^3 for its channels in LDS; gwsq[o][i] = sum_n gt[n][o] sn[n][i]^2 is written directly (every (o, i) has one
// owner); its share of gq[n][i] = sum_o gt[n][o] wsq[o][i] goes to partq[ob][n][i].
template <int NB>
__global__ __launch_bounds__(256) void style_factors_backward1_kernel(const float* sn, const float* d, const float* wsq, const float* gd, float* gwsq,
                                                                      float* partq, int N, int I, int O) {
    __shared__ float gt[NB * 64];
    __shared__ float pq[4 * NB * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * 64 + lane, ob = blockIdx.y * 64;
    const bool in = i < I;
    for (int e = tid; e < N * 64; e += 256) {
        const int n = e >> 6, o = ob + (e & 63);
        float v = 0.f;
        if (gd && o < O) {
            const float dv = d[(long)n * O + o];
            v = -0.5f * gd[(long)n * O + o] * dv * dv * dv;
        }
        gt[e] = v;
    }
    float q[NB], acc[NB], wv[16];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const float v = (in && n < N) ? sn[(long)n * I + i] : 0.f;
        q[n] = v * v;
        acc[n] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int o = ob + wave * 16 + k;
        wv[k] = (in && o < O) ? wsq[(long)o * I + i] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int ol = wave * 16 + k, o = ob + ol;
        float gw = 0.f;
#pragma unroll
        for (int n = 0; n < NB; ++n)
            if (n < N) {
                const float g = gt[n * 64 + ol];
                acc[n] += g * wv[k];
                gw += g * q[n];
            }
        if (in && o < O && gwsq) gwsq[(long)o * I + i] = gw;
    }
#pragma unroll
    for (int n = 0; n < NB; ++n) pq[(wave * NB + n) * 64 + lane] = acc[n];
    __syncthreads();
    for (int e = tid; e < N * 64; e += 256) {
        const int n = e >> 6, l = e & 63, ii = blockIdx.x * 64 + l;
        if (ii < I)
            partq[((long)blockIdx.y * N + n) * I + ii] = pq[n * 64 + l] + pq[(NB + n) * 64 + l] + pq[(2 * NB + n) * 64 + l] + pq[(3 * NB + n) * 64 + l];
    }
}

// Backward, pass 2: workgroup n finishes row n.  g_tot = gsn + 2 sn gq for ALL rows (LDS; every workgroup needs the batch-global c).
// With A_n = (1/r) sum_i g_tot sn, c = sum_n A_n, B_n = sum_i sn^2:
//   g1 = r g_tot - (r^2 c / (N I)) sn                              (through sn = s1 * rsqrt(mean s1^2))
//   gs = g1 / M_n - [|sn_i| == r] sign(sn_i) R_n / (M_n ties),  R_n = r A_n - (r c / (N I)) B_n        (through s1 = s / max|s|, prenorm only)
__global__ __launch_bounds__(256) void style_factors_backward2_kernel(const float* sn, const float* aux, const float* gsn, const float* partq, float* gs,
                                                                      int N, int I, int S, int prenorm) {
    extern __shared__ float gtot[];                // [N][I] g_tot, then [N][I] sn
    __shared__ float A[32];
    __shared__ float red[256];
    __shared__ int redi[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = blockIdx.x;
    float* snl = gtot + N * I;
    const float r = aux[N], M = aux[n];
    const int NI = N * I;
    for (int e0 = tid; e0 < NI; e0 += 1024) {      // batches of 4 elements x 8 slices: 40 independent loads requested before the first use
        float pv[4][8], sv[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * 256;
            sv[u] = e < NI ? sn[e] : 0.f;
            gv[u] = (gsn && e < NI) ? gsn[e] : 0.f;
#pragma unroll
            for (int sb = 0; sb < 8; ++sb) pv[u][sb] = (e < NI && sb < S) ? partq[(long)sb * NI + e] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * 256;
            float gq = ((pv[u][0] + pv[u][1]) + (pv[u][2] + pv[u][3])) + ((pv[u][4] + pv[u][5]) + (pv[u][6] + pv[u][7]));
            for (int sb = 8; sb < S; ++sb) gq += e < NI ? partq[(long)sb * NI + e] : 0.f;       // (O > 512)
            if (e < NI) {
                snl[e] = sv[u];
                gtot[e] = gv[u] + 2.f * sv[u] * gq;
            }
        }
    }
    __syncthreads();
    for (int k = wave; k < N; k += 4) {
        float a = 0.f;
        for (int i = lane; i < I; i += 64) a += gtot[k * I + i] * snl[k * I + i];
        for (int st = 32; st > 0; st >>= 1) a += __shfl_xor(a, st);
        if (lane == 0) A[k] = a / r;
    }
    __syncthreads();
    float c = 0.f;
    for (int k = 0; k < N; ++k) c += A[k];
    const float ni = (float)N * (float)I, coef = r * r * c / ni;
    float Rn = 0.f;
    int ties = 1;
    if (prenorm) {
        float b = 0.f;
        int t = 0;
        for (int i = tid; i < I; i += 256) {
            const float v = snl[n * I + i];
            b += v * v;
            t += (fabsf(v) == r) ? 1 : 0;
        }
        red[tid] = b; redi[tid] = t;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) { red[tid] += red[tid + st]; redi[tid] += redi[tid + st]; }
            __syncthreads();
        }
        Rn = r * A[n] - (r * c / ni) * red[0];
        ties = redi[0] > 0 ? redi[0] : 1;
    }
    for (int i = tid; i < I; i += 256) {
        const float v = snl[n * I + i];
        float g = r * gtot[n * I + i] - coef * v;
        if (prenorm) {
            g /= M;
            if (fabsf(v) == r) g -= (v > 0.f ? 1.f : -1.f) * Rn / (M * (float)ties);
        }
        gs[(long)n * I + i] = g;
    }
}

extern "C" int shg_style_factors_f32(const float* s, const float* wsq, float* sn, float* d, float* aux, int N, int I, int O, int prenorm, void* stream) {
    SHG_CHECK_ARG(s && sn && aux && N >= 1 && N <= 32 && I >= 1 && I <= 1024 && (long)N * I <= 8192 && (!d || (wsq && O >= 1)), "style_factors: N <= 32, N * I <= 8192, I <= 1024");
    const dim3 grid(d ? (O + 7) / 8 : 1);
    const size_t lds = ((size_t)N * I + (d ? 8 * (size_t)I : 0)) * sizeof(float);          // <= 32 KB + 32 KB
    if (N <= 8) hipLaunchKernelGGL((style_factors_kernel<8>), grid, dim3(256), lds, (hipStream_t)stream, s, wsq, sn, d, aux, N, I, O, prenorm);
    else if (N <= 16) hipLaunchKernelGGL((style_factors_kernel<16>), grid, dim3(256), lds, (hipStream_t)stream, s, wsq, sn, d, aux, N, I, O, prenorm);
    else hipLaunchKernelGGL((style_factors_kernel<32>), grid, dim3(256), lds, (hipStream_t)stream, s, wsq, sn, d, aux, N, I, O, prenorm);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// gs [N,I], gwsq [O,I] (may be null) from gsn [N,I] / gd [N,O] (either may be null); partq: scratch of ceil(O/64) * N * I floats
extern "C" int shg_style_factors_backward_f32(const float* sn, const float* d, const float* wsq, const float* aux, const float* gsn, const float* gd,
                                              float* gs, float* gwsq, float* partq, int N, int I, int O, int prenorm, void* stream) {
    SHG_CHECK_ARG(sn && d && wsq && aux && gs && partq && (gsn || gd) && N >= 1 && N <= 32 && I >= 1 && O >= 1 && (long)N * I <= 8192,
                  "style_factors_backward: N <= 32, N * I <= 8192");
    const dim3 grid((I + 63) / 64, (O + 63) / 64);
    if (N <= 8) hipLaunchKernelGGL((style_factors_backward1_kernel<8>), grid, dim3(256), 0, (hipStream_t)stream, sn, d, wsq, gd, gwsq, partq, N, I, O);
    else if (N <= 16) hipLaunchKernelGGL((style_factors_backward1_kernel<16>), grid, dim3(256), 0, (hipStream_t)stream, sn, d, wsq, gd, gwsq, partq, N, I, O);
    else hipLaunchKernelGGL((style_factors_backward1_kernel<32>), grid, dim3(256), 0, (hipStream_t)stream, sn, d, wsq, gd, gwsq, partq, N, I, O);
    SHG_CHECK_LAUNCH();
    hipLaunchKernelGGL(style_factors_backward2_kernel, dim3(N), dim3(256), (size_t)2 * N * I * sizeof(float), (hipStream_t)stream, sn, aux, gsn, partq, gs, N, I,
                       (int)grid.y, prenorm);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

extern "C" int shg_normalize_2nd_moment_f32(const float* x, float* y, int N, int K, float eps, void* stream) {
    SHG_CHECK_ARG(x && y && N >= 1 && K >= 1, "normalize_2nd_moment: bad arguments");
    hipLaunchKernelGGL(normalize_2nd_moment_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, x, y, K, eps);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// Block (n, oc): sample n, output channels oc*64 .. oc*64+63.  styles [N,I] (row pitch ld), s_out [N,I],
// dcoef [N,O] (null when !demod).  256 threads = 64 channels x 4 interleaved slices of the I reduction.
__global__ __launch_bounds__(256) void modconv_style_prep_kernel(const float* styles, int ld, const float* wsq, float* s_out,
                                                                 float* dcoef, int N, int I, int O, int OP, int demod,
                                                                 float pre_gain) {
    extern __shared__ float s2[];   // [I] squared normalised styles of sample n
    __shared__ float red[256];
    const int n = blockIdx.x, oc = blockIdx.y;
    float snorm = 1.f;
    if (demod) {      // batch-global RMS of the styles (stylegan.py:147); every block computes the same value
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
        __syncthreads();
    }
    for (int i = threadIdx.x; i < I; i += 256) {
        const float v = styles[(long)n * ld + i] * pre_gain * snorm;
        if (oc == 0) s_out[(long)n * I + i] = v;
        s2[i] = v * v;
    }
    if (!demod || !dcoef) return;
    __syncthreads();
    const int oo = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int o = oc * 64 + oo;
    float acc = 0.f;
    if (o < O)
        for (int i = sl; i < I; i += 4) acc += s2[i] * wsq[(long)i * OP + o];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (sl == 0 && o < O) dcoef[(long)n * O + o] = rsqrtf(red[oo] + red[64 + oo] + red[128 + oo] + red[192 + oo] + 1e-8f);
}

extern "C" int shg_modconv_style_prep_f32(const float* styles, int ld, const float* wsq, float* s_out, float* dcoef, int N, int I,
                                          int O, int OP, int demod, float pre_gain, void* stream) {
    SHG_CHECK_ARG(styles && s_out, "style_prep: null pointer");
    SHG_CHECK_ARG(!demod || (wsq && dcoef), "style_prep: demodulation needs wsq and dcoef");
    SHG_CHECK_ARG(N >= 1 && I >= 1 && ld >= I, "style_prep: bad shape");
    SHG_CHECK_ARG((size_t)I * 4 <= 64 * 1024, "style_prep: I too large");
    hipLaunchKernelGGL(modconv_style_prep_kernel, dim3(N, demod ? shg_cdiv(O, 64) : 1), dim3(256), sizeof(float) * I, (hipStream_t)stream, styles, ld, wsq,
                       s_out, dcoef, N, I, O, OP, demod, pre_gain);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// ------------------------------------------------------------------------------------------------
// Grouped forms: all the style affines of a synthesis pass in one launch, then all their
// normalisation / demodulation coefficients in a second one (23 + 23 + 23 small launches -> 2).
// ------------------------------------------------------------------------------------------------
#define SHG_MAX_GROUPS 32
struct DenseGroupArgs {
    shg_dense_group g[SHG_MAX_GROUPS];
    int first_block[SHG_MAX_GROUPS + 1];   // prefix sum of ceil(O/4)
    int G, N;
};

// Same mapping as dense_kernel (one wave per output feature); the input row is the concatenation [x1 | x2]
// (comodgan.py:245-262,316-338: cat([w_i, x_global])) read from its two sources.
__global__ __launch_bounds__(256) void dense_grouped_kernel(const DenseGroupArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int gi = 0;
    while (gi + 1 < a.G && (int)blockIdx.x >= a.first_block[gi + 1]) ++gi;
    const shg_dense_group& g = a.g[gi];
    const int o = ((int)blockIdx.x - a.first_block[gi]) * 4 + wave;
    if (o >= g.O) return;
    const int n0 = blockIdx.y * DENSE_MAXN;
    const int nb = min(DENSE_MAXN, a.N - n0);
    const int K = g.K1 + g.K2;
    const float* wr = g.w + (long)o * K;
    float acc[DENSE_MAXN];
#pragma unroll
    for (int n = 0; n < DENSE_MAXN; ++n) acc[n] = 0.f;
    for (int seg = 0; seg < 2; ++seg) {
        const float* xr = (seg == 0 ? g.x1 + (long)n0 * g.ld1 : g.x2 + (long)n0 * g.ld2);
        const int ld = seg == 0 ? g.ld1 : g.ld2, Ks = seg == 0 ? g.K1 : g.K2;
        const float* ws = wr + (seg == 0 ? 0 : g.K1);
        if (Ks == 0) continue;
#pragma unroll 2
        for (int k = lane; k < Ks; k += 64) {
            const float wv = ws[k];
#pragma unroll
            for (int n = 0; n < DENSE_MAXN; ++n) acc[n] += wv * xr[(long)min(n, nb - 1) * ld + k];
        }
    }
    const float bias = g.b ? g.b[o] * g.bgain : 0.f;
#pragma unroll
    for (int n = 0; n < DENSE_MAXN; ++n) {
        float v = acc[n];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0 && n < nb) g.y[(long)(n0 + n) * g.ldy + o] = v * g.wgain + bias;
    }
}

extern "C" int shg_dense_grouped_f32(const shg_dense_group* groups, int G, int N, void* stream) {
    SHG_CHECK_ARG(groups && G >= 1 && G <= SHG_MAX_GROUPS && N >= 1, "dense_grouped: bad arguments (at most %d groups)", SHG_MAX_GROUPS);
    DenseGroupArgs a;
    a.G = G; a.N = N; a.first_block[0] = 0;
    for (int i = 0; i < G; ++i) {
        const shg_dense_group& g = groups[i];
        SHG_CHECK_ARG(g.x1 && g.w && g.y && g.K1 >= 1 && g.K2 >= 0 && (g.K2 == 0 || g.x2) && g.O >= 1 && g.ld1 >= g.K1 && g.ld2 >= g.K2 && g.ldy >= g.O,
                      "dense_grouped: bad group %d", i);
        a.g[i] = g;
        a.first_block[i + 1] = a.first_block[i] + shg_cdiv(g.O, 4);
    }
    dim3 grid(a.first_block[G], shg_cdiv(N, DENSE_MAXN));
    hipLaunchKernelGGL(dense_grouped_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

struct StyleGroupArgs {
    shg_style_group g[SHG_MAX_GROUPS];
    int first_block[SHG_MAX_GROUPS + 1];   // prefix sum of N * (demod ? ceil(O/64) : 1)
    int G, N;
};

// modconv_style_prep_kernel for a list of layers: block -> (layer, sample n, 64-channel slab oc)
__global__ __launch_bounds__(256) void modconv_style_prep_grouped_kernel(const StyleGroupArgs a) {
    extern __shared__ float s2[];   // [I] squared normalised styles of sample n
    __shared__ float red[256];
    int gi = 0;
    while (gi + 1 < a.G && (int)blockIdx.x >= a.first_block[gi + 1]) ++gi;
    const shg_style_group& g = a.g[gi];
    const int local = (int)blockIdx.x - a.first_block[gi];
    const int n = local % a.N, oc = local / a.N;
    const int N = a.N, I = g.I;
    float snorm = 1.f;
    if (g.demod) {      // batch-global RMS of the styles (stylegan.py:147)
        float acc = 0.f;
        for (int e = threadIdx.x; e < N * I; e += 256) {
            const float v = g.styles[(long)(e / I) * g.ld + (e % I)] * g.pre_gain;
            acc += v * v;
        }
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        snorm = rsqrtf(red[0] / (float)(N * I));
        __syncthreads();
    }
    for (int i = threadIdx.x; i < I; i += 256) {
        const float v = g.styles[(long)n * g.ld + i] * g.pre_gain * snorm;
        if (oc == 0) g.s_out[(long)n * I + i] = v;
        s2[i] = v * v;
    }
    if (!g.demod || !g.dcoef) return;
    __syncthreads();
    const int oo = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int o = oc * 64 + oo;
    float acc = 0.f;
    if (o < g.O)
        for (int i = sl; i < I; i += 4) acc += s2[i] * g.wsq[(long)i * g.OP + o];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (sl == 0 && o < g.O) g.dcoef[(long)n * g.O + o] = rsqrtf(red[oo] + red[64 + oo] + red[128 + oo] + red[192 + oo] + 1e-8f);
}

extern "C" int shg_modconv_style_prep_grouped_f32(const shg_style_group* groups, int G, int N, void* stream) {
    SHG_CHECK_ARG(groups && G >= 1 && G <= SHG_MAX_GROUPS && N >= 1, "style_prep_grouped: bad arguments (at most %d groups)", SHG_MAX_GROUPS);
    StyleGroupArgs a;
    a.G = G; a.N = N; a.first_block[0] = 0;
    int imax = 1;
    for (int i = 0; i < G; ++i) {
        const shg_style_group& g = groups[i];
        SHG_CHECK_ARG(g.styles && g.s_out && g.I >= 1 && g.ld >= g.I, "style_prep_grouped: bad group %d", i);
        SHG_CHECK_ARG(!g.demod || (g.wsq && g.dcoef && g.O >= 1 && g.OP >= g.O), "style_prep_grouped: group %d: demodulation needs wsq and dcoef", i);
        SHG_CHECK_ARG((size_t)g.I * 4 <= 64 * 1024, "style_prep_grouped: I too large");
        a.g[i] = g;
        a.first_block[i + 1] = a.first_block[i] + N * (g.demod ? shg_cdiv(g.O, 64) : 1);
        if (g.I > imax) imax = g.I;
    }
    hipLaunchKernelGGL(modconv_style_prep_grouped_kernel, dim3(a.first_block[G]), dim3(256), sizeof(float) * imax, (hipStream_t)stream, a);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}
