// FID statistics on the device (reference lib/evaluator/eva_fid.py:251-263): the reference gathers
// every feature vector to rank 0 (3 x world broadcasts per batch, eva_base.py:96-188) and forms mean / covariance there in
// numpy float64.  Here each rank accumulates, in float64 on its GPU, the second-moment matrix of the AUGMENTED feature
// x' = [x, 1]:   S += sum_b w_b x'_b x'_b^T   (S[:D,:D] = sum x x^T, S[D,:D] = sum x, S[D,D] = count),
// so that one all-reduce of S at the end of the evaluation replaces all per-batch feature traffic.
// GEMM-shaped work in double precision -> v_mfma_f64_16x16x4_f64 (A, B: one f64 per lane, A[i = l&15][k = l>>4],
// B[k = l>>4][j = l&15]; D: 4 f64 per lane, column l&15, row (l>>4) + 4*reg).  One workgroup = a 32 x 32 tile of S
// (4 waves, one 16 x 16 sub-tile each), operands straight from global memory (the problem is tiny and runs once per batch).
#include "shg_common.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));

template <bool F64IN>
__global__ __launch_bounds__(256) void fid_accumulate_kernel(const void* feats, const float* weights, double* S, int B, int D, int DP) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = blockIdx.y * 32 + (wave >> 1) * 16, j0 = blockIdx.x * 32 + (wave & 1) * 16;
    if (j0 + 15 < i0) return;                      // lower-triangle sub-tiles are mirrored by the host on read-out
    const int li = lane & 15, lk = lane >> 4;
    auto at = [&](int b, int c) -> double {        // augmented, zero-padded feature matrix [B][DP]
        if (b >= B || c > D) return 0.0;
        if (c == D) return 1.0;
        return F64IN ? reinterpret_cast<const double*>(feats)[(long)b * D + c] : (double)reinterpret_cast<const float*>(feats)[(long)b * D + c];
    };
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int b0 = 0; b0 < B; b0 += 4) {
        const int b = b0 + lk;
        const double w = (weights && b < B) ? (double)weights[b] : 1.0;
        const double a = at(b, i0 + li) * w;
        const double bb = at(b, j0 + li);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = i0 + lk + 4 * r, col = j0 + li;
        if (row < DP && col < DP) S[(long)row * DP + col] += acc[r];
    }
}

// feats [B, D] (float32, or float64 when is_f64), weights [B] or NULL, S [DP, DP] float64 accumulated in place
// (DP >= D + 1, a multiple of 32; only sub-tiles on or above the diagonal are updated).
extern "C" int shg_fid_accumulate_f64(const void* feats, int is_f64, const float* weights, double* S, int B, int D, int DP,
                                      void* stream) {
    SHG_CHECK_ARG(feats && S, "fid_accumulate: null pointer");
    SHG_CHECK_ARG(B >= 1 && D >= 1 && DP >= D + 1 && DP % 32 == 0, "fid_accumulate: DP must be a multiple of 32 and >= D + 1");
    dim3 grid(DP / 32, DP / 32);
    if (is_f64) hipLaunchKernelGGL((fid_accumulate_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, feats, weights, S, B, D, DP);
    else hipLaunchKernelGGL((fid_accumulate_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, feats, weights, S, B, D, DP);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}
