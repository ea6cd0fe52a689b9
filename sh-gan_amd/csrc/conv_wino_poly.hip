// Polyphase Winograd forms of the stride-2 transposed 3x3 convolution (the "up" layers) on the fp32 matrix cores of
// gfx950 -- same 16-position / 16-wave skeleton as conv_wino.hip, different transforms.
//
// Reference semantics: conv2d_resample.py:122-137 -> F.conv_transpose2d(stride=2) as called from modulated_conv2d
// (stylegan.py:103-193) for every `conv0` of comodgan.synthesis_block (comodgan.py:304-340):
//      out[2u+ky, 2v+kx] += w[ky][kx] * x[u, v]          u in [0,H), v in [0,W), out is (2H+1) x (2W+1).
// The four sub-pixel phases (a, b) = (Y & 1, X & 1) of `out` are four INDEPENDENT small convolutions of the same
// low-resolution input (u, v index the phase plane; x = 0 outside the image):
//      ee[u,v] = sum_{ky,kx in {0,2}} w[ky][kx] x[u-ky/2, v-kx/2]        2x2 taps, (H+1) x (W+1)
//      eo[u,v] = sum_{ky in {0,2}}     w[ky][1]  x[u-ky/2, v]            2x1 taps, (H+1) x  W
//      oe[u,v] = sum_{kx in {0,2}}     w[1][kx]  x[u, v-kx/2]            1x2 taps,  H    x (W+1)
//      oo[u,v] =                       w[1][1]   x[u, v]                 1 tap,     H    x  W
// = 36 multiplies per 2x2 block of low-resolution pixels in the direct form.  Here:
//   scheme UA: `ee` as Winograd F(3x3, 2x2): 16 multiplies per 3x3 outputs (input transform B^T d B is the one of
//              F(2x2,3x3); G = [[1,0],[.5,.5],[.5,-.5],[0,1]], A^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,-1]]);
//   scheme UB: `eo`, `oe`, `oo` of a 2x2 block from its 3x3 input patch with 6 + 6 + 4 = 16 multiplies
//              (F(2,2) along the 2-tap axis: m1 = (d0-d1) g1, m2 = d1 (g0+g1), m3 = (d2-d1) g0; y0 = m1+m2, y1 = m2+m3);
//   the row u = H of `eo` and the column v = W of `oe` (one pixel thick) come from the strip tiles (poly_strip_body).
// Together (16/9 + 4) = 5.8 multiplies per low-resolution pixel instead of 9: 1.56x fewer MFMA flops, still exact fp32
// MFMA arithmetic with transform constants 0, +-1, +-1/2.
//
// Both schemes are "16 independent GEMMs M_xi[o,t] = sum_i U_xi[i,o] V_xi[i,t]" over 64 output channels x 64 blocks,
// so the kernel body is the one of conv_wino.hip: wave w owns position w (4 accumulator tiles of 32x32), U from
// registers (lane-major layout, one chunk ahead), raw input window by 16-byte LDS-DMA, V built in LDS by waves 0-7,
// double buffering with one barrier per 8-channel chunk, MFMA stream skewed one k-step across the barrier, exchange
// epilogue through LDS.  Output: raw phase planes [4][NB,O,H+1,W+1] for shg_upfir_planar_f32 (which applies the 4x4
// FIR of conv2d_resample.py:138, the demodulation coefficient and the layer tail).
//
// The stride-2 3x3 convolution of the "down" layers (conv2d_resample.py:116-120, after its FIR pre-filter) is the
// transpose of the above and gets the same treatment on the four polyphase planes P_ab[u,v] = xf[2u+a, 2v+b] of the
// filtered input (written in that layout by shg_fir_down_planar_f32):
//      y[p,q] = sum_{ky,kx in {0,2}} w[ky][kx] P_ee[p+ky/2, q+kx/2] + sum_{ky in {0,2}} w[ky][1] P_eo[p+ky/2, q]
//             + sum_{kx in {0,2}} w[1][kx] P_oe[p, q+kx/2] + w[1][1] P_oo[p,q]
//   scheme DA: the P_ee term as F(3x3,2x2) -> raw partial sums in y;
//   scheme DB: the other three terms of a 2x2 output block with 16 multiplies, added to the partial sums, then the
//              fused layer tail (bias, lrelu_agc, gain, skip).  Two launches (DB consumes DA's output).
#include "shg_common.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __attribute__((aligned(16))) float shg_poly_zeros[64];   // zero source for LDS-DMA lanes that fall into padding

struct PolySub {            // one scheme's share of a launch
    const float* wu;         // transformed weights [OP/64][nchunk][16][64 lanes][KC]
    int nby, nbx;            // blocks per image
    int tiles_x, tiles_y;    // tiles per image (flat tiling: tiles_x runs of 64 consecutive blocks, tiles_y = 1)
    int n_ttiles;            // tiles_x * tiles_y * NB
};
struct PolyParams {
    const float* x;          // [NB, I, H, W]
    float* y;                // phase planes [4][NB, O, H+1, W+1]
    const float* in_scale;   // [NB, I] or null
    int NB, I, O, OP, H, W;
    int n_otiles, nchunk;
    PolySub a, b;            // scheme UA / UB; workgroups [0, a.n_ttiles*n_otiles) run UA, the next b.n_ttiles*n_otiles UB
    const float* wt;         // GEMM-layout weights [OP/64][IPK][64] (strips); workgroups past UA + UB run four strip tiles each
    int IPK, n_strip_tiles;
    // down schemes: x = polyphase planes [4][NB, I, PH2, PP] (PP % 4 == 0), y = [NB, O, H, W] with H, W the OUTPUT extent
    int PH2, PP;
    const float* bias;       // [O] or null
    const float* residual;   // like y, added after the activation, or null
    int act;
    float alpha, gain, clamp;
    // up launches on small grids: split along the input channels (conv_wino.hip, K-split): chunks per slice, slice = blockIdx.y,
    // floats between the slices' partial planes (0 and cps = nchunk: not split; the down schemes never split)
    int cps;
    long part_stride;
};

namespace poly {
constexpr int KC = 8, BO = 64, BT = 64;
constexpr int V_SZ = 16 * KC * BT;
constexpr int NT = 1024;
constexpr int NXF = KC;

// Scheme UA: 3x3 output blocks of the `ee` plane from 4x4 input patches.  Scheme UB: 2x2 blocks of eo / oe / oo from 3x3 patches.
// DA / DB: the same for the stride-2 convolution, on the polyphase planes of the filtered input.
enum { UA = 0, UB = 1, DA = 2, DB = 3 };

// Tile = 64 blocks.  Rectangular (NBX = 0): TY x TX blocks.  Flat (scheme UA, NBX = blocks per image row, a compile-time
// constant): 64 CONSECUTIVE blocks of the row-major block list of one image -- a grid of 22 x 22 blocks (65 x 65 phase
// plane) then costs 8 tiles per image instead of the 9 of an 8 x 8 tiling, and its window spans the image width.
template <int SCHEME, int TY, int TX, int NBX>
struct Geo {
    static constexpr bool A = SCHEME == UA || SCHEME == DA;             // 3x3 blocks from 4x4 patches
    static_assert(NBX > 0 || TY * TX == BT, "64 blocks per tile");
    static_assert(NBX == 0 || A, "flat tiling is for the 3x3 blocks of schemes UA / DA");
    static constexpr int BS = A ? 3 : 2;                               // block edge in output pixels
    static constexpr int RMAX = NBX ? (NBX - 1 + BT + NBX - 1) / NBX : TY;                 // block rows a tile can touch
    static constexpr int PH = A ? 3 * RMAX + 1 : 2 * TY + 1;                               // window rows (UB; DB: see below)
    // window columns: 16-byte aligned start at or before the first needed column, rounded up to whole float4
    static constexpr int PW = SCHEME == UA ? ((3 * (NBX ? NBX : TX) + 1 + 3 + 3) / 4) * 4
                            : SCHEME == DA ? (NBX ? ((3 * NBX + 1 + 3) / 4) * 4 : ((3 * TX + 1 + 3 + 3) / 4) * 4)
                            : 2 * TX + 4;
    // DB: three windows per channel, one per plane: eo (2TY+1) x 2TX, oe 2TY x (2TX+4), oo 2TY x 2TX
    static constexpr int DB_W0 = 2 * TX, DB_W1 = 2 * TX + 4, DB_W2 = 2 * TX;
    static constexpr int DB_N0 = (2 * TY + 1) * DB_W0, DB_N1 = 2 * TY * DB_W1, DB_N2 = 2 * TY * DB_W2;
    static constexpr int RP = SCHEME == DB ? DB_N0 + DB_N1 + DB_N2 : PH * PW;
    static constexpr int PW4 = PW / 4, PATCH4 = RP / 4, R_SZ = KC * RP;
    static constexpr int NPIECE = (PATCH4 + 63) / 64;
};
}   // namespace poly

__device__ __forceinline__ int poly_xcd_remap(int bid, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <int SCHEME, int TY, int TX, int NBX>
__device__ __forceinline__ void poly_body(const PolyParams& p, const PolySub& q, const int bid, float* __restrict__ Vl,
                                          float* __restrict__ Rl) {
    using namespace poly;
    using G = Geo<SCHEME, TY, TX, NBX>;
    constexpr int PW = G::PW, PW4 = G::PW4, PATCH4 = G::PATCH4, RP = G::RP, R_SZ = G::R_SZ, BS = G::BS, NPIECE = G::NPIECE;
    constexpr bool FLAT = NBX > 0;
    constexpr bool DOWN = SCHEME == DA || SCHEME == DB;
    const int c0 = DOWN ? 0 : blockIdx.y * p.cps, nch = DOWN ? p.nchunk : min(p.cps, p.nchunk - c0), iend = min(p.I, (c0 + nch) * poly::KC);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;

    const int nwork = q.n_ttiles * p.n_otiles;
    const int work = poly_xcd_remap(bid, nwork);
    const int otile = work / q.n_ttiles;
    const int ttile = work - otile * q.n_ttiles;
    const int txb = ttile % q.tiles_x;
    const int tyb = (ttile / q.tiles_x) % q.tiles_y;
    const int n = ttile / (q.tiles_x * q.tiles_y);
    const int id0 = txb * BT;                            // flat: first block id of the tile
    const int by0 = FLAT ? id0 / (NBX ? NBX : 1) : tyb * TY, bx0 = FLAT ? 0 : txb * TX;      // first block row / column of the window
    const int o0 = otile * BO;
    const int HW = DOWN ? p.PH2 * p.PP : p.H * p.W;      // input channel stride
    // window origin: the up schemes read rows / columns BS*b - 1 ... of the low-resolution image, the down schemes rows /
    // columns BS*b ... of their polyphase planes; the column start is rounded down to a multiple of 4 floats
    // (UB: 2*bx0 - 4, i.e. three columns of slack; UA / DA: 0..3 columns; DB: 2*bx0 is a multiple of 4)
    const int wy0 = DOWN ? BS * by0 : BS * by0 - 1;
    const int wx0 = SCHEME == UA ? ((3 * bx0 - 1) & ~3) : SCHEME == DA ? ((3 * bx0) & ~3) : SCHEME == UB ? 2 * bx0 - 4 : 2 * bx0;
    const int coff = (DOWN ? BS * bx0 : BS * bx0 - 1) - wx0;           // window column of the first needed input column

    const bool xformer = wave < NXF;
    constexpr int NLD = 16 - NXF;
    constexpr int CPL = (KC + NLD - 1) / NLD;
    int roff[NPIECE];
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
        const int e = j * 64 + lane;                     // float4 index inside the channel's LDS image
        if constexpr (SCHEME == DB) {
            // three windows: plane 1 (eo), 2 (oe), 3 (oo); all start at row 2*by0, column 2*bx0 of their plane
            constexpr int N0 = G::DB_N0 / 4, N1 = G::DB_N1 / 4, W0 = G::DB_W0 / 4, W1 = G::DB_W1 / 4, W2 = G::DB_W2 / 4;
            const int pl = e < N0 ? 1 : (e < N0 + N1 ? 2 : 3);
            const int r4 = e < N0 ? e : (e < N0 + N1 ? e - N0 : e - N0 - N1);
            const int w4 = pl == 1 ? W0 : (pl == 2 ? W1 : W2);
            const int py = r4 / w4, p4 = r4 - py * w4;
            const int iy = wy0 + py, ix = wx0 + 4 * p4;
            const bool ok = e < PATCH4 && iy < p.PH2 && ix + 3 < p.PP;
            roff[j] = ok ? (pl * p.NB + n) * p.I * HW + iy * p.PP + ix : -1;
        } else {
            const int py = e / PW4, p4 = e - py * PW4;
            const int iy = wy0 + py, ix = wx0 + 4 * p4;
            if constexpr (SCHEME == DA) {
                const bool ok = e < PATCH4 && iy < p.PH2 && ix + 3 < p.PP;
                roff[j] = ok ? n * p.I * HW + iy * p.PP + ix : -1;
            } else {
                const bool ok = e < PATCH4 && iy >= 0 && iy < p.H && ix >= 0 && ix + 3 < p.W;
                roff[j] = ok ? n * p.I * HW + iy * p.W + ix : -1;
            }
        }
    }
    const bool ract_last = lane < PATCH4 - 64 * (NPIECE - 1);
    auto dma_raw = [&](int c, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int k = (wave - NXF) * CPL + q;
            if (k >= KC) continue;
            const int ch = (c0 + c) * KC + k;
            const bool chok = ch < iend;
#pragma unroll
            for (int j = 0; j < NPIECE; ++j) {
                const float* src = (chok && roff[j] >= 0) ? p.x + ((long)roff[j] + (long)ch * HW) : shg_poly_zeros;
                if (j < NPIECE - 1 || ract_last)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(Rl + buf * R_SZ + k * RP + j * 256), 16, 0, 0);
            }
        }
    };

    constexpr int NU = KC / 4;
    const size_t ustride = (size_t)16 * 64 * KC / 4;
    const f32x4* ubase = reinterpret_cast<const f32x4*>(q.wu + (((size_t)otile * p.nchunk * 16 + wave) * 64 + lane) * KC) + (size_t)c0 * ustride;
    f32x4 ua[NU], ub[NU];
    auto load_u = [&](f32x4 (&dst)[NU], int c) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NU; ++j) dst[j] = ubase[(size_t)c * ustride + j];
    };

    // ---- input transform role: channel `wave`, block `lane`
    const int tty = FLAT ? (id0 + lane) / (NBX ? NBX : 1) - by0 : lane / TX;
    const int ttx = FLAT ? (id0 + lane) % (NBX ? NBX : 1) : lane % TX;
    const float* rbase = Rl + wave * RP + (SCHEME == DB ? 0 : (BS * tty) * PW + BS * ttx + coff);
    float* vbase = Vl + wave * BT + lane;
    float scv[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int ch = (v * 64 + lane) * KC + wave;
        scv[v] = (p.in_scale && xformer && ch < p.I) ? p.in_scale[(long)n * p.I + ch] : 1.f;
    }
    auto transform = [&](int c, int buf) __attribute__((always_inline)) {
        const int ca = c0 + c;
        const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ca < 64 ? scv[0] : scv[1]), ca & 63));
        const float* rb = rbase + buf * R_SZ;
        float* vb = vbase + buf * V_SZ;
        if constexpr (SCHEME == DB) {
            // eo window rows 2ty .. 2ty+2, columns 2tx, 2tx+1: F(2,2) down the rows; positions j*2 + c
            const float* we = rb + (2 * tty) * G::DB_W0 + 2 * ttx;
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const float d0 = we[cc] * sc, d1 = we[G::DB_W0 + cc] * sc, d2 = we[2 * G::DB_W0 + cc] * sc;
                vb[(0 * 2 + cc) * KC * BT] = d0 - d1;
                vb[(1 * 2 + cc) * KC * BT] = d1;
                vb[(2 * 2 + cc) * KC * BT] = d2 - d1;
            }
            // oe window rows 2ty, 2ty+1, columns 2tx .. 2tx+2: F(2,2) along the columns; positions 6 + r*3 + j
            const float* wo = rb + G::DB_N0 + (2 * tty) * G::DB_W1 + 2 * ttx;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float e0 = wo[r * G::DB_W1] * sc, e1 = wo[r * G::DB_W1 + 1] * sc, e2 = wo[r * G::DB_W1 + 2] * sc;
                vb[(6 + r * 3 + 0) * KC * BT] = e0 - e1;
                vb[(6 + r * 3 + 1) * KC * BT] = e1;
                vb[(6 + r * 3 + 2) * KC * BT] = e2 - e1;
            }
            // oo window: positions 12 + r*2 + c
            const float* wq = rb + G::DB_N0 + G::DB_N1 + (2 * tty) * G::DB_W2 + 2 * ttx;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) vb[(12 + r * 2 + cc) * KC * BT] = wq[r * G::DB_W2 + cc] * sc;
        } else if constexpr (G::A) {
            // B^T d B on the 4x4 patch (rows first, two rows of LDS reads in flight at a time)
            float f[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d0 = rb[r * PW] * sc, d1 = rb[r * PW + 1] * sc, d2 = rb[r * PW + 2] * sc, d3 = rb[r * PW + 3] * sc;
                f[r][0] = d0 - d2; f[r][1] = d1 + d2; f[r][2] = d2 - d1; f[r][3] = d1 - d3;
                if (r == 1) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                vb[(0 * 4 + j) * KC * BT] = f[0][j] - f[2][j];
                vb[(1 * 4 + j) * KC * BT] = f[1][j] + f[2][j];
                vb[(2 * 4 + j) * KC * BT] = f[2][j] - f[1][j];
                vb[(3 * 4 + j) * KC * BT] = f[1][j] - f[3][j];
            }
        } else {
            float d[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) d[r][cc] = rb[r * PW + cc] * sc;
            // eo: positions j*2 + (c-1), F(2,2) down the rows of columns 1, 2
#pragma unroll
            for (int cc = 1; cc < 3; ++cc) {
                vb[(0 * 2 + cc - 1) * KC * BT] = d[0][cc] - d[1][cc];
                vb[(1 * 2 + cc - 1) * KC * BT] = d[1][cc];
                vb[(2 * 2 + cc - 1) * KC * BT] = d[2][cc] - d[1][cc];
            }
            // oe: positions 6 + (r-1)*3 + j, F(2,2) along the columns of rows 1, 2
#pragma unroll
            for (int r = 1; r < 3; ++r) {
                vb[(6 + (r - 1) * 3 + 0) * KC * BT] = d[r][0] - d[r][1];
                vb[(6 + (r - 1) * 3 + 1) * KC * BT] = d[r][1];
                vb[(6 + (r - 1) * 3 + 2) * KC * BT] = d[r][2] - d[r][1];
            }
            // oo: positions 12 + (r-1)*2 + (c-1)
#pragma unroll
            for (int r = 1; r < 3; ++r)
#pragma unroll
                for (int cc = 1; cc < 3; ++cc) vb[(12 + (r - 1) * 2 + cc - 1) * KC * BT] = d[r][cc];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ob][tb][r] = 0.f;
    const float* bbase = Vl + (wave * KC + half) * BT + l31;

    load_u(ua, 0);
    if (!xformer) {
        dma_raw(0, 0);
        if (nch > 1) dma_raw(1, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (xformer) transform(0, 0);
    __syncthreads();

    float b[2][2], apend[2];
    auto mma = [&](float a0, float a1, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) acc[0][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[buf][tb], acc[0][tb], 0, 0, 0);
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) acc[1][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[buf][tb], acc[1][tb], 0, 0, 0);
    };
    auto chunk = [&](auto par, int c) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par)::value;
        f32x4 (&ucur)[NU] = PAR ? ub : ua;
        f32x4 (&unxt)[NU] = PAR ? ua : ub;
        const bool more = c + 1 < nch;
        const float* bb = bbase + PAR * V_SZ;
        auto fetch = [&](int ks, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) b[buf][tb] = bb[(ks * 2) * BT + tb * 32];
        };
        fetch(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (c > 0) mma(apend[0], apend[1], 1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) load_u(unxt, c + 1);
        if (!xformer) {
            if (c + 2 < nch) dma_raw(c + 2, PAR);
        } else if (more) {
            transform(c + 1, PAR ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KC / 2 - 1; ++ks) {
            fetch(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(ucur[(ks * 2) / 4][(ks * 2) % 4], ucur[(ks * 2 + 1) / 4][(ks * 2 + 1) % 4], ks & 1);
        }
        apend[0] = ucur[(KC - 2) / 4][(KC - 2) % 4];
        apend[1] = ucur[(KC - 1) / 4][(KC - 1) % 4];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
    };
    for (int c = 0; c < nch; c += 2) {
        chunk(std::integral_constant<int, 0>{}, c);
        if (c + 1 < nch) chunk(std::integral_constant<int, 1>{}, c + 1);
    }
    mma(apend[0], apend[1], 1);

    // ---- epilogue: exchange the 16 M_xi of every (channel, block) through LDS, inverse transform, raw plane stores
    float* Mx = Vl;                           // [16][32][32]
    const int PWg = DOWN ? p.W : p.W + 1;     // output row pitch
    const long plane = DOWN ? (long)p.H * p.W : (long)(p.H + 1) * PWg;
    const int o_l = tid >> 5, t_l = tid & 31;
    // scheme DB: the partial sums of scheme DA, the skip tensor and the bias of all four passes are requested up front
    // (one memory latency instead of four)
    f32x2 pre_part[SCHEME == DB ? 4 : 1][2], pre_res[SCHEME == DB ? 4 : 1][2];
    float pre_bias[2] = {0.f, 0.f};
    if constexpr (SCHEME == DB) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int ob = pass >> 1, t = (pass & 1) * 32 + t_l;
            const int by = by0 + t / TX, bx = bx0 + t % TX;
            const int o = min(o0 + ob * 32 + o_l, p.O - 1);
            if ((pass & 1) == 0) pre_bias[ob] = p.bias ? p.bias[o] : 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                pre_part[pass][j] = f32x2{0.f, 0.f};
                pre_res[pass][j] = f32x2{0.f, 0.f};
                if (by < q.nby && bx < q.nbx && 2 * by + j < p.H) {
                    const long off = ((long)n * p.O + o) * plane + (long)(2 * by + j) * PWg + 2 * bx;
                    pre_part[pass][j] = *reinterpret_cast<const f32x2*>(p.y + off);
                    if (p.residual) pre_res[pass][j] = *reinterpret_cast<const f32x2*>(p.residual + off);
                }
            }
        }
    }
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int ob = pass >> 1, tb = pass & 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            Mx[(wave * 32 + row) * 32 + l31] = acc[ob][tb][r];
        }
        __syncthreads();
        float m[16];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) m[xi] = Mx[(xi * 32 + o_l) * 32 + t_l];
        const int t = tb * 32 + t_l;
        const int by = FLAT ? (id0 + t) / (NBX ? NBX : 1) : by0 + t / TX, bx = FLAT ? (id0 + t) % (NBX ? NBX : 1) : bx0 + t % TX;
        const int o = o0 + ob * 32 + o_l;
        if (o < p.O && by < q.nby && bx < q.nbx) {
            float* yb = p.y + (DOWN ? 0 : blockIdx.y * p.part_stride) + ((long)n * p.O + o) * plane;             // + phase * NB*O*plane
            const long pstride = (long)p.NB * p.O * plane;
            if constexpr (G::A) {
                // A^T m A with A^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,-1]]
                float t3[3][4];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    t3[0][cc] = m[0 * 4 + cc] + m[1 * 4 + cc] + m[2 * 4 + cc];
                    t3[1][cc] = m[1 * 4 + cc] - m[2 * 4 + cc];
                    t3[2][cc] = m[1 * 4 + cc] + m[2 * 4 + cc] - m[3 * 4 + cc];
                }
                constexpr int EXT = DOWN ? 0 : 1;                 // the ee plane of the up path is (H+1) x (W+1)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int u = 3 * by + i;
                    if (u >= p.H + EXT) continue;
                    const float y0 = t3[i][0] + t3[i][1] + t3[i][2], y1 = t3[i][1] - t3[i][2], y2 = t3[i][1] + t3[i][2] - t3[i][3];
                    float* yr = yb + (long)u * PWg + 3 * bx;
                    yr[0] = y0;
                    if (3 * bx + 1 < p.W + EXT) yr[1] = y1;
                    if (3 * bx + 2 < p.W + EXT) yr[2] = y2;
                }
            } else if constexpr (SCHEME == DB) {
                // y[p0+j][q0+c] = partial (scheme DA) + eo + oe + oo terms, then bias / activation / skip (W even: both
                // columns of a block are inside the image and 8-byte aligned)
                const int p0 = 2 * by, q0 = 2 * bx;
                const float bs = pre_bias[ob];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (p0 + j >= p.H) continue;
                    float* yr = yb + (long)(p0 + j) * PWg + q0;
                    const f32x2 part = pre_part[pass][j];
                    const f32x2 rs = pre_res[pass][j];
                    f32x2 out;
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        float v = part[cc] + m[j * 2 + cc] + m[(j + 1) * 2 + cc]                       // eo: m_j + m_{j+1}
                                + m[6 + j * 3 + cc] + m[6 + j * 3 + cc + 1]                          // oe
                                + m[12 + j * 2 + cc] + bs;
                        v = p.act ? shg_lrelu_agc(v, p.alpha, p.gain, p.clamp) : v * p.gain;
                        out[cc] = v + rs[cc];
                    }
                    *reinterpret_cast<f32x2*>(yr) = out;
                }
            } else {
                const int u0 = 2 * by, v0 = 2 * bx;                    // all four pixels of a body block are inside the image
                float* yeo = yb + pstride + (long)u0 * PWg + v0;
                float* yoe = yb + 2 * pstride + (long)u0 * PWg + v0;
                float* yoo = yb + 3 * pstride + (long)u0 * PWg + v0;
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    yeo[cc] = m[0 * 2 + cc] + m[1 * 2 + cc];
                    yeo[PWg + cc] = m[1 * 2 + cc] + m[2 * 2 + cc];
                }
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    yoe[r * PWg] = m[6 + r * 3] + m[6 + r * 3 + 1];
                    yoe[r * PWg + 1] = m[6 + r * 3 + 1] + m[6 + r * 3 + 2];
                    yoo[r * PWg] = m[12 + r * 2];
                    yoo[r * PWg + 1] = m[12 + r * 2 + 1];
                }
            }
        }
        if (pass < 3) __syncthreads();
    }
}

// One launch runs everything: workgroups [0, nA) take UA tiles, the next nB UB tiles, the last few four strip tiles each
// -- a single tail round instead of three.  LDS: two separate objects (the compiler then knows that the LDS-DMA into the
// raw windows cannot alias the operand reads from V, and parks no `s_waitcnt vmcnt(0)` in front of a chunk's MFMAs),
// sized for the larger scheme.
// The one-pixel strips the 2x2 body blocks of scheme UB do not reach: eo[H, v] = w[2][1] . x[H-1, v] (v < W) and
// oe[u, W] = w[1][2] . x[u, W-1] (u < H): two single-tap contractions over the input channels along one image row /
// column -- a tiny, latency-bound problem.  One strip tile (4 waves) = 32 output channels x 32 strip positions of one
// image; the four waves split the input channels and add their partial tiles through LDS, every wave keeps 32 operand
// loads in flight.  Strip tiles ride in the tail of the main launch, four per 16-wave workgroup.  Operands come straight from global memory (wt is the GEMM layout of shg_conv_weight_prep_f32:
// [OP/64][IP*9][64], row = i*9 + tap).
__device__ __forceinline__ void poly_strip_body(const PolyParams& p, const int tile, const int wave, float* __restrict__ redbuf) {
    float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(redbuf);      // [4][16][64] of this strip tile
    const float* wt = p.wt;
    const int IPK = p.IPK;
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const int nseg_r = (p.W + 31) / 32, nseg_c = (p.H + 31) / 32;
    const int no32 = p.OP / 32;
    const bool live = tile < p.n_strip_tiles;                 // (a dead tile still takes part in the barrier)
    int seg = tile % (nseg_r + nseg_c);
    const int ot32 = (tile / (nseg_r + nseg_c)) % no32;
    const int n = live ? tile / ((nseg_r + nseg_c) * no32) : 0;
    const bool col = seg >= nseg_r;
    if (col) seg -= nseg_r;
    const int len = col ? p.H : p.W;
    const int pos = seg * 32 + l31;                           // strip position of this lane's B column
    const int tap = col ? 1 * 3 + 2 : 2 * 3 + 1;
    const int HW = p.H * p.W;
    const bool pok = live && pos < len;
    const float* xb = p.x + (long)n * p.I * HW + (pok ? (col ? pos * p.W + p.W - 1 : (p.H - 1) * p.W + pos) : 0);
    const float* wb = wt + ((long)(ot32 >> 1) * IPK + tap) * 64 + (ot32 & 1) * 32 + l31;     // + i*9*64
    const float* sb = p.in_scale ? p.in_scale + (long)n * p.I : nullptr;
    const int i_lo = blockIdx.y * p.cps * poly::KC, i_hi = min(p.I, i_lo + p.cps * poly::KC);      // this slice's input channels (all of them when not split)
    const int kper = ((i_hi - i_lo + 7) / 8) * 2;             // channels per wave (even)
    const int kbeg = i_lo + wave * kper, kend = min(i_hi, kbeg + kper);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int i0 = kbeg; i0 < kend; i0 += 32) {
        float a[16], bv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int ic = i0 + 2 * k + half;
            const int i = min(ic, p.I - 1);
            a[k] = wb[(long)i * 9 * 64];
            bv[k] = xb[(long)i * HW];
            if (sb) bv[k] *= sb[i];
            if (ic >= kend) a[k] = 0.f;                   // (a lane's A operand belongs to output channel l31, its B operand to
            if (ic >= kend || !pok) bv[k] = 0.f;          //  strip position l31: only B is masked by the position)
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], bv[k], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    // thread (wave, lane) finishes registers 4*wave .. 4*wave+3 of lane `lane`
    if (!pok) return;
    const int PWg = p.W + 1;
    const long plane = (long)(p.H + 1) * PWg;
    const long pix = col ? (long)pos * PWg + p.W : (long)p.H * PWg + pos;
    float* yb = p.y + blockIdx.y * p.part_stride + (long)(col ? 2 : 1) * p.NB * p.O * plane + (long)n * p.O * plane + pix;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = wave * 4 + rr;
        const float v = red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane];
        const int o = ot32 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (o < p.O) yb[(long)o * plane] = v;
    }
}

template <int ATY, int ATX, int ANBX, int BTY, int BTX>
__global__ __launch_bounds__(1024) void conv_poly_up_kernel(const PolyParams p) {
    using namespace poly;
    using GA = Geo<UA, ATY, ATX, ANBX>;
    using GB = Geo<UB, BTY, BTX, 0>;
    constexpr int R_MAX = GA::R_SZ > GB::R_SZ ? GA::R_SZ : GB::R_SZ;
    __shared__ __attribute__((aligned(16))) float Vl[2 * V_SZ];      // [2][16][KC][64]
    __shared__ __attribute__((aligned(16))) float Rl[2 * R_MAX];     // [2][KC][RP]
    const int nA = p.a.n_ttiles * p.n_otiles, nB = p.b.n_ttiles * p.n_otiles;
    if ((int)blockIdx.x < nA) poly_body<UA, ATY, ATX, ANBX>(p, p.a, blockIdx.x, Vl, Rl);
    else if ((int)blockIdx.x < nA + nB) poly_body<UB, BTY, BTX, 0>(p, p.b, blockIdx.x - nA, Vl, Rl);
    else {
        const int wave = threadIdx.x >> 6;
        poly_strip_body(p, (blockIdx.x - nA - nB) * 4 + (wave >> 2), wave & 3, Vl + (wave >> 2) * 4 * 16 * 64);
    }
}

// Stride-2 convolution: one scheme per launch (DB consumes the partial sums DA wrote).
template <int SCHEME, int TY, int TX, int NBX>
__global__ __launch_bounds__(1024) void conv_poly_down_kernel(const PolyParams p) {
    using namespace poly;
    using G = Geo<SCHEME, TY, TX, NBX>;
    __shared__ __attribute__((aligned(16))) float Vl[2 * V_SZ];
    __shared__ __attribute__((aligned(16))) float Rl[2 * G::R_SZ];
    poly_body<SCHEME, TY, TX, NBX>(p, SCHEME == DA ? p.a : p.b, blockIdx.x, Vl, Rl);
}

// U for both schemes from w [O,I,3,3] * scale[o]; tap (ky,kx) = w[ky*3+kx] (flip: 8 - index), as the transposed kernel of
// conv_mfma.hip indexes them.  Layout as conv_wino.hip: wu[otile][chunk][xi][lane][KC].
__global__ __launch_bounds__(256) void poly_weight_kernel(const float* w, const float* scale, float* wu, int O, int I, int OP,
                                                          int nchunk, int flip, int scheme) {
    constexpr int KC = poly::KC;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)OP * nchunk * KC;
    if (e >= total) return;
    const int o = (int)(e % OP);
    const int i = (int)(e / OP);
    float g[3][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int tt = flip ? 8 - t : t;
        g[t / 3][t % 3] = (o < O && i < I) ? w[((long)o * I + i) * 9 + tt] * scale[o] : 0.f;
    }
    float u[16];
    if (scheme == poly::UA) {
        // correlation taps of the 2x2 problem: q[k][l] = w[2-2k][2-2l]; U = G q G^T, G = [[1,0],[.5,.5],[.5,-.5],[0,1]]
        const float q00 = g[2][2], q01 = g[2][0], q10 = g[0][2], q11 = g[0][0];
        float gq[4][2];
        gq[0][0] = q00; gq[0][1] = q01;
        gq[1][0] = 0.5f * (q00 + q10); gq[1][1] = 0.5f * (q01 + q11);
        gq[2][0] = 0.5f * (q00 - q10); gq[2][1] = 0.5f * (q01 - q11);
        gq[3][0] = q10; gq[3][1] = q11;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            u[r * 4 + 0] = gq[r][0];
            u[r * 4 + 1] = 0.5f * (gq[r][0] + gq[r][1]);
            u[r * 4 + 2] = 0.5f * (gq[r][0] - gq[r][1]);
            u[r * 4 + 3] = gq[r][1];
        }
    } else if (scheme == poly::UB) {
        const float g0 = g[0][1], g1 = g[2][1];        // eo: tap on x[u], tap on x[u-1]
        const float h0 = g[1][0], h1 = g[1][2];        // oe: tap on x[v], tap on x[v-1]
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) { u[0 * 2 + cc] = g1; u[1 * 2 + cc] = g0 + g1; u[2 * 2 + cc] = g0; }
#pragma unroll
        for (int r = 0; r < 2; ++r) { u[6 + r * 3 + 0] = h1; u[6 + r * 3 + 1] = h0 + h1; u[6 + r * 3 + 2] = h0; }
#pragma unroll
        for (int k = 0; k < 4; ++k) u[12 + k] = g[1][1];
    } else if (scheme == poly::DA) {
        // correlation taps on the ee plane: q[k][l] = w[2k][2l]
        const float q00 = g[0][0], q01 = g[0][2], q10 = g[2][0], q11 = g[2][2];
        float gq[4][2];
        gq[0][0] = q00; gq[0][1] = q01;
        gq[1][0] = 0.5f * (q00 + q10); gq[1][1] = 0.5f * (q01 + q11);
        gq[2][0] = 0.5f * (q00 - q10); gq[2][1] = 0.5f * (q01 - q11);
        gq[3][0] = q10; gq[3][1] = q11;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            u[r * 4 + 0] = gq[r][0];
            u[r * 4 + 1] = 0.5f * (gq[r][0] + gq[r][1]);
            u[r * 4 + 2] = 0.5f * (gq[r][0] - gq[r][1]);
            u[r * 4 + 3] = gq[r][1];
        }
    } else {
        // DB, correlation F(2,2): m1 = (d0-d1) g0, m2 = d1 (g0+g1), m3 = (d2-d1) g1
        const float g0 = g[0][1], g1 = g[2][1];        // eo plane: taps on P_eo[p], P_eo[p+1]
        const float h0 = g[1][0], h1 = g[1][2];        // oe plane: taps on P_oe[q], P_oe[q+1]
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) { u[0 * 2 + cc] = g0; u[1 * 2 + cc] = g0 + g1; u[2 * 2 + cc] = g1; }
#pragma unroll
        for (int r = 0; r < 2; ++r) { u[6 + r * 3 + 0] = h0; u[6 + r * 3 + 1] = h0 + h1; u[6 + r * 3 + 2] = h1; }
#pragma unroll
        for (int k = 0; k < 4; ++k) u[12 + k] = g[1][1];
    }
    const int k = i % KC, chunk = i / KC;
    const int ln = (k & 1) * 32 + (o & 31), slot = (k >> 1) * 2 + ((o & 63) >> 5);
    float* dst = wu + ((((long)(o >> 6) * nchunk + chunk) * 16) * 64 + ln) * KC + slot;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) dst[(long)xi * 64 * KC] = u[xi];
}

// w [O,I,3,3], wscale [O] -> wu_a, wu_b (each [OP/64][ceil(I/8)][16][64][8] floats).
extern "C" int shg_conv_weight_prep_up_poly_f32(const float* w, const float* wscale, float* wu_a, float* wu_b, int O, int I, int OP,
                                                int flip, void* stream) {
    SHG_CHECK_ARG(w && wscale && wu_a && wu_b, "weight_prep_up_poly: null pointer");
    SHG_CHECK_ARG(O >= 1 && I >= 1 && OP % 64 == 0 && OP >= O, "weight_prep_up_poly: bad shape");
    const int nchunk = shg_cdiv(I, poly::KC);
    const long total = (long)OP * nchunk * poly::KC;
    hipLaunchKernelGGL(poly_weight_kernel, dim3(shg_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wscale, wu_a, O, I, OP,
                       nchunk, flip, (int)poly::UA);
    SHG_CHECK_LAUNCH();
    hipLaunchKernelGGL(poly_weight_kernel, dim3(shg_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wscale, wu_b, O, I, OP,
                       nchunk, flip, (int)poly::UB);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// 1 when shg_conv2d_up_poly_f32 handles this geometry (otherwise use shg_conv2d_f32 mode 2, out_mode 1)
extern "C" int shg_conv2d_up_poly_supported(int NB, int I, int O, int H, int W) {
    return (H >= 16 && W >= 16 && H % 2 == 0 && W % 4 == 0 && I <= 128 * poly::KC && NB >= 1 && O >= 1) ? 1 : 0;
}

// Phase planes of the stride-2 transposed 3x3 convolution of x * in_scale[n,i]: y [4][NB,O,H+1,W+1] (same contract as
// shg_conv2d_f32 mode 2 / out_mode 1 without epilogue operands).  wt = GEMM-layout weights of shg_conv_weight_prep_f32
// (for the strips), wu_a / wu_b from shg_conv_weight_prep_up_poly_f32.
int shg_wino_ksplit(long tiles, int nchunk);                   // conv_wino.hip
void shg_launch_wino_split_reduce(const float* part, float* y, int ks, int NB, int O, int H, int W, const float* out_scale, const float* bias,
                                  const float* noise, int noise_mode, float noise_strength, int act, float alpha, float gain, float clamp,
                                  const float* residual, hipStream_t s);

// tiling of an up launch; returns the kernel variant: 0 narrow, 1 / 2 / 3 flat with 11 / 22 / 43 block columns, 4 wide, 5 rectangular 8 x 8
static int up_poly_plan(PolyParams& p, int NB, int I, int OP, int H, int W) {
    p.n_otiles = OP / 64; p.nchunk = shg_cdiv(I, poly::KC);
    // scheme UB: 2x2 body blocks, 4 x 16 blocks per tile (8 x 8 for images narrower than 32)
    const bool bwide = W >= 32;
    p.b.nby = H / 2; p.b.nbx = W / 2;
    p.b.tiles_x = shg_cdiv(p.b.nbx, bwide ? 16 : 8); p.b.tiles_y = shg_cdiv(p.b.nby, bwide ? 4 : 8);
    p.b.n_ttiles = p.b.tiles_x * p.b.tiles_y * NB;
    // scheme UA: 3x3 blocks of the ee plane; flat tiling for the block-row lengths of the generator's layers, else the
    // rectangular tile shape with the smaller padding waste
    p.a.nby = shg_cdiv(H + 1, 3); p.a.nbx = shg_cdiv(W + 1, 3);
    const bool flat = bwide && (p.a.nbx == 11 || p.a.nbx == 22 || p.a.nbx == 43);
    const long w88 = (long)shg_cdiv(p.a.nbx, 8) * shg_cdiv(p.a.nby, 8), w416 = (long)shg_cdiv(p.a.nbx, 16) * shg_cdiv(p.a.nby, 4);
    const bool wide = bwide && w416 < w88;
    if (flat) { p.a.tiles_x = shg_cdiv(p.a.nby * p.a.nbx, 64); p.a.tiles_y = 1; }
    else { p.a.tiles_x = shg_cdiv(p.a.nbx, wide ? 16 : 8); p.a.tiles_y = shg_cdiv(p.a.nby, wide ? 4 : 8); }
    p.a.n_ttiles = p.a.tiles_x * p.a.tiles_y * NB;
    // strips: tiles of 32 channels x 32 positions, four per workgroup
    p.IPK = (I + 31) / 32 * 32 * 9;
    p.n_strip_tiles = (shg_cdiv(W, 32) + shg_cdiv(H, 32)) * (OP / 32) * NB;
    if (!bwide) return 0;
    if (flat) return p.a.nbx == 11 ? 1 : (p.a.nbx == 22 ? 2 : 3);
    return wide ? 4 : 5;
}

// bytes of scratch with which shg_conv2d_up_poly_ws_f32 splits this problem along its input channels (0: it will not; conv_wino.hip, K-split)
extern "C" size_t shg_conv2d_up_poly_workspace_bytes(int NB, int I, int O, int OP, int H, int W) {
    if (!shg_conv2d_up_poly_supported(NB, I, O, H, W) || OP < 64) return 0;
    PolyParams p{};
    up_poly_plan(p, NB, I, OP, H, W);
    const int ks = shg_wino_ksplit((long)(p.a.n_ttiles + p.b.n_ttiles) * p.n_otiles, p.nchunk);
    return ks > 1 ? (size_t)ks * 4 * NB * O * (H + 1) * (W + 1) * sizeof(float) : 0;
}

extern "C" int shg_conv2d_up_poly_ws_f32(const float* x, const float* wt, const float* wu_a, const float* wu_b, float* y, int NB, int I,
                                         int O, int OP, int H, int W, const float* in_scale, void* workspace, size_t ws_bytes, void* stream) {
    SHG_CHECK_ARG(x && wt && wu_a && wu_b && y, "conv2d_up_poly: null pointer");
    SHG_CHECK_ARG(shg_conv2d_up_poly_supported(NB, I, O, H, W), "conv2d_up_poly: unsupported geometry (use shg_conv2d_f32 mode 2)");
    SHG_CHECK_ARG(OP % 64 == 0 && OP >= O, "conv2d_up_poly: OP must be a multiple of 64 and >= O");
    SHG_CHECK_ARG((long)NB * I * H * W < 2147483647L && 4L * NB * O * (H + 1) * (W + 1) < 2147483647L, "conv2d_up_poly: tensor too large");
    SHG_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0, "conv2d_up_poly: x must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    PolyParams p{};
    p.x = x; p.y = y; p.in_scale = in_scale;
    p.NB = NB; p.I = I; p.O = O; p.OP = OP; p.H = H; p.W = W;
    p.a.wu = wu_a; p.b.wu = wu_b; p.wt = wt;
    const int variant = up_poly_plan(p, NB, I, OP, H, W);
    // K split (small grids): slices write partial planes, one small launch adds them
    const size_t out_bytes = (size_t)4 * NB * O * (H + 1) * (W + 1) * sizeof(float);
    int ks = (workspace && ((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(y)) & 15) == 0)
                 ? shg_wino_ksplit((long)(p.a.n_ttiles + p.b.n_ttiles) * p.n_otiles, p.nchunk) : 1;
    while (ks > 1 && (size_t)ks * out_bytes > ws_bytes) ks /= 2;
    p.cps = shg_cdiv(p.nchunk, ks);
    ks = shg_cdiv(p.nchunk, p.cps);
    p.part_stride = 0;
    if (ks > 1) { p.y = (float*)workspace; p.part_stride = (long)(out_bytes / sizeof(float)); }
    const dim3 grid((p.a.n_ttiles + p.b.n_ttiles) * p.n_otiles + shg_cdiv(p.n_strip_tiles, 4), ks);
    if (variant == 0) hipLaunchKernelGGL((conv_poly_up_kernel<8, 8, 0, 8, 8>), grid, dim3(poly::NT), 0, s, p);
    else if (variant == 1) hipLaunchKernelGGL((conv_poly_up_kernel<1, 64, 11, 4, 16>), grid, dim3(poly::NT), 0, s, p);
    else if (variant == 2) hipLaunchKernelGGL((conv_poly_up_kernel<1, 64, 22, 4, 16>), grid, dim3(poly::NT), 0, s, p);
    else if (variant == 3) hipLaunchKernelGGL((conv_poly_up_kernel<1, 64, 43, 4, 16>), grid, dim3(poly::NT), 0, s, p);
    else if (variant == 4) hipLaunchKernelGGL((conv_poly_up_kernel<4, 16, 0, 4, 16>), grid, dim3(poly::NT), 0, s, p);
    else hipLaunchKernelGGL((conv_poly_up_kernel<8, 8, 0, 4, 16>), grid, dim3(poly::NT), 0, s, p);
    SHG_CHECK_LAUNCH();
    if (ks > 1) {
        shg_launch_wino_split_reduce((const float*)workspace, y, ks, 4 * NB, O, H + 1, W + 1, nullptr, nullptr, nullptr, 0, 0.f, 0, 0.f, 1.f, -1.f, nullptr, s);
        SHG_CHECK_LAUNCH();
    }
    return SHG_OK;
}

extern "C" int shg_conv2d_up_poly_f32(const float* x, const float* wt, const float* wu_a, const float* wu_b, float* y, int NB, int I,
                                      int O, int OP, int H, int W, const float* in_scale, void* stream) {
    return shg_conv2d_up_poly_ws_f32(x, wt, wu_a, wu_b, y, NB, I, O, OP, H, W, in_scale, nullptr, 0, stream);
}

// ---- stride-2 3x3 convolution (mode 1 of shg_conv2d_f32 after the FIR pre-filter) -----------------------------------------

extern "C" int shg_conv_weight_prep_down_poly_f32(const float* w, const float* wscale, float* wu_a, float* wu_b, int O, int I, int OP,
                                                  int flip, void* stream) {
    SHG_CHECK_ARG(w && wscale && wu_a && wu_b, "weight_prep_down_poly: null pointer");
    SHG_CHECK_ARG(O >= 1 && I >= 1 && OP % 64 == 0 && OP >= O, "weight_prep_down_poly: bad shape");
    const int nchunk = shg_cdiv(I, poly::KC);
    const long total = (long)OP * nchunk * poly::KC;
    hipLaunchKernelGGL(poly_weight_kernel, dim3(shg_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wscale, wu_a, O, I, OP,
                       nchunk, flip, (int)poly::DA);
    SHG_CHECK_LAUNCH();
    hipLaunchKernelGGL(poly_weight_kernel, dim3(shg_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wscale, wu_b, O, I, OP,
                       nchunk, flip, (int)poly::DB);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// Geometry of the polyphase planes of the filtered input for an OH x OW output: PH2 = OH + 1 rows, pitch PP = OW + 1
// rounded up to a multiple of 4 floats.  1 when shg_conv2d_down_poly_f32 serves the geometry.
extern "C" int shg_conv2d_down_poly_supported(int NB, int I, int O, int OH, int OW) {
    return (OH >= 16 && OW >= 16 && OW % 4 == 0 && OH % 2 == 0 && I <= 128 * poly::KC && NB >= 1 && O >= 1) ? 1 : 0;
}

// y [NB,O,OH,OW] = act(conv3x3_stride2(xf) + bias) * gain + residual, xf given as its four polyphase planes
// xp [4][NB,I,OH+1,PP] (shg_fir_down_planar_f32).
extern "C" int shg_conv2d_down_poly_f32(const float* xp, const float* wu_a, const float* wu_b, float* y, int NB, int I, int O, int OP,
                                        int OH, int OW, int PP, const float* in_scale, const float* bias, int act, float alpha,
                                        float gain, float clamp, const float* residual, void* stream) {
    SHG_CHECK_ARG(xp && wu_a && wu_b && y, "conv2d_down_poly: null pointer");
    SHG_CHECK_ARG(shg_conv2d_down_poly_supported(NB, I, O, OH, OW), "conv2d_down_poly: unsupported geometry (use shg_conv2d_f32 mode 1)");
    SHG_CHECK_ARG(OP % 64 == 0 && OP >= O, "conv2d_down_poly: OP must be a multiple of 64 and >= O");
    SHG_CHECK_ARG(PP % 4 == 0 && PP >= OW + 1, "conv2d_down_poly: plane pitch must be a multiple of 4 and >= OW + 1");
    SHG_CHECK_ARG(4L * NB * I * (OH + 1) * PP < 2147483647L && (long)NB * O * OH * OW < 2147483647L, "conv2d_down_poly: tensor too large");
    SHG_CHECK_ARG(((reinterpret_cast<uintptr_t>(xp) & 15) | (reinterpret_cast<uintptr_t>(y) & 7) | (reinterpret_cast<uintptr_t>(residual) & 7)) == 0,
                  "conv2d_down_poly: xp must be 16-byte, y / residual 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    PolyParams p{};
    p.x = xp; p.y = y; p.in_scale = in_scale; p.bias = bias; p.residual = residual;
    p.NB = NB; p.I = I; p.O = O; p.OP = OP; p.H = OH; p.W = OW; p.PH2 = OH + 1; p.PP = PP;
    p.act = act; p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    p.n_otiles = OP / 64; p.nchunk = shg_cdiv(I, poly::KC);
    p.cps = p.nchunk; p.part_stride = 0;
    // scheme DA: 3x3 output blocks (flat tiling for the generator's block-row lengths)
    p.a.wu = wu_a; p.a.nby = shg_cdiv(OH, 3); p.a.nbx = shg_cdiv(OW, 3);
    const bool flat = p.a.nbx == 11 || p.a.nbx == 22 || p.a.nbx == 43;
    const long w88 = (long)shg_cdiv(p.a.nbx, 8) * shg_cdiv(p.a.nby, 8), w416 = (long)shg_cdiv(p.a.nbx, 16) * shg_cdiv(p.a.nby, 4);
    const bool wide = OW >= 32 && w416 < w88;
    if (flat) { p.a.tiles_x = shg_cdiv(p.a.nby * p.a.nbx, 64); p.a.tiles_y = 1; }
    else { p.a.tiles_x = shg_cdiv(p.a.nbx, wide ? 16 : 8); p.a.tiles_y = shg_cdiv(p.a.nby, wide ? 4 : 8); }
    p.a.n_ttiles = p.a.tiles_x * p.a.tiles_y * NB;
    const dim3 ga(p.a.n_ttiles * p.n_otiles);
    if (flat && p.a.nbx == 11) hipLaunchKernelGGL((conv_poly_down_kernel<poly::DA, 1, 64, 11>), ga, dim3(poly::NT), 0, s, p);
    else if (flat && p.a.nbx == 22) hipLaunchKernelGGL((conv_poly_down_kernel<poly::DA, 1, 64, 22>), ga, dim3(poly::NT), 0, s, p);
    else if (flat) hipLaunchKernelGGL((conv_poly_down_kernel<poly::DA, 1, 64, 43>), ga, dim3(poly::NT), 0, s, p);
    else if (wide) hipLaunchKernelGGL((conv_poly_down_kernel<poly::DA, 4, 16, 0>), ga, dim3(poly::NT), 0, s, p);
    else hipLaunchKernelGGL((conv_poly_down_kernel<poly::DA, 8, 8, 0>), ga, dim3(poly::NT), 0, s, p);
    SHG_CHECK_LAUNCH();
    // scheme DB: 2x2 output blocks + the layer tail
    const bool bwide = OW >= 32, bxwide = OW >= 64;          // 4 x 16 blocks = 8 x 32 pixels; 2 x 32 blocks = 4 x 64 pixels (256-byte rows)
    p.b.wu = wu_b; p.b.nby = OH / 2; p.b.nbx = OW / 2;
    p.b.tiles_x = shg_cdiv(p.b.nbx, bxwide ? 32 : (bwide ? 16 : 8)); p.b.tiles_y = shg_cdiv(p.b.nby, bxwide ? 2 : (bwide ? 4 : 8));
    p.b.n_ttiles = p.b.tiles_x * p.b.tiles_y * NB;
    const dim3 gb(p.b.n_ttiles * p.n_otiles);
    if (bxwide) hipLaunchKernelGGL((conv_poly_down_kernel<poly::DB, 2, 32, 0>), gb, dim3(poly::NT), 0, s, p);
    else if (bwide) hipLaunchKernelGGL((conv_poly_down_kernel<poly::DB, 4, 16, 0>), gb, dim3(poly::NT), 0, s, p);
    else hipLaunchKernelGGL((conv_poly_down_kernel<poly::DB, 8, 8, 0>), gb, dim3(poly::NT), 0, s, p);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}
