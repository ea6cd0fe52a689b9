// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Alternative to conv_mfma.hip for the stride-1 3x3 'same' convolutions (pad 1) that dominate the generator forward
// (conv2d_resample.py:145-147 as called from stylegan.py:226-238 and, modulated, stylegan.py:103-193): every 2x2 block
// of outputs is computed from a 4x4 input block with 16 instead of 36 multiplications per (input, output) channel pair,
//      Y = A^T [ (G g G^T) .* (B^T d B) ] A ,
// still in exact fp32 arithmetic on the MFMA units (the transform matrices hold 0, +-1, +-1/2 only; measured
// difference to the direct form ~1e-6 relative, tolerance of the path 1e-3).
//
// GEMM view: 16 independent products  M_xi[o, t] = sum_i U_xi[i, o] * V_xi[i, t]   (xi = position in the 4x4 block,
// t = 2x2 output block).  One workgroup = 64 output channels x 64 blocks (4 x 16 blocks = 8 x 32 pixels of one image)
// x all 16 positions = 64 accumulator tiles of 32x32, spread over 16 waves: wave w owns position w (4 tiles), so there
// are four independent instruction streams per SIMD.  K is consumed in chunks of 8 input channels:
//   * U (pre-transformed weights, prepared once per parameter version) never touches LDS: position w's slice is needed
//     by wave w only, so every lane loads its own MFMA A-operands of the next chunk (8 floats, two 16-byte loads from a
//     lane-major layout) into registers while the current chunk is multiplied;
//   * the raw 10 x 40 input window of the chunk's 8 channels arrives by 16-byte LDS-DMA (waves 8-15, one channel each);
//   * waves 0-7 transform one channel each (B^T d B, 32 add/sub per 4x4 block, per-sample style applied here) and write
//     V [16][8][64] to LDS, from where every wave reads the B operands of its position;
//   * V and the raw windows are double buffered: while chunk c is multiplied, raw(c+2) is in flight and V(c+1) is
//     built; one barrier per chunk, and the MFMA stream is skewed one k-step across it so that the matrix pipe has work
//     while the next operands arrive from LDS.
// Epilogue: the 16 M_xi of an (o, t) pair live in 16 different waves -> they are exchanged through LDS in four passes
// of 32 channels x 32 blocks; each thread then applies A^T . A, the fused layer tail (demodulation coefficient, noise,
// bias, lrelu_agc, skip) and stores two pixels at a time (128-byte row segments per 16 lanes).
#include "shg_common.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __attribute__((aligned(16))) float shg_wino_zeros[64];   // zero source for LDS-DMA lanes that fall into padding

struct WinoParams {
    const float* x;          // [NB, I, H, W]
    const float* wu;         // transformed weights [OP/64][nchunk][16][64 lanes][KC]
    float* y;                // [NB, O, H, W]
    const float* in_scale;   // [NB, I] or null
    const float* out_scale;  // [NB, O] or null
    const float* bias;       // [O] or null
    const float* noise;      // see noise_mode
    const float* residual;   // like y, added after the activation
    int NB, I, O, OP, H, W;
    int tiles_x, tiles_y;    // tiles per image (8 x 32 or 16 x 16 pixels)
    int n_ttiles, n_otiles, nchunk;
    int cps;                 // chunks per K slice (= nchunk when the launch is not split); slice = blockIdx.y
    long part_stride;        // floats between the slices' partial outputs (0: y itself)
    int noise_mode;          // 0 none, 1 [H,W], 2 [NB,H,W]
    float noise_strength;
    int act;
    float alpha, gain, clamp;
    // timing studies, -DSHG_ABLATE build only (env SHG_WINO_DBG: 1 skip weight loads, 2 skip window DMA, 4 skip transform,
    // 16 skip epilogue); the product build folds every `p.dbg & ...` branch away
#ifdef SHG_ABLATE
    int dbg;
#else
    static constexpr int dbg = 0;
#endif
};

namespace wino {
// K chunk (input channels per barrier).  The weights never touch LDS (each wave keeps its own position's slice in
// registers), so LDS holds only V and the raw windows: 12 channels = 137 KB of the 160 KB.
#ifndef SHG_WINO_KC
#define SHG_WINO_KC 8
#endif
constexpr int KC = SHG_WINO_KC, BO = 64, BT = 64;
static_assert(KC % 4 == 0 && (KC / 2) % 2 == 0 && KC <= 12, "KC: multiple of 4 with an even number of k-steps");
// A tile is TY x TX blocks (TY * TX = 64) = 2TY x 2TX pixels: 4 x 16 (8 x 32 px) for images at least 32 wide, 8 x 8
// (16 x 16 px) below.  Raw window per channel: rows oy0-1 .. oy0+2TY, columns ox0-4 .. ox0+2TX+3 (16-byte aligned in global
// memory when W % 4 == 0, and every float4 is either inside the image or entirely padding), fetched by two 16-byte LDS-DMAs.
template <int TY, int TX>
struct Tile {
    static_assert(TY * TX == BT, "64 blocks per tile");
    static constexpr int PH = 2 * TY + 2, PW = 2 * TX + 8, PW4 = PW / 4, PATCH4 = PH * PW4;
    static constexpr int RP = PH * PW;                             // floats per channel (400 / 432)
    static constexpr int R_SZ = KC * RP;
    static_assert(PATCH4 <= 128, "window = two wave-wide DMA pieces");
};
constexpr int V_SZ = 16 * KC * BT;
constexpr int NT = 1024;                 // 16 waves: wave w multiplies position w
constexpr int NXF = KC;                  // waves 0..KC-1 transform one channel each; the others issue the window DMAs
static_assert(2 * V_SZ >= 16 * 32 * 32, "epilogue exchange buffer lives in the V region");
}   // namespace wino

__device__ __forceinline__ int wino_xcd_remap(int bid, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <int TY, int TX>
__global__ __launch_bounds__(1024) void conv_wino_kernel(const WinoParams p) {
    using namespace wino;
    using T = Tile<TY, TX>;
    constexpr int PW = T::PW, PW4 = T::PW4, PATCH4 = T::PATCH4, RP = T::RP, R_SZ = T::R_SZ;
    // Two separate LDS objects: the compiler then knows that the LDS-DMA into the raw windows cannot alias the operand
    // reads from V, and does not park an `s_waitcnt vmcnt(0)` (= the full latency of the DMA and weight loads it has just
    // issued) in front of the chunk's MFMAs.
    __shared__ __attribute__((aligned(16))) float Vl[2 * V_SZ];      // [2][16][KC][64]
    __shared__ __attribute__((aligned(16))) float Rl[2 * R_SZ];      // [2][KC][RP]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;

    const int nwork = p.n_ttiles * p.n_otiles;
    const int work = wino_xcd_remap(blockIdx.x, nwork);
    const int otile = work / p.n_ttiles;
    const int ttile = work - otile * p.n_ttiles;
    const int txb = ttile % p.tiles_x;
    const int tyb = (ttile / p.tiles_x) % p.tiles_y;
    const int n = ttile / (p.tiles_x * p.tiles_y);
    const int oy0 = tyb * (2 * TY), ox0 = txb * (2 * TX);
    const int o0 = otile * BO;
    const int HW = p.H * p.W;
    // K slice of this workgroup (split launches: small grids, see shg_conv2d_wino_ws_f32): chunks c0 .. c0 + nch
    const int c0 = blockIdx.y * p.cps, nch = min(p.cps, p.nchunk - c0);

    // ---- staging roles: waves 0..KC-1 transform channel `wave` of the chunk; waves KC..15 fetch the raw windows
    const bool xformer = wave < NXF;
    constexpr int NLD = 16 - NXF;             // loader waves
    constexpr int CPL = (KC + NLD - 1) / NLD; // channels per loader wave
    // raw window: float4 q = j*64 + lane (j = 0, 1; q < 100) of a channel
    int roff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = j * 64 + lane;
        const int py = q / PW4, p4 = q - py * PW4;
        const int iy = oy0 - 1 + py, ix = ox0 - 4 + 4 * p4;
        const bool ok = q < PATCH4 && iy >= 0 && iy < p.H && ix >= 0 && ix + 3 < p.W;
        roff[j] = ok ? n * p.I * HW + iy * p.W + ix : -1;
    }
    const bool ract1 = lane < PATCH4 - 64;
    auto dma_raw = [&](int c, int buf) __attribute__((always_inline)) {
        if (p.dbg & 2) return;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            const int k = (wave - NXF) * CPL + q;
            if (k >= KC) continue;
            const int ch = (c0 + c) * KC + k;
            const bool chok = ch < p.I;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float* src = (chok && roff[j] >= 0) ? p.x + ((long)roff[j] + (long)ch * HW) : shg_wino_zeros;
                if (j == 0 || ract1)       // (inactive lanes write nothing: the window is 100 float4, the second piece 36 lanes)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(Rl + buf * R_SZ + k * RP + j * 256), 16, 0, 0);
            }
        }
    };

    // ---- weights: wave w multiplies Winograd position w only, so its slice of U never needs to be shared -- each lane
    // loads its own MFMA A-operands of a chunk (KC floats: [k-step][channel block]) straight into registers, one chunk ahead
    constexpr int NU = KC / 4;
    const size_t ustride = (size_t)16 * 64 * KC / 4;            // float4 per chunk
    const f32x4* ubase = reinterpret_cast<const f32x4*>(p.wu + (((size_t)otile * p.nchunk * 16 + wave) * 64 + lane) * KC) + (size_t)c0 * ustride;
    f32x4 ua[NU], ub[NU];                                       // even / odd chunks
    auto load_u = [&](f32x4 (&dst)[NU], int c) __attribute__((always_inline)) {
        if (p.dbg & 1) return;
#pragma unroll
        for (int j = 0; j < NU; ++j) dst[j] = ubase[(size_t)c * ustride + j];
    };

    // ---- input transform role: channel `wave`, block `lane` (ty = lane/16, tx = lane%16)
    const int tty = lane / TX, ttx = lane % TX;
    const float* rbase = Rl + wave * RP + (2 * tty) * PW + 2 * ttx + 2;     // window column 2*tx+3 = patch column 2*tx, read from the even column before it
    float* vbase = Vl + wave * BT + lane;                     // + xi*KC*BT
    // styles of channel `wave` of every chunk, one lane per chunk (up to 128 chunks)
    float scv[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int ch = (v * 64 + lane) * KC + wave;
        scv[v] = (p.in_scale && xformer && ch < p.I) ? p.in_scale[(long)n * p.I + ch] : 1.f;
    }
    auto transform = [&](int c, int buf) __attribute__((always_inline)) {
        if (p.dbg & 4) return;
        // style of this wave's channel in chunk c: lane c%64 of the preloaded vector c/64
        const int ca = c0 + c;
        const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ca < 64 ? scv[0] : scv[1]), ca & 63));
        const float* rb = rbase + buf * R_SZ;
        // d B per window row first (two rows of LDS reads in flight at a time: keeps the live set small), then B^T (.)
        float f[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x2 q0 = *reinterpret_cast<const f32x2*>(rb + r * PW);
            const f32x2 q1 = *reinterpret_cast<const f32x2*>(rb + r * PW + 2);
            const f32x2 q2 = *reinterpret_cast<const f32x2*>(rb + r * PW + 4);
            const float d0 = q0[1] * sc, d1 = q1[0] * sc, d2 = q1[1] * sc, d3 = q2[0] * sc;
            f[r][0] = d0 - d2; f[r][1] = d1 + d2; f[r][2] = d2 - d1; f[r][3] = d1 - d3;
            if (r == 1) __builtin_amdgcn_sched_barrier(0);
        }
        float* vb = vbase + buf * V_SZ;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            vb[(0 * 4 + j) * KC * BT] = f[0][j] - f[2][j];
            vb[(1 * 4 + j) * KC * BT] = f[1][j] + f[2][j];
            vb[(2 * 4 + j) * KC * BT] = f[2][j] - f[1][j];
            vb[(3 * 4 + j) * KC * BT] = f[1][j] - f[3][j];
        }
    };

    // ---- MFMA role: position `wave`; 2 channel blocks x 2 pixel-block blocks
    f32x16 acc[2][2];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ob][tb][r] = 0.f;
    const float* bbase = Vl + (wave * KC + half) * BT + l31;     // + ks*2*BT + tb*32

    // ---- prologue
    load_u(ua, 0);
    if (!xformer) {
        dma_raw(0, 0);
        if (nch > 1) dma_raw(1, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (xformer) transform(0, 0);
    __syncthreads();

    // The MFMA stream is skewed by one k-step against the barriers: the operands of a chunk's last k-step are kept in
    // registers and multiplied after the barrier, while the first B operands of the next chunk are on their way from
    // LDS -- the matrix pipe has work the moment the barrier releases.
    float b[2][2], apend[2];
    auto mma = [&](float a0, float a1, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) acc[0][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[buf][tb], acc[0][tb], 0, 0, 0);
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) acc[1][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[buf][tb], acc[1][tb], 0, 0, 0);
    };
    // one chunk; PAR = c & 1 selects the register buffer holding its weights (the other receives chunk c+1)
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
        if (c > 0) mma(apend[0], apend[1], 1);                  // last k-step of the previous chunk
        __builtin_amdgcn_sched_barrier(0);
        if (more) load_u(unxt, c + 1);                          // (the previous chunk no longer reads this buffer)
        if (!xformer) {
            if (c + 2 < nch) dma_raw(c + 2, PAR);          // raw(c) was consumed during chunk c-1
        } else if (more) {
            transform(c + 1, PAR ^ 1);                          // raw(c+1) landed before the previous barrier
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
        // the last k-step's B operands stay in b[1]; their LDS reads must have completed before the buffers are handed back
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();     // (unconditional: a path without it would leave the weight loads pending for the compiler's wait-count tracking)
    };
    for (int c = 0; c < nch; c += 2) {
        chunk(std::integral_constant<int, 0>{}, c);
        if (c + 1 < nch) chunk(std::integral_constant<int, 1>{}, c + 1);
    }
    mma(apend[0], apend[1], 1);
    if (p.dbg & 16) return;

    // ---- epilogue: the 16 M_xi of an (o, t) pair live in 16 waves -> exchanged through LDS (the V buffers), four passes
    // of 32 channels x 32 blocks; inverse transform and fused layer tail, one (channel, block) item per thread and pass.
    float* Mx = Vl;                           // [16][32][32]
    const long plane = (long)p.H * p.W;
    const int o_l = tid >> 5, t_l = tid & 31;
    // per-channel and per-pixel operands of all passes are requested up front (one latency, not four)
    float osc[2], bsv[2];
    f32x2 nzv[2][2];                          // [tb][row]
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
        const int o = min(o0 + ob * 32 + o_l, p.O - 1);
        osc[ob] = p.out_scale ? p.out_scale[(long)n * p.O + o] : 1.f;
        bsv[ob] = p.bias ? p.bias[o] : 0.f;
    }
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        const int t = tb * 32 + t_l;
        const int oy = oy0 + 2 * (t / TX), ox = ox0 + 2 * (t % TX);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            nzv[tb][i] = f32x2{0.f, 0.f};
            if (p.noise_mode && oy + i < p.H && ox < p.W) {       // W % 4 == 0 and ox even: both pixels inside, 8-byte aligned
                const float* np_ = p.noise + (p.noise_mode == 2 ? (long)n * plane : 0) + (long)(oy + i) * p.W + ox;
                nzv[tb][i] = *reinterpret_cast<const f32x2*>(np_);
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
        // A^T m A, A^T = [[1,1,1,0],[0,1,-1,-1]]
        float t0[4], t1[4];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            t0[cc] = m[0 * 4 + cc] + m[1 * 4 + cc] + m[2 * 4 + cc];
            t1[cc] = m[1 * 4 + cc] - m[2 * 4 + cc] - m[3 * 4 + cc];
        }
        float yv[2][2];
        yv[0][0] = t0[0] + t0[1] + t0[2]; yv[0][1] = t0[1] - t0[2] - t0[3];
        yv[1][0] = t1[0] + t1[1] + t1[2]; yv[1][1] = t1[1] - t1[2] - t1[3];
        const int t = tb * 32 + t_l;
        const int oy = oy0 + 2 * (t / TX), ox = ox0 + 2 * (t % TX);
        const int o = o0 + ob * 32 + o_l;
        if (o < p.O && oy < p.H && ox < p.W) {
            const long base = ((long)n * p.O + o) * plane;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (oy + i >= p.H) continue;
                const int pix = (oy + i) * p.W + ox;
                f32x2 rs = f32x2{0.f, 0.f};
                if (p.residual) rs = *reinterpret_cast<const f32x2*>(p.residual + base + pix);
                f32x2 out;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    float v = yv[i][jj] * osc[ob] + nzv[tb][i][jj] * p.noise_strength + bsv[ob];
                    v = p.act ? shg_lrelu_agc(v, p.alpha, p.gain, p.clamp) : v * p.gain;
                    out[jj] = v + rs[jj];
                }
                *reinterpret_cast<f32x2*>(p.y + blockIdx.y * p.part_stride + base + pix) = out;      // W % 4 == 0, ox even: aligned, both pixels inside
            }
        }
        if (pass < 3) __syncthreads();
    }
}

// U = G g G^T per (o, i), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]; g = w[o,i] * scale[o] (flip = true convolution).
// Layout wu[otile][chunk][xi][lane][KC]: lane = (i & 1) * 32 + o % 32 holds, for its position xi, the MFMA A-operands of the
// chunk in the order [k-step = (i % KC) / 2][channel block = (o % 64) / 32] -- KC contiguous floats per lane, 64 lanes contiguous.
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* w, const float* scale, float* wu, int O, int I, int OP,
                                                          int nchunk, int flip) {
    constexpr int KC = wino::KC;
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
    float gg[4][3];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
        gg[0][cc] = g[0][cc];
        gg[1][cc] = 0.5f * (g[0][cc] + g[1][cc] + g[2][cc]);
        gg[2][cc] = 0.5f * (g[0][cc] - g[1][cc] + g[2][cc]);
        gg[3][cc] = g[2][cc];
    }
    const int k = i % KC, chunk = i / KC;
    const int ln = (k & 1) * 32 + (o & 31), slot = (k >> 1) * 2 + ((o & 63) >> 5);
    float* dst = wu + ((((long)(o >> 6) * nchunk + chunk) * 16) * 64 + ln) * KC + slot;      // + xi * 64 * KC
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        dst[(long)(r * 4 + 0) * 64 * KC] = gg[r][0];
        dst[(long)(r * 4 + 1) * 64 * KC] = 0.5f * (gg[r][0] + gg[r][1] + gg[r][2]);
        dst[(long)(r * 4 + 2) * 64 * KC] = 0.5f * (gg[r][0] - gg[r][1] + gg[r][2]);
        dst[(long)(r * 4 + 3) * 64 * KC] = gg[r][2];
    }
}

// input channels per weight chunk of the Winograd layout (sizes wu: [OP/64][ceil(I/chunk)][16][64][chunk] floats)
extern "C" int shg_conv_wino_chunk(void) { return wino::KC; }

// w [O,I,3,3], wscale [O] (per-output-channel factor: the demodulation pre-normalisation * gain of shg_conv_weight_prep_f32,
// or all `gain`), wu out (OP = O rounded up to 64; padding zero filled).
extern "C" int shg_conv_weight_prep_wino_f32(const float* w, const float* wscale, float* wu, int O, int I, int OP, int flip,
                                             void* stream) {
    SHG_CHECK_ARG(w && wscale && wu, "weight_prep_wino: null pointer");
    SHG_CHECK_ARG(O >= 1 && I >= 1 && OP % 64 == 0 && OP >= O, "weight_prep_wino: bad shape");
    const int nchunk = shg_cdiv(I, wino::KC);
    const long total = (long)OP * nchunk * wino::KC;
    hipLaunchKernelGGL(wino_weight_kernel, dim3(shg_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wscale, wu, O, I, OP,
                       nchunk, flip);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// y = act(out_scale[n,o] * conv3x3_same(x * in_scale[n,i], w) + noise*noise_strength + bias[o]) + residual, stride 1, pad 1.
// ---- K-split of the Winograd kernels (this file and conv_wino4.hip).  A workgroup owns one (tile, 64 output channels) pair and walks ALL
// input channels; 512-channel layers at 16^2 / 32^2 are 64-128 such pairs at batch 8 (32-64 at the path-length pass's batch 4): half
// of the chip or less, and a launch takes the same 150 / 200 us at batch 2 and at batch 16.  With a workspace the channel chunks are cut
// into `ks` slices (blockIdx.y), every slice writes its raw partial output, and this kernel sums them and applies the layer tail the
// unsplit kernel applies in its store pass (same expression).
__global__ __launch_bounds__(256) void wino_split_reduce_kernel(const float* part, float* y, int ks, long total, long plane, int O, const float* out_scale,
                                                                const float* bias, const float* noise, int noise_mode, float noise_strength, int act,
                                                                float alpha, float gain, float clamp, const float* residual) {
    for (long e4 = (long)blockIdx.x * 256 + threadIdx.x; e4 * 4 < total; e4 += (long)gridDim.x * 256) {
        const long e = e4 * 4;                                    // W % 4 == 0: four pixels of one row
        f32x4 v = *reinterpret_cast<const f32x4*>(part + e);
        for (int s = 1; s < ks; ++s) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(part + (long)s * total + e);
            v[0] += q[0]; v[1] += q[1]; v[2] += q[2]; v[3] += q[3];
        }
        const long no = e / plane;
        const long pix = e - no * plane;
        const int n = (int)(no / O), o = (int)(no - (long)n * O);
        const float osc = out_scale ? out_scale[(long)n * O + o] : 1.f, bs = bias ? bias[o] : 0.f;
        f32x4 nz = {0.f, 0.f, 0.f, 0.f}, rs = {0.f, 0.f, 0.f, 0.f};
        if (noise_mode) nz = *reinterpret_cast<const f32x4*>(noise + (noise_mode == 2 ? (long)n * plane : 0) + pix);
        if (residual) rs = *reinterpret_cast<const f32x4*>(residual + e);
        f32x4 out;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = v[j] * osc + nz[j] * noise_strength + bs;
            t = act ? shg_lrelu_agc(t, alpha, gain, clamp) : t * gain;
            out[j] = t + rs[j];
        }
        *reinterpret_cast<f32x4*>(y + e) = out;
    }
}

void shg_launch_wino_split_reduce(const float* part, float* y, int ks, int NB, int O, int H, int W, const float* out_scale, const float* bias,
                                  const float* noise, int noise_mode, float noise_strength, int act, float alpha, float gain, float clamp,
                                  const float* residual, hipStream_t s) {
    const long total = (long)NB * O * H * W;
    long grid = (total / 4 + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(wino_split_reduce_kernel, dim3((unsigned)grid), dim3(256), 0, s, part, y, ks, total, (long)H * W, O, out_scale, bias, noise,
                       noise_mode, noise_strength, act, alpha, gain, clamp, residual);
}

// K slices for a grid of `tiles` workgroups over `nchunk` channel chunks: fill 256 CUs once, at least 8 chunks per slice
int shg_wino_ksplit(long tiles, int nchunk) {
    int ks = 1;
    while (tiles * ks < 256 && nchunk / (ks * 2) >= 8) ks *= 2;
    return ks;
}

static void wino_plan(WinoParams& p, int NB, int I, int OP, int H, int W) {
    const bool wide = W >= 32;               // 8 x 32 pixel tiles, else 16 x 16
    p.tiles_x = shg_cdiv(W, wide ? 32 : 16); p.tiles_y = shg_cdiv(H, wide ? 8 : 16);
    p.n_ttiles = p.tiles_x * p.tiles_y * NB; p.n_otiles = OP / 64; p.nchunk = shg_cdiv(I, wino::KC);
}

// bytes of scratch with which shg_conv2d_wino_ws_f32 splits this problem along its input channels (0: it will not)
extern "C" size_t shg_conv2d_wino_workspace_bytes(int NB, int I, int O, int OP, int H, int W) {
    if (NB < 1 || I < 1 || O < 1 || OP < 64 || H < 1 || W < 1) return 0;
    WinoParams p{};
    wino_plan(p, NB, I, OP, H, W);
    const int ks = shg_wino_ksplit((long)p.n_ttiles * p.n_otiles, p.nchunk);
    return ks > 1 ? (size_t)ks * NB * O * H * W * sizeof(float) : 0;
}

extern "C" int shg_conv2d_wino_ws_f32(const float* x, const float* wu, float* y, int NB, int I, int O, int OP, int H, int W,
                                      const float* in_scale, const float* out_scale, const float* bias, const float* noise,
                                      int noise_mode, float noise_strength, int act, float alpha, float gain, float clamp,
                                      const float* residual, void* workspace, size_t ws_bytes, void* stream) {
    SHG_CHECK_ARG(x && wu && y, "conv2d_wino: null pointer");
    SHG_CHECK_ARG(NB >= 1 && I >= 1 && O >= 1 && H >= 1 && W >= 1, "conv2d_wino: empty tensor");
    SHG_CHECK_ARG(OP % 64 == 0 && OP >= O, "conv2d_wino: OP must be a multiple of 64 and >= O");
    SHG_CHECK_ARG((long)NB * I * H * W < 2147483647L && (long)NB * O * H * W < 2147483647L, "conv2d_wino: tensor too large");
    SHG_CHECK_ARG(I <= 128 * wino::KC, "conv2d_wino: too many input channels");
    SHG_CHECK_ARG(W % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "conv2d_wino: needs W %% 4 == 0 and a 16-byte aligned x (use shg_conv2d_f32 otherwise)");
    SHG_CHECK_ARG(((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(noise) | reinterpret_cast<uintptr_t>(residual)) & 7) == 0,
                  "conv2d_wino: y / noise / residual must be 8-byte aligned");
    WinoParams p{};
    p.x = x; p.wu = wu; p.y = y; p.in_scale = in_scale; p.out_scale = out_scale; p.bias = bias;
    p.noise = noise_mode ? noise : nullptr; p.residual = residual;
    p.NB = NB; p.I = I; p.O = O; p.OP = OP; p.H = H; p.W = W;
    wino_plan(p, NB, I, OP, H, W);
    p.noise_mode = noise ? noise_mode : 0; p.noise_strength = noise_strength;
    p.act = act; p.alpha = alpha; p.gain = gain; p.clamp = clamp;
#ifdef SHG_ABLATE
    { const char* d = getenv("SHG_WINO_DBG"); p.dbg = d ? atoi(d) : 0; }
#endif
    // K split: only with scratch for it, 16-byte aligned operands of the reduction (its loads are float4)
    int ks = workspace ? shg_wino_ksplit((long)p.n_ttiles * p.n_otiles, p.nchunk) : 1;
    const size_t out_bytes = (size_t)NB * O * H * W * sizeof(float);
    while (ks > 1 && (size_t)ks * out_bytes > ws_bytes) ks /= 2;
    if (((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(noise) | reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(workspace)) & 15) != 0) ks = 1;
    p.cps = shg_cdiv(p.nchunk, ks);
    ks = shg_cdiv(p.nchunk, p.cps);
    p.part_stride = 0;
    if (ks > 1) {                             // slices write raw sums; the tail moves to the reduction
        p.y = (float*)workspace; p.part_stride = (long)NB * O * H * W;
        p.out_scale = nullptr; p.bias = nullptr; p.noise = nullptr; p.noise_mode = 0; p.residual = nullptr; p.act = 0; p.gain = 1.f;
    }
    const bool wide = W >= 32;
    const dim3 grid(p.n_ttiles * p.n_otiles, ks);
    if (wide) hipLaunchKernelGGL((conv_wino_kernel<4, 16>), grid, dim3(wino::NT), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((conv_wino_kernel<8, 8>), grid, dim3(wino::NT), 0, (hipStream_t)stream, p);
    SHG_CHECK_LAUNCH();
    if (ks > 1) {
        shg_launch_wino_split_reduce((const float*)workspace, y, ks, NB, O, H, W, out_scale, bias, noise_mode ? noise : nullptr, noise ? noise_mode : 0,
                                     noise_strength, act, alpha, gain, clamp, residual, (hipStream_t)stream);
        SHG_CHECK_LAUNCH();
    }
    return SHG_OK;
}

extern "C" int shg_conv2d_wino_f32(const float* x, const float* wu, float* y, int NB, int I, int O, int OP, int H, int W,
                                   const float* in_scale, const float* out_scale, const float* bias, const float* noise,
                                   int noise_mode, float noise_strength, int act, float alpha, float gain, float clamp,
                                   const float* residual, void* stream) {
    return shg_conv2d_wino_ws_f32(x, wu, y, NB, I, O, OP, H, W, in_scale, out_scale, bias, noise, noise_mode, noise_strength, act, alpha, gain, clamp,
                                  residual, nullptr, 0, stream);
}
