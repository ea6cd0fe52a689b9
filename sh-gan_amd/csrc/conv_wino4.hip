// Winograd F(4x4, 3x3) convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Same operator as conv_wino.hip -- the stride-1 3x3 'same' convolutions (pad 1) of conv2d_resample.py:145-147 as called
// from stylegan.py:226-238 and, modulated, stylegan.py:103-193 -- with the larger Winograd tile: every 4x4 block of outputs
// comes from a 6x6 input block with 36 instead of 144 multiplications per (input, output) channel pair (F(2x2,3x3): 64),
//      Y = A^T [ (G g G^T) .* (B^T d B) ] A ,    interpolation points 0, +-1, +-2, inf (Lavin & Gray).
// Arithmetic stays fp32 on exact-fp32 MFMA; the transform constants (1/4, 1/6, 1/24, 2, 4, 5, 8) cost about one decimal
// digit against F(2x2,3x3): measured 1e-5 relative per layer at 512 channels, 5e-6 on the full generator (tolerance of the
// path: 1e-3).  F(2x2,3x3) stays available (kernels.WINO4 = False) and tested.
//
// GEMM view: 36 independent products M_xi[o,t] = sum_i U_xi[i,o] V_xi[i,t] over 64 output channels x 32 blocks (4 x 8
// blocks = 16 x 32 pixels) = 72 accumulator tiles of 32x32 on 8 waves: waves 0-3 own four positions (8 tiles) each, waves
// 4-7 five positions (10 tiles) each -- one wave of either kind per SIMD.  K is consumed in chunks of 8 input channels:
//   * U (pre-transformed weights, [O/64][chunk][k-step][unit][lane]) goes straight into registers: a ring of one chunk
//     (4 k-steps x 8 or 10 operands), every slot re-loaded for the next chunk two MFMAs after the MFMA that read it;
//   * the raw 18 x 40 window of a channel arrives by 16-byte LDS-DMA (waves 4-7, two channels each, issued as inline
//     assembly so that the compiler does not drain the weight prefetch behind it);
//   * waves 0-3 transform two channels each (B^T d B on the 6x6 patch read as conflict-free b128, styles applied) into
//     V [36][8][32] in LDS, stage by stage between their own MFMAs;
//   * V and the raw windows are double buffered, one raw s_barrier per chunk, the MFMA stream skewed one k-step across it.
// DESIGN.md section 5 ("What bounds an fp32-MFMA kernel") has the measurements behind these choices.
// Epilogue: the 36 M_xi of an (o, t) pair live in different waves -> exchanged through LDS in four passes of 16 channels x
// 32 blocks; each thread applies A^T . A, the fused layer tail (demodulation coefficient, noise, bias, lrelu_agc, skip) and
// stores its 4x4 pixels as four 16-byte rows.
#include "shg_common.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __attribute__((aligned(16))) float shg_wino4_zeros[64];

#ifdef SHG_W4_TRACE
#ifndef SHG_W4_TRACE_WG
#define SHG_W4_TRACE_WG 0          // which workgroup records (a late one shows the steady state, 0 the synchronised first round)
#endif
// timeline study (tools/w4_variant.sh trace -DSHG_W4_TRACE=1): workgroup 0 records clock64() at six points of chunks 8..15
__device__ long long shg_wino4_trace_buf[8 * 8 * 8];
extern "C" int shg_wino4_trace_read(long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(shg_wino4_trace_buf), sizeof(shg_wino4_trace_buf));
}
#define W4_TRACE(slot) do { if (blockIdx.x == SHG_W4_TRACE_WG && c >= SHG_W4_TRACE - 1 && c < SHG_W4_TRACE + 6 && lane == 0) shg_wino4_trace_buf[(wave * 8 + (c - (SHG_W4_TRACE - 1))) * 8 + (slot)] = clock64(); } while (0)
#define W4_TRACE_T(slot) do { if (blockIdx.x == SHG_W4_TRACE_WG && (threadIdx.x & 63) == 0) shg_wino4_trace_buf[((threadIdx.x >> 6) * 8 + 7) * 8 + (slot)] = clock64(); } while (0)
#define W4_TRACE_E(slot) do { if (SHG_W4_TRACE == 100 && blockIdx.x == SHG_W4_TRACE_WG && (threadIdx.x & 63) == 0) shg_wino4_trace_buf[(threadIdx.x >> 6) * 64 + (slot)] = clock64(); } while (0)
#else
#define W4_TRACE(slot) do { } while (0)
#define W4_TRACE_T(slot) do { } while (0)
#define W4_TRACE_E(slot) do { } while (0)
#endif

struct Wino4Params {
    const float* x;          // [NB, I, H, W]
    const float* wu;         // transformed weights [OP/64][nchunk][4 k-steps][72 units][64 lanes]
    float* y;                // [NB, O, H, W]
    const float* in_scale;   // [NB, I] or null
    const float* out_scale;  // [NB, O] or null
    const float* bias;       // [O] or null
    const float* noise;      // see noise_mode
    const float* residual;   // like y, added after the activation
    int NB, I, O, OP, H, W;
    int tiles_x, tiles_y;    // tiles per image
    int n_ttiles, n_otiles, nchunk;
    int cps;                 // chunks per K slice (= nchunk when the launch is not split: conv_wino.hip, K-split); slice = blockIdx.y
    long part_stride;        // floats between the slices' partial outputs (0: y itself)
    int noise_mode;          // 0 none, 1 [H,W], 2 [NB,H,W]
    float noise_strength;
    int act;
    float alpha, gain, clamp;
    // timing studies (tools/w4_variant.sh <tag> -DSHG_WINO4_DBG=<bits>: 1 skip weight loads, 2 skip window DMA, 4 skip transform,
    // 8 skip MFMA, 16 skip epilogue); the product build folds every `p.dbg & ...` branch away
#ifdef SHG_WINO4_DBG
    static constexpr int dbg = SHG_WINO4_DBG;
#else
    static constexpr int dbg = 0;
#endif
};

namespace wino4 {
constexpr int KC = 8, BO = 64, BT = 32, NPOS = 36, NW = 8, NT = NW * 64, NUNIT = 2 * NPOS;    // unit = (position, 32-channel block)
constexpr int V_SZ = NPOS * KC * BT;                        // floats per V buffer (36.9 KB)
template <int TY, int TX>
struct Tile {
    static_assert(TY * TX == BT, "32 blocks per tile");
    // a channel's window is PH x PW floats, fetched as NPIECE whole-wave pieces of 64 x 16 bytes; its LDS slot RP is rounded up to
    // whole pieces so that every lane of every piece is active (lanes past the window store zeros into the padding): a
    // partially masked piece becomes a branch, and a branch makes the compiler wait for ALL outstanding loads behind it.
    static constexpr int PH = 4 * TY + 2, PW = 4 * TX + 8, PW4 = PW / 4, PATCH4 = PH * PW4;
    static constexpr int NPIECE = (PATCH4 + 63) / 64, RP = NPIECE * 256, R_SZ = KC * RP;
};
static_assert(2 * V_SZ >= NPOS * 16 * 32, "epilogue exchange buffer lives in the V region");
}   // namespace wino4

__device__ __forceinline__ int wino4_xcd_remap(int bid, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// 1-D input transform of F(4,3): B^T d, B^T = [[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]]
template <class T>
__device__ __forceinline__ void wino4_bt(const T (&d)[6], T (&o)[6]) {
    const T a = d[4] - 4.f * d[2], b = d[3] - 4.f * d[1], c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
    o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    o[1] = a + b; o[2] = a - b;
    o[3] = c + e; o[4] = c - e;
    o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
// 1-D output transform: A^T m, A^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]]
__device__ __forceinline__ void wino4_at(const float (&m)[6], float (&o)[4]) {
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    o[0] = m[0] + s12 + s34;
    o[1] = d12 + 2.f * d34;
    o[2] = s12 + 4.f * s34;
    o[3] = d12 + 8.f * d34 + m[5];
}

template <int TY, int TX>
__global__ __launch_bounds__(512, 2) void conv_wino4_kernel(const Wino4Params p) {
    using namespace wino4;
    using T = Tile<TY, TX>;
    constexpr int PW = T::PW, PW4 = T::PW4, PATCH4 = T::PATCH4, RP = T::RP, R_SZ = T::R_SZ, NPIECE = T::NPIECE;
    __shared__ __attribute__((aligned(16))) float Vl[2 * V_SZ];      // [2][36][KC][32]
    __shared__ __attribute__((aligned(16))) float Rl[2 * R_SZ];      // [2][KC][RP]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    W4_TRACE_T(0);

    const int nwork = p.n_ttiles * p.n_otiles;
    const int work = wino4_xcd_remap(blockIdx.x, nwork);
    const int otile = work / p.n_ttiles;
    const int ttile = work - otile * p.n_ttiles;
    const int txb = ttile % p.tiles_x;
    const int tyb = (ttile / p.tiles_x) % p.tiles_y;
    const int n = ttile / (p.tiles_x * p.tiles_y);
    const int oy0 = tyb * (4 * TY), ox0 = txb * (4 * TX);
    const int o0 = otile * BO;
    const int HW = p.H * p.W;
    const int c0 = blockIdx.y * p.cps, nch = min(p.cps, p.nchunk - c0), iend = min(p.I, (c0 + nch) * KC);      // this workgroup's K slice

    // ---- staging roles: waves 0..3 transform two channels each; waves 4..7 fetch the raw windows of two channels each
    const bool xformer = wave < 4;
    int roff[NPIECE];
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
        const int q = j * 64 + lane;
        const int py = q / PW4, p4 = q - py * PW4;
        const int iy = oy0 - 1 + py, ix = ox0 - 4 + 4 * p4;
        const bool ok = q < PATCH4 && iy >= 0 && iy < p.H && ix >= 0 && ix + 3 < p.W;
        roff[j] = ok ? n * p.I * HW + iy * p.W + ix : -1;
    }
    auto dma_piece = [&](int c, int buf, int q, int j) __attribute__((always_inline)) {
        if (p.dbg & 2) return;
        const int k = (wave - 4) * 2 + q;
        const int ch = (c0 + c) * KC + k;
        const bool chok = ch < iend;                             // (also false for chunks past the end)
        const float* src = (chok && roff[j] >= 0) ? p.x + ((long)roff[j] + (long)ch * HW) : shg_wino4_zeros;
        // Issued as inline assembly, not through __builtin_amdgcn_global_load_lds: the compiler orders every later LDS read (and,
        // with 40+ loads in flight, every use of a loaded register) behind an LDS-DMA it knows about with `s_waitcnt vmcnt(0)`,
        // i.e. it waits for the weight loads issued a few instructions earlier.  The windows are ordered by hand (the counted
        // vmcnt before the chunk's barrier); to the compiler's own counts these are unknown extra loads, which only makes its
        // waits for older loads conservative.
        const unsigned lds = __builtin_amdgcn_readfirstlane(
            (unsigned)(size_t)(__attribute__((address_space(3))) void*)(Rl + buf * R_SZ + k * RP + j * 256));
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(src) : "m0", "memory");
    };
    auto dma_raw = [&](int c, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < NPIECE; ++j) dma_piece(c, buf, q, j);
    };

    // ---- input transform role: channel 2*wave + half, block l31 (ty = l31 / TX, tx = l31 % TX)
    const int tty = l31 / TX, ttx = l31 % TX;
    const float* rbase = Rl + (2 * wave + half) * RP + (4 * tty) * PW + 4 * ttx;         // patch (0,0) = window (4ty, 4tx + 3)
    float* vbase = Vl + (2 * wave + half) * BT + l31;            // + xi*KC*BT
    // styles of this wave's two channels for every chunk, one lane per chunk (up to 128 chunks)
    float sca[2], scb[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int ch = (v * 64 + lane) * KC + 2 * wave;
        sca[v] = (p.in_scale && xformer && ch < p.I) ? p.in_scale[(long)n * p.I + ch] : 1.f;
        scb[v] = (p.in_scale && xformer && ch + 1 < p.I) ? p.in_scale[(long)n * p.I + ch + 1] : 1.f;
    }
    // The transform of one chunk, cut into stages that the main loop places between the wave's MFMAs:
    //   read(r)  : row r of the 6x6 patch as three ds_read_b128 (window columns 4tx .. 4tx+11, the patch is 3..8).  With the
    //              40-float row pitch the 16-lane groups of a b128 read touch all 64 banks once; single-dword reads of the same
    //              patch are 4-way conflicted (lane pitch 16 B) and were a third of the kernel's LDS time;
    //   row(r)   : u[r][.] = d[r][.] B, style applied;
    //   col(j)   : V[.][j] = B^T u[.][j], six positions written to V.
    auto style_of = [&](int c) __attribute__((always_inline)) {
        const int ca = c0 + c;
        const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ca < 64 ? sca[0] : sca[1]), ca & 63));
        const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ca < 64 ? scb[0] : scb[1]), ca & 63));
        return half ? s1 : s0;
    };
    auto tr_read = [&](int r, f32x4 (&w)[3], int buf) __attribute__((always_inline)) {
        if (p.dbg & 4) return;
        const f32x4* rb = reinterpret_cast<const f32x4*>(rbase + buf * R_SZ + r * PW);
#pragma unroll
        for (int q = 0; q < 3; ++q) w[q] = rb[q];
    };
    auto tr_row = [&](int r, const f32x4 (&w)[3], float (&u)[6][6], float sc) __attribute__((always_inline)) {
        if (p.dbg & 4) return;
        const float di[6] = {w[0][3] * sc, w[1][0] * sc, w[1][1] * sc, w[1][2] * sc, w[1][3] * sc, w[2][0] * sc};
        wino4_bt(di, u[r]);
    };
    auto tr_col = [&](int j, const float (&u)[6][6], int buf) __attribute__((always_inline)) {
        if (p.dbg & 4) return;
        float* vb = vbase + buf * V_SZ + j * KC * BT;
        const float di[6] = {u[0][j], u[1][j], u[2][j], u[3][j], u[4][j], u[5][j]};
        float o[6];
        wino4_bt(di, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) vb[i * 6 * KC * BT] = o[i];
    };

    // ---- epilogue geometry (all threads): 16 x 32 (channel, block) items per pass
    float* Mx = Vl;                           // [36][16][32]
    const long plane = (long)p.H * p.W;
    const int o_l = tid >> 5, t_l = tid & 31;
    const int by = t_l / TX, bx = t_l % TX;
    const int oy = oy0 + 4 * by, ox = ox0 + 4 * bx;
    // ---- the two wave classes.  Transform waves (0..3) own 4 positions x 2 channel blocks = 8 accumulator tiles, fetch waves
    // (4..7) own 5 x 2 = 10: each SIMD hosts one wave of either class, so the matrix pipes are loaded evenly (18 tiles) and
    // the transform's temporaries live in the registers the two missing accumulators would take.
    auto body = [&](auto nu_c, auto xf_c) __attribute__((always_inline)) {
        constexpr int NU = decltype(nu_c)::value;                // accumulator tiles of this wave
        constexpr bool XF = decltype(xf_c)::value;
        constexpr int NP = NU / 2;
        const int pfirst = XF ? 4 * wave : 16 + 5 * (wave - 4);
        // weights: a register ring of one chunk; slot (ks, j) = MFMA A operand of unit 2*pfirst + j at k-step ks
        constexpr size_t ustride = (size_t)4 * NUNIT * 64;       // floats per chunk
        const float* ubase = p.wu + ((size_t)otile * p.nchunk * 4 * NUNIT + 2 * pfirst) * 64 + lane + (size_t)c0 * ustride;
        float ur[4][NU];
        auto load_u = [&](int c, int ks) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NU; ++j) ur[ks][j] = ubase[(size_t)c * ustride + (ks * NUNIT + j) * 64];
        };
        f32x16 acc[NU];
#pragma unroll
        for (int j = 0; j < NU; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const float* bbase = Vl + (pfirst * KC + half) * BT + l31;   // + pidx*KC*BT + ks*2*BT

        // prologue: weights of k-steps 0..2 of chunk 0 (slot 3 is filled at the top of the chunk that uses it), the first windows
        // and the first transform
        float b[2][NP];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) load_u(0, ks);
#pragma unroll
        for (int j = 0; j < NU; ++j) ur[3][j] = 0.f;
#pragma unroll
        for (int q = 0; q < NP; ++q) b[1][q] = 0.f;
        if constexpr (!XF) {
            dma_raw(0, 0);
            dma_raw(1, 1);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0), as an instruction the compiler's wait-count tracking sees
        __syncthreads();
        if constexpr (XF) {
            f32x4 w[3];
            float u[6][6];
            const float sc = style_of(0);
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                tr_read(r, w, 0);
                tr_row(r, w, u, sc);
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) tr_col(j, u, 0);
        }
        __syncthreads();

        // Main loop.  Issue order is pinned instruction group by instruction group (sched_barrier): every MFMA is followed by the
        // load that refills its weight slot (one texture instruction per 64-cycle MFMA instead of bursts from eight waves at
        // once), by one piece of the window DMA (fetch waves) or by one stage of the next chunk's input transform (transform
        // waves: ~20 VALU + a few LDS instructions, issued while the MFMA occupies the matrix pipe).  The MFMA stream is skewed by
        // one k-step against the barriers: the last k-step's operands stay in registers and are multiplied after the barrier,
        // while the first B operands of the next chunk are on their way from LDS.
        // Every iteration issues the same loads (past the end: re-fetches / zeros) so the compiler's wait counts for the weight
        // ring stay exact.
#ifdef SHG_W4_PRIO
        if constexpr (XF == (SHG_W4_PRIO == 1)) __builtin_amdgcn_s_setprio(3);      // (arbitration study: 1 = transform waves first, 2 = fetch waves)
#endif
        W4_TRACE_T(1);
        const int last = nch - 1;
        auto fetch = [&](const float* bb, int ks, int pb) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < NP; ++q) b[pb][q] = bb[(q * KC + ks * 2) * BT];
        };
        auto mma = [&](int ks, int pb, int j) __attribute__((always_inline)) {
            if (!(p.dbg & 8)) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ur[ks][j], b[pb][j >> 1], acc[j], 0, 0, 0);
        };
        auto refill = [&](int c, int ks, int j) __attribute__((always_inline)) {
#ifdef SHG_W4_SAMEADDR
            if (!(p.dbg & 1)) ur[ks][j] = ubase[(SHG_W4_SAMEADDR == 2 ? (size_t)(ks * NUNIT + j) * 64 : 0)];   // (study: L1-resident weights)
#else
            if (!(p.dbg & 1)) ur[ks][j] = ubase[(size_t)c * ustride + (ks * NUNIT + j) * 64];
#endif
        };
        for (int c = 0; c < nch; ++c) {
            const int buf = c & 1;
            const int cn = c < last ? c + 1 : last;
            const float* bb = bbase + buf * V_SZ;
            f32x4 w[2][3];
            float u[6][6];
            float sc = 1.f;
            if constexpr (XF) sc = style_of(cn);
            W4_TRACE(0);
            fetch(bb, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // MFMA m of the chunk: group m / NU (0 = k-step 3 of the previous chunk -- zeros the first time round --, 1..3 = k-steps
            // 0..2), unit m % NU.  The load that refills a weight slot is issued RFD MFMAs after the MFMA that read the slot: issued
            // right behind it, the load waits until the MFMA has released the register (about 30 cycles per MFMA when the wave has
            // the pipe to itself).
            constexpr int RFD = 2;
            auto refill_m = [&](int m) __attribute__((always_inline)) {
                if (m / NU == 0) refill(c, 3, m % NU);
                else refill(cn, m / NU - 1, m % NU);
            };
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                mma(3, 1, j);
                if constexpr (!XF) {
                    if (j < 2 * NPIECE) dma_piece(c + 2, buf, j / NPIECE, j % NPIECE);   // raw(c) was consumed during chunk c-1
                }
                if (j >= RFD) refill_m(j - RFD);
                if constexpr (XF) {
                    if (j < 6) tr_read(j, w[j & 1], buf ^ 1);                     // raw(c+1) landed before the previous barrier
                    if (j >= 1 && j < 7) tr_row(j - 1, w[(j - 1) & 1], u, sc);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            W4_TRACE(1);
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                if (ks == 1) W4_TRACE(2);
                if (ks == 2) W4_TRACE(3);
                fetch(bb, ks + 1, (ks + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NU; ++j) {
                    mma(ks, ks & 1, j);
                    refill_m((ks + 1) * NU + j - RFD);
                    if constexpr (XF) {
                        if (ks == 0 && j < 6) tr_col(j, u, buf ^ 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int m = 4 * NU - RFD; m < 4 * NU; ++m) refill_m(m);
            // the last window piece went out ahead of all but 2*NPIECE - 1 - RFD of the chunk's 4*NU weight loads: in-order retirement makes
            // "at most that many outstanding" mean "the window has landed" without waiting for the weights
            W4_TRACE(4);
            if constexpr (!XF) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NU - 2 * NPIECE + 1 + RFD) : "memory");
            // raw barrier: __syncthreads() carries a release fence that the compiler lowers to `s_waitcnt vmcnt(0)`, i.e. a wait for
            // the weight loads just issued.  LDS traffic is ordered by lgkmcnt(0), the DMA by the count above.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            W4_TRACE(5);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            W4_TRACE(6);
        }
#pragma unroll
        for (int j = 0; j < NU; ++j) mma(3, 1, j);

        W4_TRACE_T(2);
        // epilogue: four passes (channel block ob, row half h) of 16 channels x 32 blocks x 36 positions through LDS
        if (p.dbg & 16) {
            if (acc[0][0] == 12345.f) p.y[0] = acc[1][1];
            return;
        }
        // Operands of the fused tail.  Every load is unconditional (absent operands read a block of zeros) and the loads of pass
        // p+1 are issued BEFORE the stores of pass p: a conditional load makes the compiler wait with vmcnt(0), and on this chip
        // vmcnt also counts stores -- the tail then waited for its own previous row to reach memory (1200 cycles per row, 5000 of a
        // pass's 6500).  Out-of-range rows / channels are clamped for the loads and masked for the stores.
        const bool col_ok = oy < p.H && ox < p.W;
        const int oxc = ox < p.W ? ox : 0;
        long rowoff[4];
    #pragma unroll
        for (int i = 0; i < 4; ++i) rowoff[i] = (long)(oy + i < p.H ? oy + i : p.H - 1) * p.W + oxc;   // W % 4 == 0, ox % 4 == 0: 16-byte rows
        f32x4 nz[4];
        {
            const float* np_ = p.noise_mode ? p.noise + (p.noise_mode == 2 ? (long)n * plane : 0) : shg_wino4_zeros;
    #pragma unroll
            for (int i = 0; i < 4; ++i) nz[i] = *reinterpret_cast<const f32x4*>(np_ + (p.noise_mode ? rowoff[i] : 0));
        }
        struct TailOps { float osc, bs; f32x4 rs[4]; };
        auto tail_load = [&](int pass, TailOps& t) __attribute__((always_inline)) {
            const int oo = o0 + (pass >> 1) * 32 + 16 * (pass & 1) + o_l;
            const int oc = oo < p.O ? oo : p.O - 1;
            t.osc = p.out_scale ? p.out_scale[(long)n * p.O + oc] : 1.f;
            t.bs = p.bias ? p.bias[oc] : 0.f;
            const float* rp = p.residual ? p.residual + ((long)n * p.O + oc) * plane : shg_wino4_zeros;
    #pragma unroll
            for (int i = 0; i < 4; ++i) t.rs[i] = *reinterpret_cast<const f32x4*>(rp + (p.residual ? rowoff[i] : 0));
        };
        auto finish = [&](int pass, TailOps& t) __attribute__((always_inline)) {
            const int ob = pass >> 1, h = pass & 1;
            const int o = o0 + ob * 32 + 16 * h + o_l;
            W4_TRACE_E(pass * 8 + 1);
            // A^T m A: rows of m first (over the position columns), then columns
            float tmp[6][4];
    #pragma unroll
            for (int r = 0; r < 6; ++r) {
                float m[6], a4[4];
    #pragma unroll
                for (int cc = 0; cc < 6; ++cc) m[cc] = Mx[((r * 6 + cc) * 16 + o_l) * 32 + t_l];
                wino4_at(m, a4);
    #pragma unroll
                for (int k = 0; k < 4; ++k) tmp[r][k] = a4[k];
            }
            W4_TRACE_E(pass * 8 + 2);
            f32x4 out[4];
    #pragma unroll
            for (int k = 0; k < 4; ++k) {                             // output column k of the block needs tmp[.][k]
                const float col[6] = {tmp[0][k], tmp[1][k], tmp[2][k], tmp[3][k], tmp[4][k], tmp[5][k]};
                float a4[4];
                wino4_at(col, a4);
    #pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = a4[i] * t.osc + nz[i][k] * p.noise_strength + t.bs;
                    v = p.act ? shg_lrelu_agc(v, p.alpha, p.gain, p.clamp) : v * p.gain;
                    out[i][k] = v + t.rs[i][k];
                }
            }
            W4_TRACE_E(pass * 8 + 4);
            if (pass < 3) tail_load(pass + 1, t);                     // ahead of this pass's stores
            W4_TRACE_E(pass * 8 + 5);
            if (o < p.O && col_ok) {
                float* yp = p.y + blockIdx.y * p.part_stride + ((long)n * p.O + o) * plane;
    #pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (oy + i < p.H) *reinterpret_cast<f32x4*>(yp + rowoff[i]) = out[i];
            }
        };

        TailOps tops;
        tail_load(0, tops);
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int ob = pass >> 1, h = pass & 1;
            W4_TRACE_E(pass * 8 + 0);
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = 8 * h + rr;
                    const int lr = (r & 3) + 8 * ((r >> 2) & 1) + 4 * half;       // row inside the 16-row half
                    Mx[((pfirst + q) * 16 + lr) * 32 + l31] = acc[2 * q + ob][r];
                }
            }
            __syncthreads();
            finish(pass, tops);
            W4_TRACE_E(pass * 8 + 3);
            W4_TRACE_T(3 + pass);
            if (pass < 3) __syncthreads();
        }
    };
    if (xformer) body(std::integral_constant<int, 8>{}, std::true_type{});
    else body(std::integral_constant<int, 10>{}, std::false_type{});
}

// U = G g G^T per (o, i), G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]; g = w[o,i] * scale[o].
// Layout wu[otile][chunk][k-step][unit][lane]: unit u = (position u/2, channel block u%2); lane = (i & 1) * 32
// + o % 32 holds the MFMA A operand of k-step (i % 8) / 2.
__global__ __launch_bounds__(256) void wino4_weight_kernel(const float* w, const float* scale, float* wu, int O, int I, int OP,
                                                           int nchunk, int flip) {
    constexpr int KC = wino4::KC;
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
    float gg[6][3];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
        const float a = g[0][cc], b = g[1][cc], c = g[2][cc];
        gg[0][cc] = a * 0.25f;
        gg[1][cc] = (-a - b - c) * (1.f / 6.f);
        gg[2][cc] = (-a + b - c) * (1.f / 6.f);
        gg[3][cc] = a * (1.f / 24.f) + b * (1.f / 12.f) + c * (1.f / 6.f);
        gg[4][cc] = a * (1.f / 24.f) - b * (1.f / 12.f) + c * (1.f / 6.f);
        gg[5][cc] = c;
    }
    const int k = i % KC, chunk = i / KC;
    const int ln = (k & 1) * 32 + (o & 31), ks = k >> 1, obk = (o & 63) >> 5;
    float* base = wu + ((size_t)(o >> 6) * nchunk + chunk) * (4 * wino4::NUNIT * 64);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const float a = gg[r][0], b = gg[r][1], c = gg[r][2];
        const float row[6] = {a * 0.25f, (-a - b - c) * (1.f / 6.f), (-a + b - c) * (1.f / 6.f),
                              a * (1.f / 24.f) + b * (1.f / 12.f) + c * (1.f / 6.f), a * (1.f / 24.f) - b * (1.f / 12.f) + c * (1.f / 6.f), c};
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) {
            const int u = (r * 6 + cc) * 2 + obk;
            base[((size_t)ks * wino4::NUNIT + u) * 64 + ln] = row[cc];
        }
    }
}

// floats of the F(4x4,3x3) weight tensor for (OP, I)
extern "C" long shg_conv_wino4_weight_elems(int OP, int I) {
    return (long)(OP / 64) * shg_cdiv(I, wino4::KC) * 4 * wino4::NUNIT * 64;
}

extern "C" int shg_conv_weight_prep_wino4_f32(const float* w, const float* wscale, float* wu, int O, int I, int OP, int flip,
                                              void* stream) {
    SHG_CHECK_ARG(w && wscale && wu, "weight_prep_wino4: null pointer");
    SHG_CHECK_ARG(O >= 1 && I >= 1 && OP % 64 == 0 && OP >= O, "weight_prep_wino4: bad shape");
    const int nchunk = shg_cdiv(I, wino4::KC);
    const long total = (long)OP * nchunk * wino4::KC;
    hipLaunchKernelGGL(wino4_weight_kernel, dim3(shg_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wscale, wu, O, I, OP,
                       nchunk, flip);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

// 1 when shg_conv2d_wino4_f32 serves the geometry (otherwise shg_conv2d_wino_f32 / shg_conv2d_f32)
extern "C" int shg_conv2d_wino4_supported(int NB, int I, int O, int H, int W) {
    return (H >= 16 && W >= 32 && W % 4 == 0 && I <= 128 * wino4::KC && NB >= 1 && O >= 1) ? 1 : 0;
}

// y = act(out_scale[n,o] * conv3x3_same(x * in_scale[n,i], w) + noise*noise_strength + bias[o]) + residual, stride 1, pad 1.
int shg_wino_ksplit(long tiles, int nchunk);                   // conv_wino.hip
void shg_launch_wino_split_reduce(const float* part, float* y, int ks, int NB, int O, int H, int W, const float* out_scale, const float* bias,
                                  const float* noise, int noise_mode, float noise_strength, int act, float alpha, float gain, float clamp,
                                  const float* residual, hipStream_t s);

static int wino4_plan(Wino4Params& p, int NB, int I, int OP, int H, int W) {
    // tile shape: the same 32 blocks as 16 x 32 pixels, 8 x 64 for W >= 128 or 4 x 128 for W >= 256.  The 8 x 64 window has the
    // same area as the 16 x 32 one (10 x 72 against 18 x 40 floats per channel), the 4 x 128 one a fourth piece per channel
    // (6 x 136), but every output row piece is 256 / 512 contiguous bytes instead of 128, which is what the store path wants
    // (DESIGN section 5): syn512.conv1 1286 -> 1237 -> 1226 us, 256^2 layers 1037 -> 1004 -> 990; at W = 64 no gain (809 vs 813).
    const int shape = (W >= 256 && H >= 4) ? 2 : ((W >= 128 && H >= 8) ? 1 : 0);
    p.tiles_x = shg_cdiv(W, 32 << shape); p.tiles_y = shg_cdiv(H, 16 >> shape);
    p.n_ttiles = p.tiles_x * p.tiles_y * NB; p.n_otiles = OP / 64; p.nchunk = shg_cdiv(I, wino4::KC);
    return shape;
}

// bytes of scratch with which shg_conv2d_wino4_ws_f32 splits this problem along its input channels (0: it will not; see conv_wino.hip)
extern "C" size_t shg_conv2d_wino4_workspace_bytes(int NB, int I, int O, int OP, int H, int W) {
    if (NB < 1 || I < 1 || O < 1 || OP < 64 || H < 1 || W < 1) return 0;
    Wino4Params p{};
    wino4_plan(p, NB, I, OP, H, W);
    const int ks = shg_wino_ksplit((long)p.n_ttiles * p.n_otiles, p.nchunk);
    return ks > 1 ? (size_t)ks * NB * O * H * W * sizeof(float) : 0;
}

extern "C" int shg_conv2d_wino4_ws_f32(const float* x, const float* wu, float* y, int NB, int I, int O, int OP, int H, int W,
                                       const float* in_scale, const float* out_scale, const float* bias, const float* noise,
                                       int noise_mode, float noise_strength, int act, float alpha, float gain, float clamp,
                                       const float* residual, void* workspace, size_t ws_bytes, void* stream) {
    SHG_CHECK_ARG(x && wu && y, "conv2d_wino4: null pointer");
    SHG_CHECK_ARG(shg_conv2d_wino4_supported(NB, I, O, H, W), "conv2d_wino4: unsupported geometry (use shg_conv2d_wino_f32)");
    SHG_CHECK_ARG(OP % 64 == 0 && OP >= O, "conv2d_wino4: OP must be a multiple of 64 and >= O");
    SHG_CHECK_ARG((long)NB * I * H * W < 2147483647L && (long)NB * O * H * W < 2147483647L, "conv2d_wino4: tensor too large");
    SHG_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(noise) |
                    reinterpret_cast<uintptr_t>(residual)) & 15) == 0, "conv2d_wino4: x / y / noise / residual must be 16-byte aligned");
    Wino4Params p{};
    p.x = x; p.wu = wu; p.y = y; p.in_scale = in_scale; p.out_scale = out_scale; p.bias = bias;
    p.noise = noise_mode ? noise : nullptr; p.residual = residual;
    p.NB = NB; p.I = I; p.O = O; p.OP = OP; p.H = H; p.W = W;
    const int shape = wino4_plan(p, NB, I, OP, H, W);
    p.noise_mode = noise ? noise_mode : 0; p.noise_strength = noise_strength;
    p.act = act; p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    int ks = (workspace && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0) ? shg_wino_ksplit((long)p.n_ttiles * p.n_otiles, p.nchunk) : 1;
    const size_t out_bytes = (size_t)NB * O * H * W * sizeof(float);
    while (ks > 1 && (size_t)ks * out_bytes > ws_bytes) ks /= 2;
    p.cps = shg_cdiv(p.nchunk, ks);
    ks = shg_cdiv(p.nchunk, p.cps);
    p.part_stride = 0;
    if (ks > 1) {                             // slices write raw sums; the tail moves to the reduction
        p.y = (float*)workspace; p.part_stride = (long)NB * O * H * W;
        p.out_scale = nullptr; p.bias = nullptr; p.noise = nullptr; p.noise_mode = 0; p.residual = nullptr; p.act = 0; p.gain = 1.f;
    }
    const dim3 grid(p.n_ttiles * p.n_otiles, ks);
    if (shape == 2) hipLaunchKernelGGL((conv_wino4_kernel<1, 32>), grid, dim3(wino4::NT), 0, (hipStream_t)stream, p);
    else if (shape == 1) hipLaunchKernelGGL((conv_wino4_kernel<2, 16>), grid, dim3(wino4::NT), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((conv_wino4_kernel<4, 8>), grid, dim3(wino4::NT), 0, (hipStream_t)stream, p);
    SHG_CHECK_LAUNCH();
    if (ks > 1) {
        shg_launch_wino_split_reduce((const float*)workspace, y, ks, NB, O, H, W, out_scale, bias, noise_mode ? noise : nullptr, noise ? noise_mode : 0,
                                     noise_strength, act, alpha, gain, clamp, residual, (hipStream_t)stream);
        SHG_CHECK_LAUNCH();
    }
    return SHG_OK;
}

extern "C" int shg_conv2d_wino4_f32(const float* x, const float* wu, float* y, int NB, int I, int O, int OP, int H, int W,
                                    const float* in_scale, const float* out_scale, const float* bias, const float* noise,
                                    int noise_mode, float noise_strength, int act, float alpha, float gain, float clamp,
                                    const float* residual, void* stream) {
    return shg_conv2d_wino4_ws_f32(x, wu, y, NB, I, O, OP, H, W, in_scale, out_scale, bias, noise, noise_mode, noise_strength, act, alpha, gain, clamp,
                                   residual, nullptr, 0, stream);
}
