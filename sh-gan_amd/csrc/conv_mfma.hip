// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces, for the generator forward path, what the reference delegates to cuDNN through
//   lib/model_zoo/stylegan_utils/conv2d_gradfix.py:35-43,109-116  (F.conv2d / F.conv_transpose2d)
// as called from conv2d_resample.py:26-51 and stylegan.py:103-193 (modulated_conv2d).
//
// Formulation (MI355X-first, not a translation of the grouped-conv trick of stylegan.py:187-190):
//   y[n,o,P] = act( out_scale[n,o] * sum_{i,t} Wt[i,t,o] * (x[n,i,P*S + d_t] * in_scale[n,i])
//                   + noise + bias[o] ) + residual
// i.e. the *non-fused* modulation algebra (stylegan.py:172-181): the weight matrix is shared by the
// whole batch, so one [I*taps, O] operand stays L2/LDS resident while all samples stream through.
// GEMM view: M = O (A operand = weights), N = pixels (B operand = im2col of x, built in LDS from a
// halo patch), K = I*taps.
//
// Kernel variants (one template):
//   * tap-list conv  (UP = false): stride-1 'same', stride-2, single-tap 1x1;
//   * all-phase transposed stride-2 3x3 conv (UP = true): one pass over the low-resolution input
//     produces the four sub-pixel phases of the (2H+1)x(2W+1) result at once -- each of the nine taps
//     feeds exactly one phase, so the MFMA work equals the reference's conv_transpose2d and the input
//     and weights are staged once instead of four times.
// Tiling: workgroup = WO x WP waves; each wave owns (MO*32) x (NP*32) outputs = MO*NP accumulators of
// 32x32 (x4 phases for UP).  K is consumed in chunks of KC input channels: weights [KC*NTAPS][BO] and
// the input patch [PATCH][KC+1] live in LDS; every MFMA covers two channels of one tap (lanes 0-31:
// channel c, lanes 32-63: channel c+1).  Operand reads are conflict-free ds_read_b32 with immediate
// offsets, prefetched one step ahead of the MFMAs that consume them.
// Two families of instantiations:
//   * DB (large images): 8 or 16 waves, one workgroup per CU, double-buffered LDS and one barrier per
//     chunk.  Weights go global -> LDS by LDS-DMA (1 KiB per wave instruction, no registers); the patch
//     is register-staged (styles applied on the way) and handed over to LDS by the wave groups of a
//     SIMD at evenly spread points of the chunk, so some wave always has MFMAs to issue.  Workgroups
//     beyond the last full round over the CUs are cut into K-slices (tail split) and finished by a tiny
//     second launch of the same kernel.
//   * single buffer (small images / 64-channel tiles): 4 waves, 2-3 workgroups per CU cover each
//     other's barrier gaps; small grids are split along K (partial sums to a workspace + a fused
//     reduce/epilogue kernel) so that 4x4..16x16 layers still fill 256 CUs.
// The stride-1 3x3 layers of images >= 16x16 normally run on the Winograd kernel (conv_wino.hip); this
// file serves them when W % 4 != 0 and for everything else (stride 2, transposed, 1x1, tiny images).
#include "shg_common.h"
#include <stdlib.h>
#include <utility>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in registers (HIP's float4 struct copies may not)

struct ConvParams {
    const float* x;          // [NB, I, H, W]
    const float* wt;         // prepped weights [wgroups][OP/64][IP*NTAPS][64]  (column-blocked: a tile's chunk is contiguous)
    float* y;                // output (see out_mode)
    float* part;             // split-K partial sums [ksplit][...same indexing as y...] or null
    const float* in_scale;   // [NB, I] or null
    const float* out_scale;  // [NB, O] or null
    const float* bias;       // [O] or null
    const float* noise;      // see noise_mode
    const float* residual;   // like y, added after the activation
    int NB, I, O, OP;        // OP = O rounded up to 64
    int IPK;                 // IP*NTAPS rows per 64-column weight block (IP = I rounded up to 32)
    int H, W;
    int OHp, OWp;            // output grid computed by this launch
    int OHt, OWt;            // full output tensor extent
    int S;                   // input stride
    int dy0, dx0;            // patch origin = (oy0*S + dy0, ox0*S + dx0)
    int span_y, span_x;      // tap span: patch rows = (TH-1)*S + span_y, cols = (TW-1)*S + span_x
    // Pixel tiles come in up to three classes, each a rectangular region of the launch grid with its own tile
    // shape: the transposed conv covers u < H, v < W with well-shaped tiles and the extra row u = H / column
    // v = W of its (H+1) x (W+1) grid with thin strips; every other convolution has one class.
    struct TileClass {
        int n_tiles;             // tiles of this class (per o-tile, per K-slice)
        int TW, TH, TN;          // TW x TH pixels of TN images per tile (TW*TH*TN <= BP)
        int tiles_x, tiles_y;
        int oy_base, ox_base, rows, cols;   // region [oy_base, oy_base+rows) x [ox_base, ox_base+cols)
        int PH, PW, PATCH;       // input patch of a tile: PATCH = TN*PH*PW elements per channel
    } cls[3];
    int patch_max, tn_max;   // LDS sizing
    int n_ptiles, n_otiles;
    int wgroups;
    long wstride;
    int ksplit, i_per_slice;
    long part_stride;        // elements per split-K slice
    int out_mode;            // UP only: 0 = interleaved [NB,O,2H+1,2W+1]; 1 = planar [4][NB,O,H+1,W+1]
    int noise_mode;          // 0 none, 1 [OHt,OWt], 2 [NB,OHt,OWt]
    float noise_strength;
    int act;
    float alpha, gain, clamp;
    int tap_off[9];          // LDS patch offset (in patch elements) of tap t
    // Tail split (one-workgroup-per-CU kernels): the workgroups beyond the last full round over the CUs are cut into
    // tail_ks K-slices each, so the last, partially filled round costs 1/tail_ks of a round; their raw accumulators go
    // to tail_ws [tile][slice][reg][thread] and a second, tiny launch (fix = 1) sums them and runs the epilogue.
    int tail_first, tail_ks, tail_ips, fix;
    float* tail_ws;
    int raw_reduce;          // split-K tail only sums the slices (transposed conv: its epilogue lives in the FIR kernel)
    // Ablation bits for kernel timing studies -- only in the -DSHG_ABLATE build used by tools/ (env SHG_CONV_DBG: 1 skip W
    // loads, 2 skip X loads, 4 skip LDS stores, 8 skip barriers, 16 skip epilogue, 32 no tail split).  The product build
    // has no such switch: `dbg` is the constant 0 there and every `p.dbg & ...` branch folds away.
#ifdef SHG_ABLATE
    int dbg;
#else
    static constexpr int dbg = 0;
#endif
};

// Compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{}).  Used for the MFMA
// streams so that every accumulator / operand index is a constant no matter what the loop unroller decides.
template <class F, int... S>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, S...>) {
    (f(std::integral_constant<int, S>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// Bijective XCD-aware remap (blocks b, b+8, ... share an XCD and its L2): every XCD walks a
// contiguous range of the o-tile-major work list, so its L2 holds one weight slice at a time.
__device__ __forceinline__ int xcd_remap(int bid, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ float conv_epilogue(const ConvParams& p, float v, int n, int o, long idx, float nz) {
    if (p.out_scale) v *= p.out_scale[n * p.O + o];
    v += nz;
    if (p.bias) v += p.bias[o];
    v = p.act ? shg_lrelu_agc(v, p.alpha, p.gain, p.clamp) : v * p.gain;
    if (p.residual) v += p.residual[idx];
    return v;
}

// DB = double-buffered LDS (one barrier per chunk, staging hand-over in the middle of the MFMA block; meant for
// 8-wave workgroups, 1 per CU, whose two wave groups hand over at different times).  !DB = single LDS buffer,
// two barriers per chunk, small footprint: 3 workgroups per CU cover each other's barrier / staging gaps.
// OCC = waves per SIMD the register allocator must leave room for.
template <int NTAPS, int KC, int MO, int NP, int WO, int WP, int XQ, bool UP, bool DB, int OCC>
__global__ __launch_bounds__(WO * WP * 64, OCC) void conv_mfma_kernel(const ConvParams p) {
    constexpr int BO = MO * 32 * WO;
    constexpr int NT = WO * WP * 64;
    constexpr int ROWS = KC * NTAPS;
    constexpr int XP = KC + 1;       // odd pitch: conflict-free staging writes and B reads
    constexpr int NPH = UP ? 4 : 1;
    constexpr int V4 = ROWS * BO / 4;
    constexpr int PER = (V4 + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // LDS: two weight buffers [ROWS][BO], the tile's input scales [TN][i_per_slice] (modulated layers only),
    // two patch buffers [PATCH][XP].  Chunk c lives in buffer c&1: while chunk c is multiplied, chunk c+1 is
    // written to the other buffer in the middle of the MFMA block -> one barrier per chunk, no exposed store.
    constexpr int WSZ = ROWS * BO;
    constexpr int NBUF = DB ? 2 : 1;
    float* Wl = smem;
    float* Sl = smem + NBUF * WSZ;
    float* Xl = Sl + (p.in_scale ? p.tn_max * p.i_per_slice : 0);
    const int XSZ = XP * p.patch_max;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wo = wave / WP, wp = wave % WP;

    const int nwork = p.n_ptiles * p.n_otiles;
    int kslice, work, tslice = -1;                  // tslice >= 0: this block is K-slice `tslice` of tail tile `work`
    if (p.fix) { kslice = 0; work = p.tail_first + blockIdx.x; }
    else if ((int)blockIdx.x >= p.tail_first && p.ksplit == 1) {
        const int tb = blockIdx.x - p.tail_first;
        kslice = 0; work = p.tail_first + tb / p.tail_ks; tslice = tb - (tb / p.tail_ks) * p.tail_ks;
    } else if (p.ksplit == 1) { kslice = 0; work = xcd_remap(blockIdx.x, p.tail_first); }
    else { kslice = blockIdx.x / nwork; work = xcd_remap(blockIdx.x - kslice * nwork, nwork); }
    const int otile = work / p.n_ptiles;
    int ptile = work - otile * p.n_ptiles;
    int ci = 0;
    if (ptile >= p.cls[0].n_tiles) { ptile -= p.cls[0].n_tiles; ci = 1; if (ptile >= p.cls[1].n_tiles) { ptile -= p.cls[1].n_tiles; ci = 2; } }
    const ConvParams::TileClass& tc = p.cls[ci];
    const int TW = tc.TW, TH = tc.TH, TN = tc.TN, PH_ = tc.PH, PW_ = tc.PW, PATCH = tc.PATCH;
    const int txb = ptile % tc.tiles_x;
    const int tyb = (ptile / tc.tiles_x) % tc.tiles_y;
    const int tnb = ptile / (tc.tiles_x * tc.tiles_y);
    const int THW = TH * TW;
    const int n0 = tnb * TN, oy0 = tc.oy_base + tyb * TH, ox0 = tc.ox_base + txb * TW;
    const int oy_end = tc.oy_base + tc.rows, ox_end = tc.ox_base + tc.cols;
    const int o0 = otile * BO;
    const int HW = p.H * p.W;
    const int PHW = PH_ * PW_;
    const int i_begin = tslice >= 0 ? tslice * p.tail_ips : kslice * p.i_per_slice;
    const int i_end = p.fix ? i_begin : min(p.I, i_begin + (tslice >= 0 ? p.tail_ips : p.i_per_slice));

    // ---- input scales (styles) of the tile's images -> LDS, applied when the patch is written to LDS
    if (p.in_scale) {
        const int span = i_end - i_begin;
        for (int e = tid; e < TN * span; e += NT) {
            const int tn = e / span, ii = e - tn * span;
            const int n = n0 + tn;
            Sl[tn * p.i_per_slice + ii] = n < p.NB ? p.in_scale[(long)n * p.I + i_begin + ii] : 0.f;
        }
    }

    // ---- per-lane staging assignment: patch elements q = tid + k*NT -> global offset (or -1 = padding)
    int xoff[XQ], xsn[XQ];
#pragma unroll
    for (int k = 0; k < XQ; ++k) {
        const int q = tid + k * NT;
        const int tn = q / PHW;
        const int rem = q - tn * PHW;
        const int py = rem / PW_, px = rem - py * PW_;
        const int n = n0 + tn;
        const int iy = oy0 * p.S + p.dy0 + py, ix = ox0 * p.S + p.dx0 + px;
        const bool ok = (q < PATCH) && (n < p.NB) && (iy >= 0) && (iy < p.H) && (ix >= 0) && (ix < p.W);
        xoff[k] = ok ? (n * p.I * HW + iy * p.W + ix) : -1;
        xsn[k] = tn * p.i_per_slice;      // row of this element's image in the LDS scale table
    }

    // ---- per-lane B-fragment base inside the patch (pixel j of the tile -> patch element)
    const float* xbase[NP];
#pragma unroll
    for (int np = 0; np < NP; ++np) {
        const int j = (wp * NP + np) * 32 + l31;
        int tn = j / THW;
        const int rem = j - tn * THW;
        int ty = rem / TW;
        const int tx = rem - ty * TW;
        if (tn >= TN) { tn = 0; ty = 0; }   // out-of-tile lanes read pixel row 0 of image 0 (masked on store)
        xbase[np] = Xl + (tn * PHW + ty * p.S * PW_ + tx * p.S) * XP + half;
    }
    const float* wa = Wl + wo * MO * 32 + l31 + half * NTAPS * BO;

    f32x16 acc[NPH][MO][NP];
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
        for (int mo = 0; mo < MO; ++mo)
#pragma unroll
            for (int np = 0; np < NP; ++np)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ph][mo][np][r] = 0.f;

    const float* wbase = p.wt + (long)(p.wgroups > 1 ? (n0 % p.wgroups) : 0) * p.wstride;

    // ---- staging registers and their per-lane constant offsets.  Staging is cut into small "pieces"
    // (one weight float4, or four channels of one patch element) that are slotted between the MFMA
    // groups of the running chunk: the MFMA pipe never waits for a monolithic copy phase.
    // Loads are unconditional (addresses clamped into valid memory; the weight buffer is zero padded to a
    // multiple of 32 input channels by shg_conv_weight_prep_f32) and nothing is computed on loaded values
    // until they are written to LDS, so they stay in flight for a whole chunk.
    f32x4 wv[PER];
    float xv[XQ][KC];
    unsigned wl[PER], xo[XQ];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int e = min(tid + k * NT, V4 - 1);
        const int row = e / (BO / 4), c4 = e - row * (BO / 4);
        const int o = min(o0 + c4 * 4, p.OP - 4);
        wl[k] = (unsigned)(((o >> 6) * p.IPK + row) * 64 + (o & 63));
    }
#pragma unroll
    for (int k = 0; k < XQ; ++k) xo[k] = xoff[k] >= 0 ? (unsigned)xoff[k] : 0u;
    constexpr int XG = KC / 4;                       // 4-channel groups per patch element
    // Double-buffered variants move the weights global -> LDS directly (LDS-DMA, global_load_lds_dwordx4: the LDS
    // image [ROWS][BO] is lane-linear, 1 KiB per wave instruction), so they need no staging registers or ds_writes;
    // the single-buffer variants stage them through registers like the patch.
    constexpr bool WDMA = DB;
    constexpr int PW0 = WDMA ? 0 : PER;              // weight pieces in the register-staged list
    constexpr int NPIECE = PW0 + XQ * XG;

    auto piece_load = [&](int j, int i0) __attribute__((always_inline)) {
        if (j < PW0) {
            if (!(p.dbg & 1)) wv[j] = *reinterpret_cast<const f32x4*>(wbase + (size_t)i0 * NTAPS * 64 + wl[j]);
        } else {
            const int k = (j - PW0) / XG, g = (j - PW0) % XG;
            if (!(p.dbg & 2)) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int ic = g * 4 + c;
                    xv[k][ic] = p.x[xo[k] + (unsigned)(min(i0 + ic, p.I - 1) * HW)];
                }
            }
        }
    };
    auto piece_store = [&](int j, int i0, int buf) __attribute__((always_inline)) {
        if (p.dbg & 4) return;
        if (j < PW0) {
            const int e = tid + j * NT;
            if (e < V4) *reinterpret_cast<f32x4*>(Wl + buf * WSZ + e * 4) = wv[j];
        } else {
            const int k = (j - PW0) / XG, g = (j - PW0) % XG;
            const int q = tid + k * NT;
            f32x4 sc = {1.f, 1.f, 1.f, 1.f};
            if (p.in_scale) sc = *reinterpret_cast<const f32x4*>(Sl + xsn[k] + (i0 - i_begin) + g * 4);
            if (q < PATCH) {
                float* dst = Xl + buf * XSZ + q * XP + g * 4;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    dst[c] = (xoff[k] >= 0 && i0 + g * 4 + c < i_end) ? xv[k][g * 4 + c] * sc[c] : 0.f;
            }
        }
    };

    // weights of the chunk starting at channel i0 -> LDS buffer `buf` (whole waves: V4 is a multiple of 64)
    auto w_dma = [&](int i0, int buf) __attribute__((always_inline)) {
        if (p.dbg & 1) return;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if ((k + 1) * NT <= V4 || k * NT + wave * 64 < V4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + (size_t)i0 * NTAPS * 64 + wl[k]),
                                                 (__attribute__((address_space(3))) void*)(Wl + buf * WSZ + (k * NT + wave * 64) * 4), 16, 0, 0);
        }
    };
    if (!p.fix) {
        if (WDMA) w_dma(i_begin, 0);
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) piece_load(j, i_begin);
        __syncthreads();                 // scale table visible
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) piece_store(j, i_begin, 0);
        if (DB && i_begin + KC < i_end) {
#pragma unroll
            for (int j = 0; j < NPIECE; ++j) piece_load(j, i_begin + KC);
        }
        __syncthreads();
    }

    int cur = 0;
    for (int i0 = i_begin; i0 < i_end; i0 += KC, cur ^= (DB ? 1 : 0)) {
        const bool more = i0 + KC < i_end;          // DB: registers hold chunk i0+KC (loads issued one chunk ago)
        const bool more2 = i0 + 2 * KC < i_end;
        if (WDMA && more) w_dma(i0 + KC, cur ^ 1);   // the other buffer is free since the barrier that ended chunk i0-KC
        if (!DB && more) {                           // single buffer: prefetch the next chunk into registers now
#pragma unroll
            for (int j = 0; j < NPIECE; ++j) piece_load(j, i0 + KC);
        }
        const float* wa_c = wa + cur * WSZ;
        const float* xb_c[NP];
#pragma unroll
        for (int np = 0; np < NP; ++np) xb_c[np] = xbase[np] + cur * XSZ;
        // hand-over: park the prefetched chunk i0+KC in the other LDS buffer, then reuse the staging registers
        // for chunk i0+2KC.  It is a VALU/LDS/VMEM-heavy block during which this wave issues no MFMA; with
        // two waves per SIMD (8-wave workgroups) the two wave groups do it at different points of the chunk,
        // so the partner wave's MFMAs keep the pipe busy meanwhile.
        auto handover = [&]() __attribute__((always_inline)) {
            if (more) {
#pragma unroll
                for (int j = 0; j < NPIECE; ++j) piece_store(j, i0 + KC, cur ^ 1);
            }
            if (more2) {
#pragma unroll
                for (int j = 0; j < NPIECE; ++j) piece_load(j, i0 + 2 * KC);
            }
        };
        // waves w, w+4, w+8, ... share a SIMD: NGRP wave groups hand over at evenly spread points of the chunk
        constexpr int NGRP = (DB && WO * WP >= 8) ? (WO * WP) / 4 : 1;
        const int grp = wave >> 2;

        if constexpr (!UP) {
            // operands of step s+1 are read from LDS while the MFMAs of step s execute (explicit register
            // double buffering: a single wave per SIMD then never stalls on LDS latency)
            constexpr int NSTEP = NTAPS * (KC / 2);
            float a[2][MO], b[2][NP];
            auto fetch = [&](int step, int buf) __attribute__((always_inline)) {
                const int t = step / (KC / 2), c2 = step % (KC / 2);
                const int toff = p.tap_off[t] * XP;
#pragma unroll
                for (int mo = 0; mo < MO; ++mo) a[buf][mo] = wa_c[((c2 * 2) * NTAPS + t) * BO + mo * 32];
#pragma unroll
                for (int np = 0; np < NP; ++np) b[buf][np] = xb_c[np][toff + c2 * 2];
            };
            fetch(0, 0);
            auto step_body = [&](int step) __attribute__((always_inline)) {
                if (step + 1 < NSTEP) fetch(step + 1, (step + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);   // keep the next step's LDS reads ahead of this step's MFMAs
#pragma unroll
                for (int mo = 0; mo < MO; ++mo)
#pragma unroll
                    for (int np = 0; np < NP; ++np)
                        acc[0][mo][np] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[step & 1][mo], b[step & 1][np], acc[0][mo][np], 0, 0, 0);
            };
            // one straight-line stream; the hand-over blocks are emitted right after the MFMAs of the chosen steps
            // group g hands over after step ((2g+1) * NSTEP) / (2 * NGRP)
#pragma unroll
            for (int step = 0; step < NSTEP; ++step) {
                step_body(step);
                if (DB) {
                    constexpr int QA = NSTEP / (2 * NGRP), QB = (3 * NSTEP) / (2 * NGRP), QC = (5 * NSTEP) / (2 * NGRP),
                                  QD = (7 * NSTEP) / (2 * NGRP);
                    if (step == QA || (NGRP >= 2 && step == QB) || (NGRP >= 4 && (step == QC || step == QD))) {
                        const int g = step == QA ? 0 : (step == QB ? 1 : (step == QC ? 2 : 3));
                        __builtin_amdgcn_sched_barrier(0);
                        if (NGRP == 1 || grp == g) handover();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        } else {
            // tap t = ky*3+kx feeds phase (ky&1, kx&1) from the input pixel shifted by (ky==2 ? -1 : 0, kx==2 ? -1 : 0);
            // the patch origin is (u0-1, v0-1), so shift (sy,sx) sits at patch offset (1+sy)*PW + (1+sx)
            const int sh[4] = {(PW_ + 1) * XP, PW_ * XP, XP, 0};     // (0,0) (0,-1) (-1,0) (-1,-1)
            constexpr int PHASE[9] = {0, 1, 0, 2, 3, 2, 0, 1, 0};
            constexpr int SHIFT[9] = {0, 0, 1, 0, 0, 1, 2, 2, 3};
            // A slot = three taps (6 or 3*MO MFMAs): weight operands of slot s+1 are read while slot s multiplies, so
            // LDS latency has >= 384 MFMA cycles of cover.
            constexpr int NSLOT = 3 * (KC / 2);
            float a[2][3][MO], b[2][4][NP];
            auto fetch_a = [&](int slot, int buf) __attribute__((always_inline)) {
                const int c2 = slot / 3, g = slot % 3;
#pragma unroll
                for (int tt = 0; tt < 3; ++tt)
#pragma unroll
                    for (int mo = 0; mo < MO; ++mo) a[buf][tt][mo] = wa_c[((c2 * 2) * 9 + g * 3 + tt) * BO + mo * 32];
            };
            auto fetch_b = [&](int c2, int buf) __attribute__((always_inline)) {
#pragma unroll
                for (int np = 0; np < NP; ++np)
#pragma unroll
                    for (int q = 0; q < 4; ++q) b[buf][q][np] = xb_c[np][sh[q] + c2 * 2];
            };
            fetch_b(0, 0);
            fetch_a(0, 0);
            auto slot_body = [&](int slot) __attribute__((always_inline)) {
                const int c2 = slot / 3, g = slot % 3;
                if (slot + 1 < NSLOT) fetch_a(slot + 1, (slot + 1) & 1);
                if (g == 0 && c2 + 1 < KC / 2) fetch_b(c2 + 1, (c2 + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tt = 0; tt < 3; ++tt) {
                    const int t = g * 3 + tt;
#pragma unroll
                    for (int mo = 0; mo < MO; ++mo)
#pragma unroll
                        for (int np = 0; np < NP; ++np)
                            acc[PHASE[t]][mo][np] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                a[slot & 1][tt][mo], b[c2 & 1][SHIFT[t]][np], acc[PHASE[t]][mo][np], 0, 0, 0);
                }
            };
            static_for<NSLOT>([&](auto sc) __attribute__((always_inline)) {
                constexpr int slot = decltype(sc)::value;
                slot_body(slot);
                if constexpr (DB) {
                    static_for<NGRP>([&](auto gc) __attribute__((always_inline)) {
                        constexpr int g = decltype(gc)::value;
                        if constexpr (slot == ((2 * g + 1) * NSLOT) / (2 * NGRP)) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (NGRP == 1 || grp == g) handover();
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                }
            });
        }
        if (!(p.dbg & 8)) __syncthreads();
        if (!DB && more) {                           // single buffer: everyone is done reading -> overwrite, publish
#pragma unroll
            for (int j = 0; j < NPIECE; ++j) piece_store(j, i0 + KC, 0);
            __syncthreads();
        }
    }
    if (p.dbg & 16) return;

    // tail split: slice blocks park their raw accumulators in tail_ws [tile][slice][reg][thread]; the fix launch reads
    // them back inside the epilogue loops below (same code path as an ordinary tile from there on)
    constexpr int NREG = NPH * MO * NP * 16;
    float* const tws = (tslice >= 0 || p.fix)
        ? p.tail_ws + ((size_t)(work - p.tail_first) * p.tail_ks + (tslice >= 0 ? tslice : 0)) * NREG * NT + tid : nullptr;

    // ---- epilogue: D[row = out channel][col = pixel]; row = (r&3) + 8*(r>>2) + 4*half
#pragma unroll
    for (int np = 0; np < NP; ++np) {
        const int j = (wp * NP + np) * 32 + l31;
        const int tn = j / THW;
        const int rem = j - tn * THW;
        const int ty = rem / TW, tx = rem - ty * TW;
        const int n = n0 + tn, oy = oy0 + ty, ox = ox0 + tx;
        if (tn >= TN || n >= p.NB || oy >= oy_end || ox >= ox_end) continue;
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            long plane = (long)p.OHt * p.OWt;    // elements per (n,o) plane of the output
            long base;                            // offset of (n, o=0, pixel)
            int pix;
            if constexpr (UP) {
                const int a = ph >> 1, b = ph & 1;
                if (a == 1 && oy >= p.H) continue;                    // Y = 2u+a <= 2H
                if (b == 1 && ox >= p.W) continue;                    // X = 2v+b <= 2W
                if (p.out_mode == 1) {
                    plane = (long)(p.H + 1) * (p.W + 1);
                    pix = oy * (p.W + 1) + ox;
                    base = ((long)ph * p.NB + n) * p.O * plane + pix;
                } else {
                    pix = (2 * oy + a) * p.OWt + 2 * ox + b;
                    base = (long)n * p.O * plane + pix;
                }
            } else {
                pix = oy * p.OWt + ox;
                base = (long)n * p.O * plane + pix;
            }
            float nz = 0.f;
            if (p.ksplit == 1) {
                if (p.noise_mode == 1) nz = p.noise[pix] * p.noise_strength;
                else if (p.noise_mode == 2) nz = p.noise[(long)n * plane + pix] * p.noise_strength;
            }
#pragma unroll
            for (int mo = 0; mo < MO; ++mo) {
                const int ob = o0 + (wo * MO + mo) * 32 + 4 * half;        // channel of register r: ob + (r&3) + 8*(r>>2)
                if (p.ksplit > 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int o = ob + (r & 3) + 8 * (r >> 2);
                        if (o < p.O) p.part[(long)kslice * p.part_stride + base + (long)o * plane] = acc[ph][mo][np][r];
                    }
                    continue;
                }
                float* const tw = tws + (size_t)(((ph * MO + mo) * NP + np) * 16) * NT;
                if (tslice >= 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) tw[r * NT] = acc[ph][mo][np][r];
                    continue;
                }
                if (p.fix) {
                    for (int sl = 0; sl < p.tail_ks; ++sl)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[ph][mo][np][r] += tw[(size_t)sl * NREG * NT + r * NT];
                }
                if constexpr (UP && OCC == 4) {
                    // the 16-wave transposed variant is dispatched for the planar hand-over to the FIR kernel only
                    // (no fused tail here: it lives in fir_up_planar): plain stores, no operand gather
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int o = ob + (r & 3) + 8 * (r >> 2);
                        if (o < p.O) p.y[base + (long)o * plane] = acc[ph][mo][np][r] * p.gain;
                    }
                    continue;
                }
                // gather the per-channel operands first (independent loads, one wait), then compute and store
                float osc[16], bs[16], rs[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = min(ob + (r & 3) + 8 * (r >> 2), p.O - 1);
                    osc[r] = p.out_scale ? p.out_scale[n * p.O + o] : 1.f;
                    bs[r] = p.bias ? p.bias[o] : 0.f;
                    rs[r] = p.residual ? p.residual[base + (long)o * plane] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = ob + (r & 3) + 8 * (r >> 2);
                    if (o >= p.O) continue;
                    float v = acc[ph][mo][np][r] * osc[r] + nz + bs[r];
                    v = p.act ? shg_lrelu_agc(v, p.alpha, p.gain, p.clamp) : v * p.gain;
                    p.y[base + (long)o * plane] = v + rs[r];
                }
            }
        }
    }
}

// split-K tail: y = epilogue(sum_s part[s]) over an [NB, O, plane] tensor
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvParams p, long total, long plane) {
    const long stride = (long)gridDim.x * 256;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
        float v = 0.f;
        for (int s = 0; s < p.ksplit; ++s) v += p.part[(long)s * p.part_stride + e];
        if (p.raw_reduce) { p.y[e] = v; continue; }
        const long no = e / plane;
        const int pix = (int)(e - no * plane);
        const int n = (int)(no / p.O), o = (int)(no - (long)n * p.O);
        float nz = 0.f;
        if (p.noise_mode == 1) nz = p.noise[pix] * p.noise_strength;
        else if (p.noise_mode == 2) nz = p.noise[(long)n * plane + pix] * p.noise_strength;
        p.y[e] = conv_epilogue(p, v, n, o, e, nz);
    }
}

// ------------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------------


// Tile shape for one rectangular region of `rows` x `cols` output pixels of NB images.  The BP pixels of a tile are
// TW x TH pixels of TN images; any integers with TW*TH*TN <= BP are allowed, so odd extents do not pay the 2x padding
// waste of power-of-two tiles.  Chosen: the shape with the fewest tiles whose input patch fits the staging capacity;
// ties go to rows that are multiples of 32 pixels (full 128-byte store segments), then to the smallest patch.
static void tile_search(ConvParams::TileClass& t, int NB, int rows, int cols, int BP, int wgroups, int S, int span_y, int span_x,
                        int patch_cap) {
    long best = -1, bpatch = 0;
    int btw = 1, bth = 1, btn = 1, bfull = 0;
    const int tw_max = cols < BP ? cols : BP;
    for (int tw = 1; tw <= tw_max; ++tw) {
        if (tw < 8 && tw != cols && cols >= 8) continue;         // keep rows reasonably long
        int th = BP / tw; if (th > rows) th = rows;
        const long pw = (tw - 1) * S + span_x;
        while (th > 1 && ((th - 1) * S + span_y) * pw > patch_cap) --th;   // shrink to the staging capacity
        int tn = BP / (tw * th); if (tn > NB) tn = NB;
        if (wgroups > 1) tn = 1;                                 // per-slot weights: one image per tile
        if (tn < 1) tn = 1;
        while (tn > 1 && (long)tn * ((th - 1) * S + span_y) * pw > patch_cap) --tn;
        const long patch = (long)tn * ((th - 1) * S + span_y) * pw;
        if (patch > patch_cap) continue;
        const long tiles = (long)shg_cdiv(cols, tw) * shg_cdiv(rows, th) * shg_cdiv(NB, tn);
        const int full = (tw % 32 == 0) ? 1 : 0;
        if (best < 0 || tiles < best || (tiles == best && (full > bfull || (full == bfull && patch < bpatch)))) {
            best = tiles; bpatch = patch; btw = tw; bth = th; btn = tn; bfull = full;
        }
    }
    if (best < 0) { btw = cols < 32 ? cols : 32; bth = 1; btn = 1; }   // nothing fits: one row segment
    t.TW = btw; t.TH = bth; t.TN = btn;
    t.tiles_x = shg_cdiv(cols, btw); t.tiles_y = shg_cdiv(rows, bth);
    t.n_tiles = t.tiles_x * t.tiles_y * shg_cdiv(NB, btn);
    t.rows = rows; t.cols = cols;
    t.PH = (bth - 1) * S + span_y; t.PW = (btw - 1) * S + span_x;
    t.PATCH = btn * t.PH * t.PW;
}

// Fills p.cls / n_ptiles / n_otiles / patch_max / tn_max.  up: the (H+1) x (W+1) grid of the transposed conv is split
// into the H x W body and the two one-pixel strips.
static void conv_tiles(ConvParams& p, int BO, int BP, bool up, int patch_cap) {
    for (int i = 0; i < 3; ++i) p.cls[i] = ConvParams::TileClass{};
    if (!up) {
        tile_search(p.cls[0], p.NB, p.OHp, p.OWp, BP, p.wgroups, p.S, p.span_y, p.span_x, patch_cap);
    } else {
        tile_search(p.cls[0], p.NB, p.H, p.W, BP, p.wgroups, 1, 2, 2, patch_cap);            // u < H, v < W
        tile_search(p.cls[1], p.NB, 1, p.W + 1, BP, p.wgroups, 1, 2, 2, patch_cap);          // row u = H
        p.cls[1].oy_base = p.H;
        tile_search(p.cls[2], p.NB, p.H, 1, BP, p.wgroups, 1, 2, 2, patch_cap);              // column v = W
        p.cls[2].ox_base = p.W;
    }
    p.n_ptiles = p.cls[0].n_tiles + p.cls[1].n_tiles + p.cls[2].n_tiles;
    p.n_otiles = shg_cdiv(p.O, BO);
    p.patch_max = 0; p.tn_max = 1;
    for (int i = 0; i < 3; ++i) {
        if (p.cls[i].n_tiles == 0) continue;
        if (p.cls[i].PATCH > p.patch_max) p.patch_max = p.cls[i].PATCH;
        if (p.cls[i].TN > p.tn_max) p.tn_max = p.cls[i].TN;
    }
}

// Split-K factor for a grid of `grid` workgroups over `chunks` K-chunks: aim at >= 2 workgroups per CU (512) while
// keeping >= 4 chunks per slice.
static int conv_ksplit(int grid, int chunks, bool allow) {
    int ks = 1;
    if (allow) while (grid * ks < 512 && chunks / (ks * 2) >= 4 && ks < 64) ks *= 2;
    return ks;
}

// smallest phase-grid width (W+1) that goes to the 8-wave double-buffered transposed-conv tiles (SHG_UP_MIN)
static int conv_up_min() {
#ifdef SHG_ABLATE
    static const int v = getenv("SHG_UP_MIN") ? atoi(getenv("SHG_UP_MIN")) : 16;
    return v;
#else
    return 16;
#endif
}

static int conv_cu_count() { return shg_cu_count(); }

// Upper bound of the tail-split workspace of the one-workgroup-per-CU kernels: < CU-count tile-slices of at most
// 128 x 128 x 4 (transposed) or 128 x 256 accumulators.
static size_t conv_tail_ws_bound(bool up) { return (size_t)256 * (up ? 128 * 128 * 4 : 128 * 256) * sizeof(float); }

template <int NTAPS, int KC, int MO, int NP, int WO, int WP, int XQ, bool UP, bool DB, int OCC>
static int launch_conv(ConvParams& p, void* workspace, size_t ws_bytes, hipStream_t s) {
    constexpr int BO = MO * 32 * WO, BP = NP * 32 * WP, NT = WO * WP * 64;
    const long out_elems = (UP && p.out_mode == 1) ? 4L * p.NB * p.O * (p.H + 1) * (p.W + 1) : (long)p.NB * p.O * p.OHt * p.OWt;
    conv_tiles(p, BO, BP, UP, XQ * NT);
    const int chunks = shg_cdiv(p.I, KC);
    int ks = conv_ksplit(p.n_ptiles * p.n_otiles, chunks, workspace != nullptr);
    if (DB && p.n_ptiles * p.n_otiles >= conv_cu_count()) ks = 1;              // one workgroup per CU and a full round already: the tail split below handles the rest
    while (ks > 1 && (size_t)ks * out_elems * sizeof(float) > ws_bytes) ks /= 2;  // workspace too small: the widest split that fits
    p.i_per_slice = shg_cdiv(chunks, ks) * KC;
    p.ksplit = shg_cdiv(p.I, p.i_per_slice);
    p.raw_reduce = UP ? 1 : 0;
    p.part = p.ksplit > 1 ? (float*)workspace : nullptr;
    p.part_stride = out_elems;
    if (p.patch_max > XQ * NT) { shg_set_error("conv: patch of %d elements exceeds the staging capacity %d", p.patch_max, XQ * NT); return SHG_ERR_UNSUPPORTED; }
    if (!UP)
        for (int t = 0; t < NTAPS; ++t) {
            const int dyr = p.tap_off[t] >> 6, dxr = p.tap_off[t] & 63;   // packed (dy-dy0, dx-dx0)
            p.tap_off[t] = dyr * p.cls[0].PW + dxr;
        }
    const size_t lds = sizeof(float) * ((DB ? 2 : 1) * ((size_t)KC * NTAPS * BO + (size_t)(KC + 1) * p.patch_max) +
                                        (p.in_scale ? (size_t)p.tn_max * p.i_per_slice : 0));
    auto kern = conv_mfma_kernel<NTAPS, KC, MO, NP, WO, WP, XQ, UP, DB, OCC>;
    if (lds > 64 * 1024) {
        if (lds > 160 * 1024) { shg_set_error("conv: LDS request %zu exceeds 160 KiB", lds); return SHG_ERR_UNSUPPORTED; }
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { shg_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return SHG_ERR_LAUNCH; }
    }
    // tail split: one-workgroup-per-CU kernels only (the double-buffered variants), when the last round is < 60 % full
    const int nwork = p.n_ptiles * p.n_otiles;
    p.tail_first = nwork; p.tail_ks = 1; p.tail_ips = p.I; p.fix = 0; p.tail_ws = nullptr;
    int n_tail = 0;
    if (DB && p.ksplit == 1 && workspace && !(p.dbg & 32)) {
        const int slots = conv_cu_count();
        const int tail = nwork % slots;
        if (nwork > slots && tail > 0 && tail * 10 < slots * 6) {
            int tk = 1;
            while (tk < 8 && tail * tk * 2 <= slots && chunks / (tk * 2) >= 4) tk *= 2;
            const size_t need = (size_t)tail * tk * (BO * BP * (UP ? 4 : 1)) * sizeof(float);
            if (tk > 1 && need <= ws_bytes) {
                n_tail = tail; p.tail_first = nwork - tail; p.tail_ks = tk; p.tail_ips = shg_cdiv(chunks, tk) * KC;
                p.tail_ws = (float*)workspace;
            }
        }
    }
    const int grid = p.ksplit > 1 ? nwork * p.ksplit : p.tail_first + n_tail * p.tail_ks;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, s, p);
    SHG_CHECK_LAUNCH();
    if (n_tail) {
        p.fix = 1;
        hipLaunchKernelGGL(kern, dim3(n_tail), dim3(NT), lds, s, p);
        SHG_CHECK_LAUNCH();
        p.fix = 0;
    }
    if (p.ksplit > 1) {
        const long plane = (long)p.OHt * p.OWt;
        int rg = shg_cdiv(out_elems, 256); if (rg > 2048) rg = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rg), dim3(256), 0, s, p, out_elems, plane);
        SHG_CHECK_LAUNCH();
    }
    return SHG_OK;
}

// ---------------------------------------------------------------------------------------------
// 1x1 convolutions as a plain GEMM (round 6): y[n, o, P] = epilogue(sum_i W[o, i] x[n, i, P]) for the critic's skip branches and their
// input gradients (stylegan.py:413-421 via conv2d_resample.py:108-113: FIR-decimate, then a 1x1 layer).  The tap-list kernel above served
// them with register-staged 4-byte patch loads into a pixel-major LDS image at 2 - 2.6 x the larger of their HBM and MFMA floors
// (tools/conv1x1_bench.py).  Here both operands arrive by 16-byte LDS-DMA in the layouts the MFMA reads: the weight chunk [64-block][k][64 o]
// (the prepped tensor's own order: 1 KiB = four rows per wave instruction) and the activations [k][pixel] (a channel's pixels are contiguous
// in NCHW: 1 KiB = 256 pixels of one channel or 128 of two) -- no staging registers, no LDS writes, two stages, one barrier per chunk, and
// the operands of k-step j + 1 are read before the MFMAs of step j issue.  Workgroup = 4 waves of 64 x 64 outputs: 128 o x 128 px, or
// 64 o x 256 px for 64 output channels.  Eligible: whole tiles (O % BO, pixels % BP, I % KC == 0), no style / noise operands.
// ---------------------------------------------------------------------------------------------
template <int WO, int WP, int KC>
__global__ __launch_bounds__(256, 2) void conv1x1_gemm_kernel(const ConvParams p) {
    constexpr int BO = 64 * WO, BP = 64 * WP, WSZ = KC * BO, XSZ = KC * BP;
    static_assert(WO * WP == 4 && KC % 4 == 0, "four waves");
    extern __shared__ __attribute__((aligned(16))) float c1_lds[];
    float* Ws = c1_lds;                       // [2][WO][KC][64]
    float* Xs = c1_lds + 2 * WSZ;             // [2][KC][BP]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wo = wave / WP, wp = wave % WP;
    const int P = p.H * p.W, ptiles = P / BP, otiles = p.O / BO;
    int t = xcd_remap(blockIdx.x, gridDim.x);
    const int ot = t % otiles; t /= otiles;
    const int pt = t % ptiles, n = t / ptiles;
    const int o0 = ot * BO, p0 = pt * BP;
    const float* xn = p.x + ((long)n * p.I) * P + p0;
    const float* wb = p.wt + (long)(o0 >> 6) * p.IPK * 64;
    // DMA requests of a chunk: weights 4 rows (of one 64-block) per request, activations 256 pixels per request
    constexpr int WREQ = WO * KC / 4, XREQ = KC * BP / 256, RPW = (WREQ + XREQ) / 4;
    static_assert((WREQ + XREQ) % 4 == 0, "requests divide over the four waves");
    auto issue = [&](int i0, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const int rq = wave + 4 * k;                                  // wave-uniform
            if (rq < WREQ) {
                const int b = rq / (KC / 4), r4 = rq % (KC / 4);
                const float* src = wb + ((long)b * p.IPK + i0 + r4 * 4 + (lane >> 4)) * 64 + (lane & 15) * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(Ws + buf * WSZ + (b * KC + r4 * 4) * 64), 16, 0, 0);
            } else {
                const int xr = rq - WREQ;                                 // 256 pixels: BP = 128 -> rows 2 xr, 2 xr + 1; BP = 256 -> row xr
                const int row = BP == 128 ? 2 * xr + (lane >> 5) : xr, px = BP == 128 ? (lane & 31) * 4 : lane * 4;
                const float* src = xn + (long)(i0 + row) * P + px;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(Xs + buf * XSZ + xr * 256), 16, 0, 0);
            }
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    issue(0, 0);
    int buf = 0;
    for (int i0 = 0; i0 < p.I; i0 += KC, buf ^= 1) {
        __syncthreads();                                                   // chunk i0 landed (vmcnt(0) before the barrier); the other stage is free
        if (i0 + KC < p.I) issue(i0 + KC, buf ^ 1);
        const float* wa = Ws + buf * WSZ + (wo * KC + half) * 64 + l31;
        const float* xa = Xs + buf * XSZ + half * BP + wp * 64 + l31;
        float a_cur[2] = {wa[0], wa[32]}, b_cur[2] = {xa[0], xa[32]};
#pragma unroll
        for (int j = 0; j < KC / 2; ++j) {
            float a_nxt[2] = {0.f, 0.f}, b_nxt[2] = {0.f, 0.f};
            if (j + 1 < KC / 2) {
                a_nxt[0] = wa[(2 * j + 2) * 64]; a_nxt[1] = wa[(2 * j + 2) * 64 + 32];
                b_nxt[0] = xa[(2 * j + 2) * BP]; b_nxt[1] = xa[(2 * j + 2) * BP + 32];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[a], b_cur[b], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a_cur[0] = a_nxt[0]; a_cur[1] = a_nxt[1]; b_cur[0] = b_nxt[0]; b_cur[1] = b_nxt[1];
        }
    }
    // epilogue: register r of tile (a, b) = channel o0 + wo 64 + a 32 + (r & 3) + 8 (r >> 2) + 4 half, pixel p0 + wp 64 + b 32 + l31.  Stored from
    // the registers a store instruction would write 2 x 128 bytes of 2 channels; the wave's 64 x 64 block therefore goes through LDS (its own
    // 16 KiB of the stages, 16 channels at a time in the padded form [16][68]) and leaves as 16-byte pieces: 256 contiguous bytes per channel row.
    __syncthreads();                                                       // every wave is done reading the stages
    float* tr = c1_lds + wave * (16 * 68);
    const ShgAct act = shg_act_make(p.act, p.alpha, p.gain, p.clamp);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {                                   // 16 channels: registers r = 8 hb .. 8 hb + 7 of both pixel blocks
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = 8 * hb + rr, cl = (rr & 3) + 8 * (rr >> 2) + 4 * half;      // channel within the 16: (r & 3) + 8 ((r >> 2) & 1) + 4 half
                    tr[cl * 68 + b * 32 + l31] = acc[a][b][r];
                }
            // (a wave's own region: no barrier, the LDS pipe returns in order)
#pragma unroll
            for (int k = 0; k < 4; ++k) {                                  // 16 channels x 16 pieces = 256 pieces over 64 lanes
                const int e = lane + 64 * k, cl = e >> 4, pc = e & 15;
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(tr + cl * 68 + pc * 4);
                const int o = o0 + wo * 64 + a * 32 + 16 * hb + cl;
                const long idx = ((long)n * p.O + o) * P + p0 + wp * 64 + pc * 4;
                const float osc = p.out_scale ? p.out_scale[n * p.O + o] : 1.f, bs = p.bias ? p.bias[o] : 0.f;
                f32x4 rs = {0.f, 0.f, 0.f, 0.f};
                if (p.residual) rs = *reinterpret_cast<const f32x4*>(p.residual + idx);
                f32x4 out;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = v4[q] * osc + bs;
                    v = p.act ? shg_lrelu_agc(v, p.alpha, p.gain, p.clamp) : v * p.gain;   // (same expression as conv_epilogue: identical bits)
                    out[q] = v + rs[q];
                }
                *reinterpret_cast<f32x4*>(p.y + idx) = out;
            }
        }
    (void)act;
}

// the GEMM form applies: stride-1 1x1 without style / noise operands, whole tiles, enough of them to fill the chip
static bool conv1x1_gemm_ok(const ConvParams& p, int* narrow) {
    const long P = (long)p.H * p.W;
    if (p.wgroups != 1 || p.in_scale || p.noise_mode || p.OHt != p.H || p.OWt != p.W) return false;
    if ((reinterpret_cast<uintptr_t>(p.x) | reinterpret_cast<uintptr_t>(p.wt) | reinterpret_cast<uintptr_t>(p.y) | reinterpret_cast<uintptr_t>(p.residual)) & 15) return false;   // 16-byte DMA / vector accesses
    const bool nar = p.O % 128 != 0;
    *narrow = nar;
    const int BO = nar ? 64 : 128, BP = nar ? 256 : 128, KC = 16;
    if (p.O % BO || P % BP || p.I % KC) return false;
    return (long)p.NB * (P / BP) * (p.O / BO) >= conv_cu_count();          // (at least one tile per CU: below that the split-K tap-list kernel fills the chip better)
}

static int launch_conv1x1_gemm(ConvParams& p, int narrow, hipStream_t s) {
    const long P = (long)p.H * p.W;
    if (narrow) {
        constexpr int lds = 2 * (16 * 64 + 16 * 256) * 4;
        hipLaunchKernelGGL((conv1x1_gemm_kernel<1, 4, 16>), dim3((unsigned)(p.NB * (P / 256) * (p.O / 64))), dim3(256), lds, s, p);
    } else {
        constexpr int lds = 2 * (16 * 128 + 16 * 128) * 4;            // 32 KiB: four workgroups per CU
        hipLaunchKernelGGL((conv1x1_gemm_kernel<2, 2, 16>), dim3((unsigned)(p.NB * (P / 128) * (p.O / 128))), dim3(256), lds, s, p);
    }
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

static int conv_dispatch(ConvParams& p, int K, int S, bool up, void* workspace, size_t ws_bytes, hipStream_t s) {
    const bool narrow = p.O <= 64;    // 64 x 256 tile instead of 128 x 128
    // SHG_CONV_VARIANT (tuning knob, bit flags): 1 = force the single-buffer 4-wave kernels; 2 / 4 / 8 = try the 8-wave
    // double-buffered variants for 64-channel layers / stride-2 (128 px) / stride-2 (256 px); 16 = transposed conv back on 8 waves;
    // 32 = stride-2 back on 8 waves
#ifdef SHG_ABLATE
    static const int variant = getenv("SHG_CONV_VARIANT") ? atoi(getenv("SHG_CONV_VARIANT")) : 0;
#else
    constexpr int variant = 0;
#endif
    if (up) {
        // all-phase transposed conv: large grids use 8-wave double-buffered tiles (128 ch x 128 px, or 64 ch x 256 px),
        // small ones the 4-wave 64 ch x 128 px tile (+ split-K)
        const bool big = !(variant & 1) && p.OWp >= conv_up_min() && p.OHp >= 16 && p.wgroups == 1;
        const bool raw_out = p.out_mode == 1 && !p.out_scale && !p.bias && !p.noise && !p.residual && !p.act;
        // planar hand-over to the FIR kernel: 16 waves (four instruction streams per SIMD), 128 ch x 128 px or 64 ch x 256 px
        if (big && raw_out && !(variant & 16))
            return narrow ? launch_conv<9, 8, 1, 1, 2, 8, 1, true, true, 4>(p, workspace, ws_bytes, s)
                          : launch_conv<9, 8, 1, 1, 4, 4, 1, true, true, 4>(p, workspace, ws_bytes, s);
        if (big) return narrow ? launch_conv<9, 8, 2, 1, 1, 8, 1, true, true, 2>(p, workspace, ws_bytes, s)
                               : launch_conv<9, 8, 2, 1, 2, 4, 1, true, true, 2>(p, workspace, ws_bytes, s);
        return launch_conv<9, 8, 2, 1, 1, 4, 2, true, false, 2>(p, workspace, ws_bytes, s);
    }
    if (K == 9 && S == 1) {
        // large images, many channels: 8-wave 128 x 256 tile, double-buffered LDS, staggered hand-over
        if (!narrow && !(variant & 1) && p.OWp >= 32 && p.OHp >= 8 && p.wgroups == 1)
            return launch_conv<9, 8, 2, 2, 2, 4, 1, false, true, 2>(p, workspace, ws_bytes, s);
        if (narrow && (variant & 2) && p.OWp >= 32 && p.OHp >= 16 && p.wgroups == 1)
            return launch_conv<9, 8, 2, 2, 1, 8, 2, false, true, 2>(p, workspace, ws_bytes, s);
        if (narrow && (variant & 8) && p.OWp >= 32 && p.OHp >= 16 && p.wgroups == 1)
            return launch_conv<9, 8, 2, 2, 1, 4, 2, false, true, 2>(p, workspace, ws_bytes, s);
        return narrow ? launch_conv<9, 8, 2, 2, 1, 4, 2, false, false, 3>(p, workspace, ws_bytes, s)
                      : launch_conv<9, 8, 2, 2, 2, 2, 2, false, false, 2>(p, workspace, ws_bytes, s);
    }
    if (K == 9 && S == 2) {
        if (!narrow && (variant & 4) && p.OWp >= 32 && p.OHp >= 4 && p.wgroups == 1)
            return launch_conv<9, 8, 2, 1, 2, 4, 2, false, true, 2>(p, workspace, ws_bytes, s);
        // 128 ch x 256 px tile, double-buffered: 16 waves (4 instruction streams per SIMD; measured 3 % faster than 8 waves)
        if (!narrow && !(variant & 33) && p.OWp >= 32 && p.OHp >= 8 && p.wgroups == 1 && !p.in_scale)
            return launch_conv<9, 8, 1, 2, 4, 4, 2, false, true, 4>(p, workspace, ws_bytes, s);
        if (!narrow && !(variant & 1) && p.OWp >= 32 && p.OHp >= 8 && p.wgroups == 1 && !p.in_scale)
            return launch_conv<9, 8, 2, 2, 2, 4, 3, false, true, 2>(p, workspace, ws_bytes, s);
        return narrow ? launch_conv<9, 8, 2, 2, 1, 4, 5, false, false, 2>(p, workspace, ws_bytes, s)
                      : launch_conv<9, 8, 2, 2, 2, 2, 3, false, false, 2>(p, workspace, ws_bytes, s);
    }
    int c1_narrow = 0;
    if (K == 1 && S == 1 && !(variant & 64) && conv1x1_gemm_ok(p, &c1_narrow)) return launch_conv1x1_gemm(p, c1_narrow, s);
    if (K == 1 && S == 1) return narrow ? launch_conv<1, 32, 2, 2, 1, 4, 1, false, false, 3>(p, workspace, ws_bytes, s)
                                        : launch_conv<1, 32, 2, 2, 2, 2, 1, false, false, 3>(p, workspace, ws_bytes, s);
    shg_set_error("conv2d: 1x1 stride-2 convolution is not implemented (decimate with upfirdn2d first)");
    return SHG_ERR_UNSUPPORTED;
}

static int conv_fill(ConvParams& p, const float* x, const float* wt, float* y, int NB, int I, int O, int OP, int H, int W, int kh,
                     int kw, int mode, int pad, int wgroups, long wstride, const float* in_scale, const float* out_scale,
                     const float* bias, const float* noise, int noise_mode, float noise_strength, int act, float alpha,
                     float gain, float clamp, const float* residual, int out_mode) {
    SHG_CHECK_ARG(x && wt && y, "conv2d: null pointer");
    SHG_CHECK_ARG(NB >= 1 && I >= 1 && O >= 1 && H >= 1 && W >= 1, "conv2d: empty tensor");
    SHG_CHECK_ARG((kh == 3 && kw == 3) || (kh == 1 && kw == 1), "conv2d: only 3x3 and 1x1 kernels (got %dx%d)", kh, kw);
    SHG_CHECK_ARG(OP % 64 == 0 && OP >= O, "conv2d: OP must be a multiple of 64 and >= O");
    SHG_CHECK_ARG(mode >= 0 && mode <= 2, "conv2d: bad mode %d", mode);
    SHG_CHECK_ARG(mode != 2 || (kh == 3 && pad == 0), "conv2d: transposed mode supports 3x3, padding 0");
    SHG_CHECK_ARG(pad >= 0 && pad < 32, "conv2d: bad padding");
    SHG_CHECK_ARG((long)NB * I * H * W < 2147483647L, "conv2d: x is too large");
    p = ConvParams{};
    p.x = x; p.wt = wt; p.y = y; p.in_scale = in_scale; p.out_scale = out_scale; p.bias = bias;
    p.noise = noise_mode ? noise : nullptr; p.residual = residual;
    p.NB = NB; p.I = I; p.O = O; p.OP = OP; p.H = H; p.W = W;
    p.IPK = (I + 31) / 32 * 32 * kh * kw;
    p.wgroups = wgroups < 1 ? 1 : wgroups; p.wstride = wstride;
    p.noise_mode = noise ? noise_mode : 0; p.noise_strength = noise_strength;
    p.act = act; p.alpha = alpha; p.gain = gain; p.clamp = clamp; p.out_mode = out_mode; p.ksplit = 1;
#ifdef SHG_ABLATE
    { const char* d = getenv("SHG_CONV_DBG"); p.dbg = d ? atoi(d) : 0; }
#endif
    if (mode == 2) {
        // transposed stride 2 (conv2d_resample.py:130-137): Y = 2u+a, X = 2v+b over u in [0,H], v in [0,W]
        p.OHt = 2 * H + 1; p.OWt = 2 * W + 1; p.OHp = H + 1; p.OWp = W + 1; p.S = 1;
        p.dy0 = -1; p.dx0 = -1; p.span_y = 2; p.span_x = 2;    // taps reach rows u-1..u, cols v-1..v
        SHG_CHECK_ARG(4L * NB * O * (H + 1) * (W + 1) < 2147483647L, "conv2d: y is too large");
        return SHG_OK;
    }
    const int S = mode == 0 ? 1 : 2;
    const int OH = (H + 2 * pad - kh) / S + 1, OW = (W + 2 * pad - kw) / S + 1;
    SHG_CHECK_ARG(OH >= 1 && OW >= 1, "conv2d: output must be at least 1x1");
    SHG_CHECK_ARG((long)NB * O * OH * OW < 2147483647L, "conv2d: y is too large");
    p.OHp = p.OHt = OH; p.OWp = p.OWt = OW; p.S = S;
    p.dy0 = -pad; p.dx0 = -pad; p.span_y = kh; p.span_x = kw;
    for (int t = 0; t < kh * kw; ++t) p.tap_off[t] = ((t / kw) << 6) | (t % kw);
    return SHG_OK;
}

// mode: 0 = stride 1, symmetric padding `pad`        (conv2d_resample.py:145-147)
//       1 = stride 2, symmetric padding `pad`        (conv2d_resample.py:116-120, strided conv)
//       2 = transposed stride 2, padding 0 -> [2H+1, 2W+1]   (conv2d_resample.py:122-137)
// out_mode (mode 2 only): 0 = interleaved result; 1 = four phase planes [4][NB,O,H+1,W+1] for
// shg_upfir_planar_f32.  workspace (optional, caller owned) enables split-K for small grids; its size comes
// from shg_conv2d_workspace_bytes.
extern "C" int shg_conv2d_f32(const float* x, const float* wt, float* y, int NB, int I, int O, int OP, int H, int W, int kh, int kw,
                              int mode, int pad, int wgroups, long wstride, const float* in_scale, const float* out_scale,
                              const float* bias, const float* noise, int noise_mode, float noise_strength, int act, float alpha,
                              float gain, float clamp, const float* residual, int out_mode, void* workspace, size_t ws_bytes,
                              void* stream) {
    ConvParams p;
    int rc = conv_fill(p, x, wt, y, NB, I, O, OP, H, W, kh, kw, mode, pad, wgroups, wstride, in_scale, out_scale, bias, noise,
                       noise_mode, noise_strength, act, alpha, gain, clamp, residual, out_mode);
    if (rc != SHG_OK) return rc;
    return conv_dispatch(p, kh * kw, p.S, mode == 2, workspace, ws_bytes, (hipStream_t)stream);
}

// Bytes of split-K workspace that shg_conv2d_f32 can use for this problem (0 = it will not split).
extern "C" size_t shg_conv2d_workspace_bytes(int NB, int I, int O, int H, int W, int kh, int kw, int mode, int pad, int wgroups) {
    if (NB < 1 || I < 1 || O < 1) return 0;
    ConvParams p{};
    p.NB = NB; p.I = I; p.O = O; p.H = H; p.W = W; p.wgroups = wgroups < 1 ? 1 : wgroups;
    if (mode == 2) {   // small transposed convs (the 4-wave 64 x 128 tile) may split; planar output of 4 phase planes
        p.S = 1; p.span_y = 2; p.span_x = 2;
        const bool big = W + 1 >= conv_up_min() && H + 1 >= 16 && p.wgroups == 1;
        if (big) conv_tiles(p, O <= 64 ? 64 : 128, O <= 64 ? 256 : 128, true, 512);
        else conv_tiles(p, 64, 128, true, 512);
        const int ks = conv_ksplit(p.n_ptiles * p.n_otiles, shg_cdiv(I, 8), true);
        const size_t split = ks > 1 ? (size_t)ks * 4 * NB * O * (H + 1) * (W + 1) * sizeof(float) : 0;
        const size_t tail = big ? conv_tail_ws_bound(true) : 0;
        return split > tail ? split : tail;
    }
    const int S = mode == 0 ? 1 : 2;
    const int OH = (H + 2 * pad - kh) / S + 1, OW = (W + 2 * pad - kw) / S + 1;
    if (OH < 1 || OW < 1) return 0;
    const bool narrow = O <= 64;
    const int KC = kh * kw == 9 ? 8 : 32;
    const int xq = kh * kw == 1 ? 1 : (S == 1 ? 2 : (narrow ? 5 : 3));
    p.OHp = OH; p.OWp = OW; p.S = S; p.span_y = kh; p.span_x = kw;
    conv_tiles(p, narrow ? 64 : 128, narrow ? 256 : 128, false, xq * 256);
    int ks = conv_ksplit(p.n_ptiles * p.n_otiles, shg_cdiv(I, KC), true);
    if (kh * kw == 9 && !narrow && OW >= 32 && OH >= 8) {
        // the double-buffered kernels of conv_dispatch own 128 x 256 tiles: half as many workgroups, so they may split twice as wide.  (Planned
        // with the 128 x 128 tile alone, the 512-channel stride-2 layer at 65^2 -> 32^2 got scratch for 2 slices, the launch wanted 4,
        // found too little and ran UNSPLIT: 128 workgroups on 256 CUs, 604 us at batch 8 and at batch 2 alike.)
        ConvParams q = p;
        conv_tiles(q, 128, 256, false, xq * 256);
        const int ks2 = conv_ksplit(q.n_ptiles * q.n_otiles, shg_cdiv(I, KC), true);
        if (ks2 > ks) ks = ks2;
    }
    if (ks > 1) return (size_t)ks * NB * O * OH * OW * sizeof(float);
    return (kh * kw == 9 && !narrow && OW >= 32 && OH >= 8) ? conv_tail_ws_bound(false) : 0;
}

// ------------------------------------------------------------------------------------------------
// Weight preparation: [O,I,KH,KW] -> GEMM layout [I][taps][OP] (o contiguous, zero padded to OP)
// ------------------------------------------------------------------------------------------------

// scale[o] = gain * rsqrt(mean_{i,k} w^2)  (stylegan.py:146) when demod, else gain.
__global__ __launch_bounds__(256) void weight_scale_kernel(const float* w, float* scale, int IK, int demod, float gain) {
    const int o = blockIdx.x;
    float acc = 0.f;
    if (demod) {
        for (int k = threadIdx.x; k < IK; k += 256) { const float v = w[(long)o * IK + k]; acc += v * v; }
        __shared__ float red[256];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        acc = red[0];
    }
    if (threadIdx.x == 0) scale[o] = demod ? gain * rsqrtf(acc / (float)IK) : gain;
}

// wt[((o/64)*IP*KK + i*KK + t')*64 + o%64], t' = flip ? KK-1-t : t   (flip = w.flip([2,3]), conv2d_resample.py:32-33):
// 64-column blocks, so the [KC*KK][BO] slice a workgroup stages per chunk is one contiguous run of memory (a row-major
// [I*KK][OP] matrix would put every row of a tile on the same two L2 channels).  Rows for i in [I, IP) and columns
// in [O, OP) are zeros (IP = I rounded up to 32: the conv kernel reads whole chunks).
__global__ __launch_bounds__(256) void weight_transpose_kernel(const float* w, const float* scale, float* wt, int O, int I,
                                                               int IP, int KK, int OP, int flip) {
    __shared__ float tile[32][33];
    const int IK = I * KK;
    const int IPK = IP * KK;
    const int k0 = blockIdx.x * 32, o0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int o = o0 + r, k = k0 + tx;
        tile[r][tx] = (o < O && k < IK) ? w[(long)o * IK + k] * scale[o] : 0.f;      // zero beyond I*KK
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, o = o0 + tx;
        if (k >= IPK || o >= OP) continue;
        const int i = k / KK, t = k - i * KK;
        const int tt = flip ? KK - 1 - t : t;
        wt[((long)(o >> 6) * IPK + (long)i * KK + tt) * 64 + (o & 63)] = tile[tx][r];
    }
}

// wsq[i][o] = sum_t wt[i][t][o]^2   (for the demodulation coefficients, stylegan.py:155); row pitch OP
__global__ __launch_bounds__(256) void weight_sq_kernel(const float* wt, float* wsq, int I, int IPK, int KK, int OP) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)I * OP) return;
    const int i = (int)(e / OP), o = (int)(e - (long)i * OP);
    float acc = 0.f;
    for (int t = 0; t < KK; ++t) { const float v = wt[((long)(o >> 6) * IPK + (long)i * KK + t) * 64 + (o & 63)]; acc += v * v; }
    wsq[e] = acc;
}

// w: [O,I,KH,KW] fp32.  wt: [OP/64][IP*KK][64] out (OP = O rounded up to 64, IP = I rounded up to 32, padding zero).  wscale: [O] scratch/out.  wsq: [I*OP] out or null.
// demod=1 reproduces stylegan.py:146 (per-output-channel RMS normalisation) times `gain`;
// demod=0 multiplies by `gain` (conv2d_layer weight_gain, stylegan.py:227).
// One layout serves all three convolution modes (the transposed kernel indexes taps as ky*3+kx).
extern "C" int shg_conv_weight_prep_f32(const float* w, float* wt, float* wscale, float* wsq, int O, int I, int KH, int KW,
                                        int OP, int demod, float gain, int flip, void* stream) {
    SHG_CHECK_ARG(w && wt && wscale, "weight_prep: null pointer");
    SHG_CHECK_ARG(O >= 1 && I >= 1 && KH >= 1 && KW >= 1, "weight_prep: empty weight");
    SHG_CHECK_ARG(OP % 64 == 0 && OP >= O, "weight_prep: OP must be a multiple of 64 and >= O");
    hipStream_t s = (hipStream_t)stream;
    const int KK = KH * KW, IK = I * KK;
    hipLaunchKernelGGL(weight_scale_kernel, dim3(O), dim3(256), 0, s, w, wscale, IK, demod, gain);
    SHG_CHECK_LAUNCH();
    const int IP = (I + 31) / 32 * 32;
    hipLaunchKernelGGL(weight_transpose_kernel, dim3(shg_cdiv((long)IP * KK, 32), shg_cdiv(OP, 32)), dim3(256), 0, s, w, wscale, wt, O,
                       I, IP, KK, OP, flip);
    SHG_CHECK_LAUNCH();
    if (wsq) {
        hipLaunchKernelGGL(weight_sq_kernel, dim3(shg_cdiv((long)I * OP, 256)), dim3(256), 0, s, wt, wsq, I, IP * KK, KK, OP);
        SHG_CHECK_LAUNCH();
    }
    return SHG_OK;
}
