// Weight gradient of a 2-D convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces the cuDNN call inside Conv2dGradWeight of the reference's conv2d_gradfix.py:140-146
// (`aten::cudnn_convolution_backward_weight` / `..._transpose_backward_weight`), i.e. for y = conv2d(x, w, stride s, padding p)
//     dw[o, i, ky, kx] = sum_{n, oy, ox} g[n, o, oy, ox] * x[n, i, oy*s - p + ky, ox*s - p + kx]        (zero outside x).
// The transposed convolution's weight gradient is the same sum with the roles of the two tensors exchanged
// (dw[ci, co, ky, kx] = sum x[n,ci,y,x] * g[n,co, y*s - p + ky, x*s - p + kx]): the host passes g as `x` and x as `g`.
//
// GEMM view: M = 64 output channels, N = 64 input channels (x kh*kw taps), K = all output pixels of the batch.  One workgroup owns a
// 64 x 64 x taps tile of dw for one slice of K; its four waves own the 2 x 2 sub-tiles of 32 x 32 with all taps in registers
// (9 accumulator tiles for a 3x3 kernel) -- no cross-wave reduction.  K is walked in chunks of 64 (stride 1) or 32 (stride 2) pixels of one output row:
// the 64-channel piece of g and the kh x (64 s + kw - 1)-column window of x are staged in LDS pixel-major
// ([pixel][channel], pitch 65: coalesced global reads along a row become conflict-free LDS writes, and the operand reads of 32
// consecutive channels are conflict-free too); one A operand feeds the nine taps' MFMAs.  Staging is software-pipelined through
// registers (one workgroup per CU, 68 KB of LDS, the staged values live in the 512-register budget beside the 144
// accumulators).  K slices are summed by a second tiny kernel in slice order, so the result is
// deterministic (no atomics).
#include "shg_common.h"

typedef float wg_f32x16 __attribute__((ext_vector_type(16)));

struct WgradParams {
    const float* x;      // [NB, I, H, W]
    const float* g;      // [NB, O, OH, OW]
    float* out;          // dw [O, I, kh, kw] (nslice == 1) or partials [nslice][O][I][kh*kw]
    int NB, I, O, H, W, OH, OW, stride, pad;
    int chunks_x;        // PX-pixel pieces per output row
    int nchunk;          // NB * OH * chunks_x
    int nslice;
};

// ROWS = 2 (stride 1, OW == 32): a chunk is TWO whole output rows of 32 pixels (the same 64 operand columns; a one-row chunk would leave half
// of them empty: 990 us for the 512-channel 32^2 layers) -- the window is then KH + 1 rows of 34 columns, a staging line = two window rows.
template <int KH, int KW, int S, int ROWS = 1>
__global__ __launch_bounds__(512, 2) void conv_wgrad_kernel(const WgradParams p) {
    constexpr int TAPS = KH * KW, PX = S == 1 ? 64 : 32, RW = PX / ROWS, XW = RW * S + KW - 1, WROWS = KH + ROWS - 1, CP = 65;    // CP: channel pitch (odd)
    static_assert(ROWS == 1 || S == 1, "two-row chunks are a stride-1 form");
    __shared__ float Gs[PX * CP];                                // [pixel][o]
    __shared__ float Xs[WROWS * XW * CP];                        // [row][col][i]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int grp = wave >> 2, mo = (wave >> 1) & 1, nt = wave & 1;   // grp: the two k-step groups (two waves per SIMD)
    const int i0 = blockIdx.x * 64, o0 = blockIdx.y * 64, slice = blockIdx.z;
    const int per = (p.nchunk + p.nslice - 1) / p.nslice;
    const int c_begin = slice * per, c_end = min(p.nchunk, c_begin + per);
    wg_f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const long gplane = (long)p.OH * p.OW, xplane = (long)p.H * p.W;
    // Staging is software-pipelined through registers: the global loads of chunk c+1 are issued before the MFMAs of chunk c and
    // written to LDS after them, so their latency is covered by ~20 k cycles of matrix work instead of standing in front of it.
    // The window is staged LINE by line: a line = (channel, window row) = 64 columns, one per lane; wave w takes lines w, w+8, ...
    // -- the channel / row / image-row arithmetic of a line is wave-uniform (scalar unit) and the vector unit is left with one
    // address add and one select per element.  The KW-1 halo columns of all lines follow, one lane per element.
    // (Tried and slower: keeping the loaded values raw and masking them when they are written to LDS, 68 vs 77 TFLOP/s at
    // 512 channels; an explicit LDS-operand prefetch in the MFMA loop, 64.)
    constexpr int NLINE = 64 * WROWS / ROWS, LPW = NLINE / 8;    // x lines (ROWS window rows each), per wave
    constexpr int HC = XW - RW * S;                              // halo columns of a window row (KW - 1)
    constexpr int NHE = 64 * WROWS * HC, NHT = (NHE + 511) / 512;   // halo elements, per thread
    constexpr int GL = 64 * PX / 64 / 8;                         // g: wave-instructions per wave (a 64-lane instruction = 64 / PX lines)
    static_assert(RW * S * ROWS == 64, "a staging line is 64 columns (+ halo)");
    float rg[GL], rx[LPW + (NHT ? NHT : 1)];
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    auto load_chunk = [&](int c) __attribute__((always_inline)) {
        const int cx = ROWS == 1 ? c % p.chunks_x : 0;
        const int oy = ROWS == 1 ? (c / p.chunks_x) % p.OH : ROWS * (c % (p.OH / ROWS)), n = ROWS == 1 ? c / (p.chunks_x * p.OH) : c / (p.OH / ROWS);
        const int ox0 = cx * PX;                                 // (ROWS = 2: the chunk's 64 pixels are rows oy, oy + 1 -- contiguous in g)
        {
            const int gpx = lane % PX, gsub = lane / PX;         // pixel, line within the instruction
            const bool pok = ROWS > 1 || ox0 + gpx < p.OW;      // (ROWS = 2: pixel gpx of rows oy, oy + 1 -- 64 contiguous floats)
            const float* gp = p.g + ((long)n * p.O + o0) * gplane + (long)oy * p.OW + ox0 + (pok ? gpx : 0);
#pragma unroll
            for (int j = 0; j < GL; ++j) {
                const int o = (wv + 8 * j) * (64 / PX) + gsub;
                const bool ok = pok && o0 + o < p.O;
                const float v = gp[ok ? (long)o * gplane : 0];
                rg[j] = ok ? v : 0.f;
            }
        }
        const int lcol = ROWS == 1 ? lane : (lane & (RW - 1)), lrow = ROWS == 1 ? 0 : lane / RW;     // column / row of this lane inside a line
        const int ixl = ox0 * S - p.pad + lcol;
        const bool cok = ixl >= 0 && ixl < p.W;
        const float* xn = p.x + ((long)n * p.I + i0) * xplane;
#pragma unroll
        for (int j = 0; j < LPW; ++j) {
            const int L = wv + 8 * j, i = L / (WROWS / ROWS), r = (L - i * (WROWS / ROWS)) * ROWS + lrow;        // uniform up to lrow
            const int iy = oy * S - p.pad + r;
            const bool rok = iy >= 0 && iy < p.H && i0 + i < p.I;
            const bool ok = rok && cok;
            const float v = xn[ok ? (long)i * xplane + (long)iy * p.W + ixl : 0];
            rx[j] = ok ? v : 0.f;
        }
#pragma unroll
        for (int t = 0; t < NHT; ++t) {
            const int e = tid + 512 * t;
            const int L = e / (HC ? HC : 1), hc = RW * S + e - L * (HC ? HC : 1), i = L / WROWS, r = L - i * WROWS;
            const int iy = oy * S - p.pad + r, ix = ox0 * S - p.pad + hc;
            const bool ok = e < NHE && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && i0 + i < p.I;
            const float v = xn[ok ? (long)i * xplane + (long)iy * p.W + ix : 0];
            rx[LPW + t] = ok ? v : 0.f;
        }
    };
    auto store_chunk = [&]() __attribute__((always_inline)) {
        {
            const int gpx = lane % PX, gsub = lane / PX;
#pragma unroll
            for (int j = 0; j < GL; ++j) Gs[gpx * CP + (wv + 8 * j) * (64 / PX) + gsub] = rg[j];
        }
        const int lcol = ROWS == 1 ? lane : (lane & (RW - 1)), lrow = ROWS == 1 ? 0 : lane / RW;
#pragma unroll
        for (int j = 0; j < LPW; ++j) {
            const int L = wv + 8 * j, i = L / (WROWS / ROWS), r = (L - i * (WROWS / ROWS)) * ROWS + lrow;
            Xs[(r * XW + lcol) * CP + i] = rx[j];
        }
#pragma unroll
        for (int t = 0; t < NHT; ++t) {
            const int e = tid + 512 * t;
            const int L = e / (HC ? HC : 1), hc = RW * S + e - L * (HC ? HC : 1), i = L / WROWS, r = L - i * WROWS;
            if (e < NHE) Xs[(r * XW + hc) * CP + i] = rx[LPW + t];
        }
    };
    if (c_begin < c_end) load_chunk(c_begin);
    for (int c = c_begin; c < c_end; ++c) {
        __syncthreads();                                         // previous chunk fully consumed
        store_chunk();
        __syncthreads();
        if (c + 1 < c_end) load_chunk(c + 1);
        for (int ks = grp; ks < PX / 2; ks += 2) {               // the two wave groups take alternate k-steps
            const int k = 2 * ks + half;                         // pixel of this lane's operand row
            const float a = Gs[k * CP + mo * 32 + l31];
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Xs[((t / KW + (ROWS == 1 ? 0 : k / RW)) * XW + (ROWS == 1 ? k : (k & (RW - 1))) * S + (t % KW)) * CP + nt * 32 + l31],
                                                              acc[t], 0, 0, 0);
        }
    }
    // the second wave group hands its partial sums to the first through LDS (three taps at a time: 48 KB over the window buffer)
    float* red = Xs;
    constexpr int TB = sizeof(Xs) >= sizeof(float) * 4 * 3 * 1024 ? (TAPS < 3 ? TAPS : 3) : (sizeof(Xs) >= sizeof(float) * 4 * 2 * 1024 ? (TAPS < 2 ? TAPS : 2) : 1);
    static_assert(sizeof(Xs) >= sizeof(float) * 4 * TB * 1024, "reduction buffer");
#pragma unroll
    for (int t0 = 0; t0 < TAPS; t0 += TB) {
        __syncthreads();
        if (grp == 1) {
#pragma unroll
            for (int tt = 0; tt < TB; ++tt)
                if (t0 + tt < TAPS)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[(((wave & 3) * TB + tt) * 16 + r) * 64 + lane] = acc[t0 + tt][r];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int tt = 0; tt < TB; ++tt)
                if (t0 + tt < TAPS)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t0 + tt][r] += red[(((wave & 3) * TB + tt) * 16 + r) * 64 + lane];
        }
    }
    if (grp == 1) return;
    float* dst = p.out + (p.nslice > 1 ? (long)slice * p.O * p.I * TAPS : 0);
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + mo * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, i = i0 + nt * 32 + l31;
            if (o < p.O && i < p.I) dst[((long)o * p.I + i) * TAPS + t] = acc[t][r];
        }
}

// ---------------------------------------------------------------------------------------------
// The unmasked form (round 6).  Host-checked: pad == 0, I and O multiples of 64, OW a multiple of the chunk width, every window row and
// column inside x -- the FIR-padded stride-2 layers and the 1x1 layers of the training step: no element of a chunk needs a mask.
// What the ISA of the masked kernel above showed (tools/isa_flow.py): its selects (`ok ? v : 0`) were scheduled, with the `s_waitcnt
// vmcnt(0)` they need, BEFORE the MFMA loop -- the register pipeline described there did not exist in the binary (39 % of the wave cycles
// parked, profiles/r05_wgrad_pmc_summary.txt) -- and every operand was read from LDS right before its MFMA.  Here
//   * loads are unconditional (a wave-uniform line pointer + one lane offset); their values are first touched by LDS writes that are
//     issued in the SECOND half of the chunk's MFMA stream, into the other of two LDS stages: one barrier per chunk, the writes and
//     their address arithmetic in the shadow of the matrix pipe;
//   * the k-steps of a chunk are unrolled and the operands of step j + 1 are read before the MFMAs of step j issue.
// 1 047 -> 846 us (loads only) -> see MEASUREMENTS.md for the two-stage form, 128 -> 256 channels at 257^2 x 8.
// ---------------------------------------------------------------------------------------------
// ROWS > 1 (stride 2, narrow images: OW = 32 / ROWS = 16 or 8 with W = 2 OW + 1, the FIR-padded width): a chunk is ROWS WHOLE output rows.
// Its window -- (2 ROWS + 1) whole input rows -- is then ONE contiguous span of at most 192 floats per channel (5 x 33 = 165, 9 x 17 = 153):
// three 64-lane lines per channel, the same 24 loads per wave as the three window rows of the wide form, no halo; LDS position = offset in the
// span, so pixel (ry, rx) and tap (ky, kx) read position (2 ry + ky) W + 2 rx + kx.  (These layers ran on the masked kernel with half or
// three quarters of every chunk's operand columns empty: 35 TFLOP/s at 512 channels, 33^2 -> 16^2.)
template <int KH, int KW, int S, int ROWS = 1>
__global__ __launch_bounds__(512, 2) void conv_wgrad_full_kernel(const WgradParams p) {
    constexpr int TAPS = KH * KW, PX = S == 1 ? 64 : 32, RW = PX / ROWS, CP = 65;
    static_assert(ROWS == 1 || (S == 2 && KH == 3 && KW == 3), "several rows per chunk: the stride-2 3x3 form");
    constexpr int XW = ROWS == 1 ? PX * S + KW - 1 : S * RW + KW - S;          // window pitch (ROWS > 1: the image width itself)
    constexpr int HC = ROWS > 1 ? 0 : (KW > S ? KW - S : 0);     // halo columns a tap reads (window column PX * S + hc)
    constexpr int SPAN = (S * (ROWS - 1) + KH) * XW, LPC = ROWS == 1 ? KH : (SPAN + 63) / 64;     // ROWS > 1: floats / 64-lane lines per channel
    constexpr int GSZ = PX * CP, XSZ = (ROWS == 1 ? KH * XW : LPC * 64) * CP;
    __shared__ float Gs[2][GSZ];                                 // [stage][pixel][o]
    __shared__ float Xs[2][XSZ];                                 // [stage][row][col][i]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int grp = wave >> 2, mo = (wave >> 1) & 1, nt = wave & 1;
    const int i0 = blockIdx.x * 64, o0 = blockIdx.y * 64, slice = blockIdx.z;
    const int per = (p.nchunk + p.nslice - 1) / p.nslice;
    const int c_begin = slice * per, c_end = min(p.nchunk, c_begin + per);
    wg_f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const long gplane = (long)p.OH * p.OW, xplane = (long)p.H * p.W;
    constexpr int LPW = 8 * LPC;                                 // x lines (64 floats of one channel) per wave
    constexpr int NHE = 64 * KH * HC, NHT = (NHE + 511) / 512;   // halo elements, per thread
    constexpr int GL = 64 * PX / 64 / 8;                         // g: wave-instructions per wave (one = 64 / PX channel lines)
    constexpr int NST = GL + LPW + NHT;                          // staged values per thread
    float rg[GL], rx[LPW + (NHT ? NHT : 1)];
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    unsigned hoff[NHT ? NHT : 1];                                // a thread's halo elements: offset from the chunk's window origin / LDS slot
    int hdst[NHT ? NHT : 1];
#pragma unroll
    for (int t = 0; t < NHT; ++t) {
        const int e = min(tid + 512 * t, NHE - 1);               // (threads past the last element repeat it: same value, no mask)
        const int L = e / (HC ? HC : 1), hc = PX * S + e - L * (HC ? HC : 1), i = L / KH, r = L - i * KH;
        hoff[t] = (unsigned)i * (unsigned)xplane + (unsigned)r * (unsigned)p.W + (unsigned)hc;
        hdst[t] = (r * XW + hc) * CP + i;
    }
    // Wave w stages channels 8 w .. 8 w + 7 of both tensors: a line's address is the previous line's plus a plane (scalar adds, no division
    // by the row count), its LDS slot an immediate offset from one lane base.  The chunk coordinates advance incrementally.
    constexpr int CPI = 64 / PX;                                 // g channels per wave instruction
    static_assert(GL * CPI == 8, "eight channels per wave");
    const unsigned goff = (unsigned)(lane / PX) * (unsigned)gplane + (unsigned)(lane % PX);
    const unsigned xlast = (unsigned)min((LPC - 1) * 64 + lane, SPAN - 1);      // ROWS > 1: the span's last line stops at its end (no read past the tensor)
    const int gdst = (lane % PX) * CP + wv * 8 + lane / PX, xdst = lane * CP + wv * 8;
    const int rows_c = ROWS == 1 ? p.OH : p.OH / ROWS;           // chunk rows per image
    int ncx = c_begin % p.chunks_x, noy = (c_begin / p.chunks_x) % rows_c, nn = c_begin / (p.chunks_x * rows_c);      // the chunk load_next() fetches
    auto load_next = [&](bool advance) __attribute__((always_inline)) {
        const int ox0 = ncx * PX, oy = noy * ROWS;
        const float* gp = p.g + ((long)nn * p.O + o0 + wv * 8) * gplane + (long)oy * p.OW + ox0;            // wave-uniform (ROWS > 1: ROWS whole rows = PX contiguous pixels)
#pragma unroll
        for (int j = 0; j < GL; ++j) rg[j] = (gp + (long)(j * CPI) * gplane)[goff];
        const float* xb = p.x + ((long)nn * p.I + i0) * xplane + (long)(oy * S) * p.W + ox0 * S;            // window row 0, column 0, channel i0
        const float* xw = xb + (long)(wv * 8) * xplane;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
#pragma unroll
            for (int r = 0; r < LPC; ++r) {
                if constexpr (ROWS == 1) rx[jj * LPC + r] = (xw + (long)jj * xplane + (long)r * p.W)[lane];
                else rx[jj * LPC + r] = r + 1 < LPC ? (xw + (long)jj * xplane + r * 64)[lane] : (xw + (long)jj * xplane)[xlast];
            }
#pragma unroll
        for (int t = 0; t < NHT; ++t) rx[LPW + t] = xb[hoff[t]];
        if (advance) {
            if (++ncx == p.chunks_x) {
                ncx = 0;
                if (++noy == rows_c) { noy = 0; ++nn; }
            }
        }
    };
    // staged value q of this thread -> LDS stage b (q is a compile-time constant wherever this is called)
    auto store_item = [&](int q, int b) __attribute__((always_inline)) {
        if (q < GL) Gs[b][gdst + q * CPI] = rg[q];
        else if (q < GL + LPW) {
            const int jj = (q - GL) / LPC, r = (q - GL) - jj * LPC;
            Xs[b][xdst + r * (ROWS == 1 ? XW : 64) * CP + jj] = rx[q - GL];
        } else Xs[b][hdst[q - GL - LPW]] = rx[q - GL];
    };
    // LDS position (in pixels of the window image) of k-step j's first pixel, tap t: pixel k = 4 j + c (c = 2 grp + half < 4 stays inside a row)
    auto xpos = [](int j, int t) constexpr { return (S * ((4 * j) / RW) + t / KW) * XW + S * ((4 * j) % RW) + t % KW; };
    constexpr int NJ = PX / 4, H0 = NJ / 2, NH = NJ - H0;        // k-steps per wave and chunk; the stores ride on steps H0 .. NJ - 1
    if (c_begin < c_end) {
        load_next(c_begin + 1 < c_end);
#pragma unroll
        for (int q = 0; q < NST; ++q) store_item(q, 0);
    }
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const int b = (c - c_begin) & 1;
        const float* ga = Gs[b] + (2 * grp + half) * CP + mo * 32 + l31;
        const float* xa = Xs[b] + (2 * grp + half) * S * CP + nt * 32 + l31;
        float a_cur = ga[0], b_cur[TAPS];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) b_cur[t] = xa[xpos(0, t) * CP];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {                           // k = 2 (grp + 2 j) + half: constant offsets from the two lane bases
            float a_nxt = 0.f, b_nxt[TAPS];
            if (j + 1 < NJ) {
                a_nxt = ga[4 * (j + 1) * CP];
#pragma unroll
                for (int t = 0; t < TAPS; ++t) b_nxt[t] = xa[xpos(j + 1, t) * CP];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[t], acc[t], 0, 0, 0);
            // the loads of chunk c + 1 (and their scalar address arithmetic) are issued in the shadow of the first k-step's MFMAs, not between
            // the barrier and them (the last chunk loads itself again: no branch around the loads)
            if (j == 0) load_next(c + 2 < c_end);
            if (j >= H0) {                                       // this step's share of the next chunk's values -> the other stage
#pragma unroll
                for (int q = (j - H0) * NST / NH; q < (j - H0 + 1) * NST / NH; ++q) store_item(q, b ^ 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (j + 1 < NJ) {
                a_cur = a_nxt;
#pragma unroll
                for (int t = 0; t < TAPS; ++t) b_cur[t] = b_nxt[t];
            }
        }
        __syncthreads();
    }
    // the second wave group hands its partial sums to the first through LDS (three taps at a time: 48 KB over the window stages)
    float* red = &Xs[0][0];
    constexpr int TB = 2 * XSZ >= 4 * 3 * 1024 ? (TAPS < 3 ? TAPS : 3) : 1;
    static_assert(2 * XSZ >= 4 * TB * 1024, "reduction buffer");
#pragma unroll
    for (int t0 = 0; t0 < TAPS; t0 += TB) {
        if (t0) __syncthreads();
        if (grp == 1) {
#pragma unroll
            for (int tt = 0; tt < TB; ++tt)
                if (t0 + tt < TAPS)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[(((wave & 3) * TB + tt) * 16 + r) * 64 + lane] = acc[t0 + tt][r];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int tt = 0; tt < TB; ++tt)
                if (t0 + tt < TAPS)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t0 + tt][r] += red[(((wave & 3) * TB + tt) * 16 + r) * 64 + lane];
        }
    }
    if (grp == 1) return;
    float* dst = p.out + (p.nslice > 1 ? (long)slice * p.O * p.I * TAPS : 0);
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + mo * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, i = i0 + nt * 32 + l31;
            dst[((long)o * p.I + i) * TAPS + t] = acc[t][r];
        }
}

// Small images (OW a power of two below the chunk width: the 4^2 ... 32^2 layers).  A chunk of the kernel above is a piece of ONE
// output row, so a 4-pixel row would fill 4 of its 64 operand columns.  Here a chunk is PX consecutive pixels of the flattened
// (image, row, column) space = R = PX / OW whole output rows, each with its own KH-row window in LDS ([row-in-chunk][window row]
// [column][channel]); the operand address of a pixel is computed from (k / OW, k % OW).  No register pipeline: these layers are
// small, the staging is element-indexed (shifts: the window pitch is padded to a power of two for the index arithmetic only).
template <int KH, int KW, int S>
__global__ __launch_bounds__(512, 2) void conv_wgrad_packed_kernel(const WgradParams p) {
    constexpr int TAPS = KH * KW, PX = S == 1 ? 64 : 32, CP = 65;
    constexpr int XMAX = S == 1 ? 288 : 240;                    // window elements per channel, worst case (OW = 4)
    __shared__ float Gs[PX * CP];
    __shared__ float Xs[XMAX * CP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int grp = wave >> 2, mo = (wave >> 1) & 1, nt = wave & 1;
    const int i0 = blockIdx.x * 64, o0 = blockIdx.y * 64, slice = blockIdx.z;
    const int per = (p.nchunk + p.nslice - 1) / p.nslice;
    const int c_begin = slice * per, c_end = min(p.nchunk, c_begin + per);
    const int lw = 31 - __builtin_clz(p.OW), R = PX >> lw;      // log2 OW, output rows per chunk
    const int lr = 31 - __builtin_clz(R), loh = 31 - __builtin_clz(p.OH);      // (host: OH a power of two)
    const int XWs = p.OW * S + KW - 1;                           // window columns
    const int lxp = 32 - __builtin_clz(XWs - 1);                 // log2 of the padded pitch used for indexing
    const int WR = R * KH;                                       // window rows per chunk
    const int NE = (64 * WR) << lxp;                             // indexed elements
    const int rows_total = p.NB * p.OH;
    wg_f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const long gplane = (long)p.OH * p.OW, xplane = (long)p.H * p.W;
    const int gpx = lane % PX, gsub = lane / PX;
    int tapoff[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) tapoff[t] = ((t / KW) * XWs + (t % KW)) * CP;
    for (int c = c_begin; c < c_end; ++c) {
        __syncthreads();
        {   // g: pixel gpx of the chunk = (row c*R + gpx / OW, column gpx % OW)
            const int row = c * R + (gpx >> lw), cx = gpx & (p.OW - 1);
            const int n = row >> loh, oy = row & (p.OH - 1);
            const bool pok = row < rows_total;
            const float* gp = p.g + ((long)(pok ? n : 0) * p.O + o0) * gplane + (long)(pok ? oy : 0) * p.OW + cx;
#pragma unroll
            for (int j = 0; j < 64 * PX / 512; ++j) {
                const int o = (wave + 8 * j) * (64 / PX) + gsub;
                const bool ok = pok && o0 + o < p.O;
                const float v = gp[ok ? (long)o * gplane : 0];
                Gs[gpx * CP + o] = ok ? v : 0.f;
            }
        }
        for (int e0 = 0; e0 < NE; e0 += 512 * 8) {
            float v[8];
            int dst[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // element = ((channel, row in chunk), window row, column): R, OH and the padded pitch are powers of two, KH a constant --
                // shifts and one division by a constant (the decode with runtime divisions by WR and OH cost more issue slots than the MFMAs)
                const int e = e0 + tid + 512 * j;
                const int col = e & ((1 << lxp) - 1), t = e >> lxp;
                const int q = t / KH, ty = t - q * KH;
                const int rr = q & (R - 1), ch = q >> lr, wr = rr * KH + ty;
                const int row = c * R + rr;
                const int n = row >> loh, oy = row & (p.OH - 1);
                const int iy = oy * S - p.pad + ty, ix = col - p.pad;
                const bool inwin = e < NE && col < XWs;
                const bool ok = inwin && row < rows_total && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && i0 + ch < p.I;
                const float val = p.x[ok ? ((long)n * p.I + i0 + ch) * xplane + (long)iy * p.W + ix : 0];
                v[j] = ok ? val : 0.f;
                dst[j] = inwin ? (wr * XWs + col) * CP + ch : -1;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (dst[j] >= 0) Xs[dst[j]] = v[j];
        }
        __syncthreads();
        for (int ks = grp; ks < PX / 2; ks += 2) {
            const int k = 2 * ks + half;
            const float a = Gs[k * CP + mo * 32 + l31];
            const int base = ((k >> lw) * KH * XWs + (k & (p.OW - 1)) * S) * CP + nt * 32 + l31;
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Xs[base + tapoff[t]], acc[t], 0, 0, 0);
        }
    }
    float* red = Xs;
    constexpr int TB = TAPS < 3 ? TAPS : 3;
#pragma unroll
    for (int t0 = 0; t0 < TAPS; t0 += TB) {
        __syncthreads();
        if (grp == 1) {
#pragma unroll
            for (int tt = 0; tt < TB; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(((wave & 3) * TB + tt) * 16 + r) * 64 + lane] = acc[t0 + tt][r];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int tt = 0; tt < TB; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t0 + tt][r] += red[(((wave & 3) * TB + tt) * 16 + r) * 64 + lane];
        }
    }
    if (grp == 1) return;
    float* dst = p.out + (p.nslice > 1 ? (long)slice * p.O * p.I * TAPS : 0);
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + mo * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, i = i0 + nt * 32 + l31;
            if (o < p.O && i < p.I) dst[((long)o * p.I + i) * TAPS + t] = acc[t][r];
        }
}

// G slice groups per workgroup (256 / G elements each): thread (g, el) sums slices g, g + G, ... of its element in four interleaved
// accumulators, the groups are added in group order through LDS -- a fixed order for a given shape (deterministic).  One thread per element
// over ALL slices left a 64-channel layer (36 864 weights x 1 024 slices = 151 MB) to 144 workgroups of serial strided loads.
template <int G>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* part, float* dw, long n, int nslice) {
    __shared__ float red[256];
    constexpr int EL = 256 / G;
    const int el = threadIdx.x % EL, g = threadIdx.x / EL;
    const long e = (long)blockIdx.x * EL + el;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (e < n) {
        const float* pe = part + e;
        int sl = g;
        for (; sl + 3 * G < nslice; sl += 4 * G) {
            v0 += pe[(long)sl * n]; v1 += pe[(long)(sl + G) * n]; v2 += pe[(long)(sl + 2 * G) * n]; v3 += pe[(long)(sl + 3 * G) * n];
        }
        for (; sl < nslice; sl += G) v0 += pe[(long)sl * n];
    }
    float v = (v0 + v1) + (v2 + v3);
    if (G > 1) {
        red[threadIdx.x] = v;
        __syncthreads();
        if (g == 0)
            for (int k = 1; k < G; ++k) v += red[k * EL + el];
    }
    if (g == 0 && e < n) dw[e] = v;
}

static void launch_wgrad_reduce(const float* part, float* dw, long n, int nslice, hipStream_t s);
void shg_launch_wgrad_reduce(const float* part, float* dw, long n, int nslice, hipStream_t s) { launch_wgrad_reduce(part, dw, n, nslice, s); }    // conv_wgrad_wino.hip
static void launch_wgrad_reduce(const float* part, float* dw, long n, int nslice, hipStream_t s) {
    int G = 1;
    while (G < 16 && (n + 256 / G - 1) / (256 / G) < 2048 && nslice >= 8 * G) G *= 2;
    const dim3 grid((unsigned)((n + 256 / G - 1) / (256 / G)));
    switch (G) {
        case 1: hipLaunchKernelGGL(wgrad_reduce_kernel<1>, grid, dim3(256), 0, s, part, dw, n, nslice); break;
        case 2: hipLaunchKernelGGL(wgrad_reduce_kernel<2>, grid, dim3(256), 0, s, part, dw, n, nslice); break;
        case 4: hipLaunchKernelGGL(wgrad_reduce_kernel<4>, grid, dim3(256), 0, s, part, dw, n, nslice); break;
        case 8: hipLaunchKernelGGL(wgrad_reduce_kernel<8>, grid, dim3(256), 0, s, part, dw, n, nslice); break;
        default: hipLaunchKernelGGL(wgrad_reduce_kernel<16>, grid, dim3(256), 0, s, part, dw, n, nslice); break;
    }
}

// small-image form: 3x3, OW a power of two of at most a quarter of a chunk (whole rows per chunk), x rows fully inside the window
static bool wgrad_packed(int OW, int OH, int W, int kh, int stride, int pad) {
    const int PX = stride == 1 ? 64 : 32;
    // (rows of half a chunk stay on the row-piece kernel: 990 vs 1039 us at 512 channels, 32^2)
    return kh == 3 && OW >= 4 && OW <= PX / 4 && (OW & (OW - 1)) == 0 && (OH & (OH - 1)) == 0 && W <= OW * stride + 2 && pad <= 2;
}

static int wgrad_slices(int NB, int I, int O, int OH, int OW, int taps) {
    const long tiles = (long)shg_cdiv(I, 64) * shg_cdiv(O, 64);
    const long nchunk = (long)NB * OH * shg_cdiv(OW, 32);
    // 3x3: ONE workgroup per CU in ONE round (238 registers x 512 threads: a CU holds one) -- 256 / 512 / 768 / 1024 workgroups measured
    // 86 / 83 / 80 / 78 TFLOP/s at 64 channels x 512^2 and 34 / 29 / 25 / 23 at 512 channels x 16^2 (every further round re-pays prologue,
    // epilogue and 147 KB of partial sums per workgroup); the thin 1x1 layers are load-bound and want more workgroups in flight
    long s = ((taps > 1 ? 256 : 4 * 256) + tiles - 1) / tiles;
    const long per_slice = (long)O * I * taps * (long)sizeof(float);
    const long cap = (256L << 20) / per_slice;                   // at most 256 MB of partial sums
#ifdef SHG_ABLATE
    if (const char* e = getenv("SHG_WGRAD_WGS")) s = (atol(e) + tiles - 1) / tiles;       // study switch: workgroups in total (--ablate build, tools/_variants)
#endif
    if (s > cap) s = cap;
    if (s > nchunk) s = nchunk;
    if (s > 1024) s = 1024;
    return s < 1 ? 1 : (int)s;
}

// ---------------------------------------------------------------------------------------------
// Thin 1x1 layers (fromRGB: 4 -> 64, toRGB: C -> 3): dw[o, i] = sum_{n, p} g[n, o, p] x[n, i, p] with one side of at most 8 channels is a
// STREAMING problem -- read the fat tensor once (1.07 GB for the critic's fromRGB at 512^2 x 16: 215 us at the rate the pointwise kernels
// reach) -- that the MFMA tiles above served at 2.5x that (553 us: one (o, i) tile, channels padded to the tile, parallel only over pixel
// slices).  Here a wave owns IT x 256 pixels of one image (its thin-side values stay in registers), walks 64 fat channels with
// coalesced 16-byte loads, and reduces each channel's T dot products across its lanes; a workgroup = 4 waves = 4 adjacent pixel segments,
// blockIdx.y = the 64-channel block.  Partial sums [pixel chunk][O x I] -> the fixed-order reduction above (deterministic).
// ---------------------------------------------------------------------------------------------
template <int TT, int IT>
__global__ __launch_bounds__(256) void wgrad_thin_kernel(const float* fat, const float* thin, float* part, int C, int T, int HW, int chunks_per_n,
                                                         int out_sc, int out_st, int OI) {
    __shared__ float tab[4][64][TT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x / chunks_per_n, ch = blockIdx.x - n * chunks_per_n;
    const long p0 = ((long)ch * 4 + wave) * (IT * 256) + lane * 4;
    float4 tv[TT][IT];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int it = 0; it < IT; ++it)
            tv[t][it] = t < T ? *reinterpret_cast<const float4*>(thin + ((long)n * T + t) * HW + p0 + it * 256) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int c0 = blockIdx.y * 64, cn = min(64, C - c0);
    const float* fp = fat + ((long)n * C + c0) * HW + p0;
#pragma unroll 2
    for (int cc = 0; cc < cn; ++cc) {
        float4 f[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) f[it] = *reinterpret_cast<const float4*>(fp + (long)cc * HW + it * 256);
        float acc[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            float a = 0.f;
#pragma unroll
            for (int it = 0; it < IT; ++it) a += f[it].x * tv[t][it].x + f[it].y * tv[t][it].y + f[it].z * tv[t][it].z + f[it].w * tv[t][it].w;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m, 64);
            acc[t] = a;
        }
        if (lane == 0) {
#pragma unroll
            for (int t = 0; t < TT; ++t) tab[wave][cc][t] = acc[t];
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < cn * TT; k += 256) {
        const int cc = k / TT, t = k - cc * TT;
        if (t < T) part[(long)blockIdx.x * OI + (long)(c0 + cc) * out_sc + (long)t * out_st] = (tab[0][cc][t] + tab[1][cc][t]) + (tab[2][cc][t] + tab[3][cc][t]);
    }
}

// pixel chunks of the thin form (0: the layer is not one -- 1x1, stride 1, one side of at most 8 channels, whole 1024-pixel workgroup chunks)
static int wgrad_thin_chunks(int NB, int I, int O, int H, int W, int OH, int OW, int kh, int kw, int stride, int pad) {
    if (kh != 1 || kw != 1 || stride != 1 || pad != 0 || H != OH || W != OW) return 0;
    if ((I > 8 && O > 8) || ((long)H * W) % 1024 != 0) return 0;
    const long hw = (long)H * W;
    const long chunks = (long)NB * (hw % 4096 == 0 ? hw / 4096 : hw / 1024);
    return chunks <= 8192 ? (int)chunks : 0;
}

// bytes of scratch shg_conv2d_wgrad_f32 needs for this problem (0: none)
extern "C" size_t shg_conv2d_wgrad_workspace_bytes(int NB, int I, int O, int OH, int OW, int kh, int kw) {
    const int s = wgrad_slices(NB, I, O, OH, OW, kh * kw);
    size_t need = s > 1 ? (size_t)s * O * I * kh * kw * sizeof(float) : 0;
    const size_t thin = (size_t)wgrad_thin_chunks(NB, I, O, OH, OW, OH, OW, kh, kw, 1, 0) * O * I * sizeof(float);      // (taken when stride 1, pad 0)
    return thin > need ? thin : need;
}

// dw [O, I, kh, kw] = weight gradient of y = conv2d(x [NB,I,H,W], w, stride, pad) given g = dL/dy [NB,O,OH,OW];
// kh x kw = 3x3 or 1x1, stride 1 or 2.  workspace: shg_conv2d_wgrad_workspace_bytes.
extern "C" int shg_conv2d_wgrad_f32(const float* x, const float* g, float* dw, int NB, int I, int O, int H, int W, int OH, int OW,
                                    int kh, int kw, int stride, int pad, void* workspace, size_t ws_bytes, void* stream) {
    SHG_CHECK_ARG(x && g && dw, "conv2d_wgrad: null pointer");
    SHG_CHECK_ARG(NB >= 1 && I >= 1 && O >= 1 && H >= 1 && W >= 1 && OH >= 1 && OW >= 1, "conv2d_wgrad: bad shape");
    SHG_CHECK_ARG((kh == 3 && kw == 3) || (kh == 1 && kw == 1), "conv2d_wgrad: 3x3 and 1x1 kernels only");
    SHG_CHECK_ARG(stride == 1 || stride == 2, "conv2d_wgrad: stride 1 or 2");
    SHG_CHECK_ARG(pad >= 0 && (OH - 1) * stride - pad + kh - 1 < H + pad && (OW - 1) * stride - pad + kw - 1 < W + pad,
                  "conv2d_wgrad: output extent does not match x, stride and padding");
    const int thin_chunks = wgrad_thin_chunks(NB, I, O, H, W, OH, OW, kh, kw, stride, pad);
    if (thin_chunks > 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g)) & 15) == 0 && workspace &&
        ws_bytes >= (size_t)thin_chunks * O * I * sizeof(float)) {
        const bool thin_in = I <= O;                               // fat side: the gradient (fromRGB) or the activations (toRGB)
        const float* fat = thin_in ? g : x;
        const float* thin = thin_in ? x : g;
        const int C = thin_in ? O : I, T = thin_in ? I : O, HW = H * W, it4 = HW % 4096 == 0;
        const int cpn = thin_chunks / NB, sc = thin_in ? I : 1, st = thin_in ? 1 : I;
        const dim3 grid(thin_chunks, shg_cdiv(C, 64));
        hipStream_t s = (hipStream_t)stream;
        float* part = (float*)workspace;
        if (T <= 4 && it4) hipLaunchKernelGGL((wgrad_thin_kernel<4, 4>), grid, dim3(256), 0, s, fat, thin, part, C, T, HW, cpn, sc, st, O * I);
        else if (T <= 4) hipLaunchKernelGGL((wgrad_thin_kernel<4, 1>), grid, dim3(256), 0, s, fat, thin, part, C, T, HW, cpn, sc, st, O * I);
        else if (it4) hipLaunchKernelGGL((wgrad_thin_kernel<8, 4>), grid, dim3(256), 0, s, fat, thin, part, C, T, HW, cpn, sc, st, O * I);
        else hipLaunchKernelGGL((wgrad_thin_kernel<8, 1>), grid, dim3(256), 0, s, fat, thin, part, C, T, HW, cpn, sc, st, O * I);
        SHG_CHECK_LAUNCH();
        launch_wgrad_reduce(part, dw, (long)O * I, thin_chunks, s);
        SHG_CHECK_LAUNCH();
        return SHG_OK;
    }
    WgradParams p{};
    p.x = x; p.g = g; p.NB = NB; p.I = I; p.O = O; p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.stride = stride; p.pad = pad;
    // the unmasked forms (conv_wgrad_full_kernel): every line, halo column and channel of every chunk inside the tensors.
    //   * narrow FIR-padded stride-2 layers (OW = 16 / 8, W = 2 OW + 1): 2 / 4 whole output rows per chunk;
    //   * 1x1 layers: an image is one row of H W pixels (x and g have the same extent, nothing couples neighbouring pixels);
    //   * FIR-padded stride-2 layers with rows of whole 32-pixel chunks.
    const bool chan64 = pad == 0 && I % 64 == 0 && O % 64 == 0 && 64L * H * W < (1L << 31) && 2L * OH * OW < (1L << 31);
    int full_rows = 0;
    if (chan64 && kh == 3 && stride == 2 && (OW == 16 || OW == 8) && W == 2 * OW + 1 && (OH - 1) * 2 + 2 < H && OH % (32 / OW) == 0) full_rows = 32 / OW;
    if (chan64 && kh == 1 && stride == 1 && H == OH && W == OW && ((long)H * W) % 64 == 0) {
        p.H = p.OH = 1;
        p.W = p.OW = H * W;
        H = OH = 1;
        W = OW = p.W;
    }
    const bool packed = !full_rows && wgrad_packed(OW, OH, W, kh, stride, pad);
    const bool two_rows = !packed && kh == 3 && stride == 1 && OW == 32 && (OH % 2) == 0 && W == 32 + 2 - 2 * pad;      // 32-pixel rows: two per chunk
    const int px = stride == 1 ? 64 : 32;
    const bool full = !packed && !two_rows && !full_rows && chan64 && OW % px == 0 && (OH - 1) * stride + kh - 1 < H &&
                      (OW - px) * stride + 63 + (kw > stride ? kw - stride : 0) < W;
    p.chunks_x = full_rows ? 1 : shg_cdiv(OW, stride == 1 ? 64 : 32);
    p.nchunk = full_rows ? NB * (OH / full_rows)
                         : (packed ? shg_cdiv(NB * OH * OW, stride == 1 ? 64 : 32) : (two_rows ? NB * (OH / 2) : NB * OH * p.chunks_x));
    p.nslice = wgrad_slices(NB, I, O, OH, OW, kh * kw);
    if (p.nslice > p.nchunk) p.nslice = p.nchunk;
    const size_t need = p.nslice > 1 ? (size_t)p.nslice * O * I * kh * kw * sizeof(float) : 0;
    SHG_CHECK_ARG(need == 0 || (workspace && ws_bytes >= need), "conv2d_wgrad: workspace too small (shg_conv2d_wgrad_workspace_bytes)");
    p.out = p.nslice > 1 ? (float*)workspace : dw;
    const dim3 grid(shg_cdiv(I, 64), shg_cdiv(O, 64), p.nslice);
    hipStream_t s = (hipStream_t)stream;
    if (packed && stride == 1) hipLaunchKernelGGL((conv_wgrad_packed_kernel<3, 3, 1>), grid, dim3(512), 0, s, p);
    else if (packed) hipLaunchKernelGGL((conv_wgrad_packed_kernel<3, 3, 2>), grid, dim3(512), 0, s, p);
    else if (two_rows) hipLaunchKernelGGL((conv_wgrad_kernel<3, 3, 1, 2>), grid, dim3(512), 0, s, p);
    else if (full_rows == 2) hipLaunchKernelGGL((conv_wgrad_full_kernel<3, 3, 2, 2>), grid, dim3(512), 0, s, p);
    else if (full_rows == 4) hipLaunchKernelGGL((conv_wgrad_full_kernel<3, 3, 2, 4>), grid, dim3(512), 0, s, p);
    else if (full && kh == 3 && stride == 2) hipLaunchKernelGGL((conv_wgrad_full_kernel<3, 3, 2>), grid, dim3(512), 0, s, p);
    else if (full && kh == 1 && stride == 1) hipLaunchKernelGGL((conv_wgrad_full_kernel<1, 1, 1>), grid, dim3(512), 0, s, p);
    else if (kh == 3 && stride == 1) hipLaunchKernelGGL((conv_wgrad_kernel<3, 3, 1>), grid, dim3(512), 0, s, p);
    else if (kh == 3) hipLaunchKernelGGL((conv_wgrad_kernel<3, 3, 2>), grid, dim3(512), 0, s, p);
    else if (stride == 1) hipLaunchKernelGGL((conv_wgrad_kernel<1, 1, 1>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<1, 1, 2>), grid, dim3(512), 0, s, p);
    SHG_CHECK_LAUNCH();
    if (p.nslice > 1) {
        const long n = (long)O * I * kh * kw;
        launch_wgrad_reduce((const float*)workspace, dw, n, p.nslice, s);
        SHG_CHECK_LAUNCH();
    }
    return SHG_OK;
}
