// fp16 gather convolution, persistent LDS-DMA form ("ring" kernel) -- the stride-1-read launches of conv_f16.hip's conv_taps (3x3 stride-1
// layers forward and as input gradients; reference: the cuDNN half convolutions behind stylegan.py:136-138,172-181, comodgan.py:40-47,305-309,
// conv2d_gradfix.py:118-139).  gfx950 only.  Same arithmetic as conv_f16_kernel in the same order (chunk -> tap -> k-step on
// v_mfma_f32_32x32x16_f16, fp32 accumulation, one rounding): bit-identical results.
//
// What is different from conv_f16_kernel (profiles/r05_f16_pmc_base_summary.txt: matrix pipe busy 0.41, half the wave cycles in issue stalls,
// 8 192 workgroups of 2 chunks each at 64 channels with a 4 000-cycle entry and an 8 000-cycle LDS-transposing epilogue per workgroup):
//   * one PERSISTENT 512-thread workgroup per CU walks a list of tiles (16 x 32 output pixels x 64 output channels); a step = one 32-channel
//     chunk of one tile; the operands of step s+1 -- the 18 x 34-pixel input patch AND the chunk's weight slab (36 KiB in MFMA operand order)
//     -- are moved global -> LDS by `buffer_load_dwordx4 ... lds` (no staging registers, the descriptor's range check writes the zero padding)
//     while step s is multiplied, straight through tile boundaries: no per-tile entry latency;
//   * BOTH operands come from LDS (the four waves of conv_f16_kernel each loaded every weight operand from L1/L2: 64 B/clk/CU, the whole
//     vector-L1 rate).  Weight pieces are lane-linear (conflict-free ds_read_b128); the patch is stored as two k-step planes of 32 bytes per
//     pixel with the two 16-byte halves of a pixel exchanged where bit 3 of the pixel index is set: the 16 lanes of a ds_read_b128 group then
//     hit 16 different 16-byte slots of the 256-byte bank row for every tap shift (the swizzle is applied on the GLOBAL side of the DMA, the
//     LDS image of a DMA piece is lane-linear by construction);
//   * with I = 64 and one output-channel tile the two weight slabs stay resident (loaded once per workgroup): the layer is one pass over the
//     NHWC tensor (HBM-bound regime);
//   * the accumulators leave through v_permlane32_swap pairs as 16-byte channel runs (no LDS transpose, no epilogue barriers); the stores of
//     tile T are issued after the DMA requests of the following step, so the vmcnt(0) that ends a step never waits for a fresh store.
//   * the layer tail of conv_f16_kernel's store pass (y = A(half(conv) * out_scale[n,o] + noise * strength + bias[o]): the critic's and the encoder's
//     bias + lrelu_agc, the modulated layers' demodulation / noise of the inference route) is applied to the accumulators in registers, same
//     operations in the same order; its operands of a tile (64 bias and out_scale values, 16 x 32 noise values) arrive by DMA like everything else.
// LDS map (bytes): [0, 73 728) two weight stages | [73 728, 155 648) two patch stages of 2 planes x 640 pixels x 32 B | 2 x 4 KiB per-tile
// parameters (bias | out_scale | noise 16 rows x 128 B, each in its own KiB).
#include <type_traits>
#include "shg_common.h"
#include "conv_f16_p.h"

namespace f16 {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifdef SHG_RING_TRACE
// timeline study (python sh-gan_amd/build.py --variant=ringtrace -DSHG_RING_TRACE=1, tools/ring_trace.py): workgroup 0 records s_memtime of waves 0 and 7 at
// four points of its first 64 steps: step entry | past vmcnt(0) + barrier | requests of the next step issued (+ deferred stores) | multiplied (+ packed)
__device__ long long shg_ring_trace_buf[2 * 64 * 4];
extern "C" int shg_ring_trace_read(long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(shg_ring_trace_buf), sizeof(shg_ring_trace_buf)); }
#define RING_TRACE(slot) do { if (blockIdx.x == 0 && s < 64 && (wave == 0 || wave == 7) && lane == 0) shg_ring_trace_buf[((wave ? 1 : 0) * 64 + s) * 4 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define RING_TRACE(slot) do { } while (0)
#endif

namespace ring {

constexpr int TH = 16, TW = 32;                 // output pixels of a tile: wave w owns rows 2w, 2w + 1 (two 32-pixel MFMA column blocks)
constexpr int PPX = 640;                        // patch pixels (<= 18 x 34 = 612) rounded up to whole 32-pixel DMA pieces
constexpr int PLANE = PPX * 32;                 // bytes of one k-step plane (16 channels = 32 B per pixel)
constexpr int PSTAGE = 2 * PLANE;               // 40 960
constexpr int WSTAGE = 36 * 1024;               // 9 taps x 2 k-steps x 2 channel blocks x 1 KiB
constexpr int L_W = 0, L_P = 2 * WSTAGE, L_PRM = L_P + 2 * PSTAGE;
constexpr int PRM = 4096, PRM_SCALE = 1024, PRM_NOISE = 2048;      // a tile's parameter slot: bias | out_scale | noise rows -- 1 KiB apart: a DMA request
                                                                   // writes all 64 lanes' 16 bytes (zeros for the lanes out of range)
constexpr int LDS_BYTES = L_PRM + 2 * PRM;      // 163 840: all of the CU's LDS
constexpr unsigned OOB = 0x80000000u;           // a byte offset no descriptor range admits: the DMA writes zeros for that lane

__device__ __forceinline__ i32x4 make_srd(const void* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    i32x4 s;
    s[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    s[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    s[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    s[3] = 0x00020000;                          // raw buffer, dword data format (tools/micro/lds_dma_probe.hip pins the semantics used here)
    return s;
}

// 64 lanes x 16 bytes global -> LDS [lds_addr + 16 lane]; lanes whose voff + soff fails the range check deliver zeros
__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned voff, i32x4 srd, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(srd), "s"(soff));
}

struct Coord { int n, ty, tx, ot; };

struct RingDiv { unsigned n_ot, tiles_x, tiles_y, m_ot, m_tx, m_ty; };    // divisors of the tile decode and floor(2^32 / divisor)

// n / d for wave-uniform values with a precomputed m = floor((2^32 - 1) / d): the estimate is at most one short (scalar unit: ~8 instructions
// instead of the ~40 of a runtime division -- the decode of the next tile stood between two steps' multiplies: tools/ring_trace.py 'issue')
__device__ __forceinline__ unsigned fastdiv(unsigned n, unsigned d, unsigned m, unsigned& rem) {
    unsigned q = __umulhi(n, m), r = n - q * d;
    if (r >= d) { ++q; r -= d; }
    rem = r;
    return q;
}

template <int NT>
__global__ __launch_bounds__(512) void conv_f16_ring_kernel(const ConvP p, const int ntiles, const RingDiv dv) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int n_ot = (int)dv.n_ot;
    const int PW = NT == 9 ? TW + 2 : p.PW;                      // (the 3x3 patch is 18 x 34: a constant divisor for the piece -> (row, column) map)
    const int nchunks = p.I >> 5, c16n = p.I >> 4, npix = p.PH * PW;
    const bool resident = n_ot == 1 && nchunks <= 2;            // weight stage of step s = chunk's own slab: loaded by the first two steps only

    auto decode = [&](int tile) __attribute__((always_inline)) -> Coord {
        Coord c;
        unsigned r, t = fastdiv((unsigned)tile, dv.n_ot, dv.m_ot, r);
        c.ot = (int)r;
        t = fastdiv(t, dv.tiles_x, dv.m_tx, r);
        c.tx = (int)r;
        c.n = (int)fastdiv(t, dv.tiles_y, dv.m_ty, r);
        c.ty = (int)r;
        return c;
    };

    // ---- what this lane moves in a patch DMA piece: wave w owns pieces 5w .. 5w+4 of the 40 (plane = w / 4, pixel group = 5 (w & 3) + i);
    // lane l of a piece = patch pixel 32 grp + l / 2, 16-byte half (l & 1) ^ bit 3 of the pixel index
    const int ks_dma = wave >> 2;
    unsigned pvoff[5];
    i32x4 srd_x = make_srd(p.x, 0);
    auto tile_addresses = [&](const Coord& c) __attribute__((always_inline)) {
        const int iy0 = c.ty * TH + p.org_y, ix0 = c.tx * TW + p.org_x;
#pragma unroll
        for (int i = 0; i < 5; ++i) {       // (recomputed per tile: five cached (row, column, channel) triples cost 15 registers for the whole kernel)
            const int pp = ((wave & 3) * 5 + i) * 32 + (lane >> 1), py = pp / PW, px = pp - py * PW;
            const int iy = iy0 + py, ix = ix0 + px, ch = ks_dma * 16 + (((lane & 1) ^ ((pp >> 3) & 1)) << 3);
            // (one select, no short-circuit branches: this runs between two steps' multiplies on every wave)
            const bool ok = (pp < npix) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            pvoff[i] = ok ? (unsigned)(((iy * p.W + ix) * p.I + ch) * 2) : OOB;
        }
        srd_x = make_srd(p.x + (long)c.n * p.H * p.W * p.I, (unsigned)(p.H * p.W * p.I * 2));
    };
    const i32x4 srd_w = make_srd(p.w, (unsigned)((long)((p.OB + 3) / 4 * 4) * p.wslots * c16n * 1024));
    // ---- this wave's requests of a step: five patch pieces and five weight pieces (piece 8 i + wave; i = 4 exists for waves 0-3 only: the others
    // send theirs -- every lane out of range, zeros -- to the dump slot, wave 7 uses that slot for the tile's bias when there is one).  The requests
    // are issued one at a time BETWEEN the MFMAs of the running step: a `buffer_load ... lds` costs its wave 100-200 cycles at issue, and with the
    // 80 requests of a step issued together after the barrier the matrix pipe idled 1 700-3 500 cycles per step (tools/ring_trace.py).
    unsigned wsoff[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int pi = i * 8 + wave, t = (pi >> 2) < NT ? (pi >> 2) : 0, ks = (pi >> 1) & 1, m = pi & 1;
        wsoff[i] = (unsigned)(((m * p.wslots + p.tw[t]) * c16n + ks) * 1024);
    }
    const bool w4_ok = 32 + wave < NT * 4;
    auto dma_patch = [&](int i, int stage, unsigned chunk_off) __attribute__((always_inline)) {
        dma16(lds0 + L_P + stage * PSTAGE + (ks_dma * 20 + (wave & 3) * 5 + i) * 1024, pvoff[i], srd_x, chunk_off);
    };
    auto dma_weight = [&](int i, int stage, unsigned tile_off) __attribute__((always_inline)) {
        if (i < 4) dma16(lds0 + L_W + stage * WSTAGE + (i * 8 + wave) * 1024, (unsigned)(lane * 16), srd_w, wsoff[i] + tile_off);
        else if (w4_ok) dma16(lds0 + L_W + stage * WSTAGE + (32 + wave) * 1024, (unsigned)(lane * 16), srd_w, wsoff[4] + tile_off);
    };
    // a tile's parameters, one request each from the waves without a fifth weight piece: wave 7 the 64 bias values of the channel tile, wave 6 its
    // 64 out_scale values of sample n (beyond the tensor: zeros; those channels are never stored), waves 4 / 5 noise rows 0-7 / 8-15 of the
    // tile (lane = row l / 8, columns 4 (l % 8) ..; OW % 4 == 0 keeps a piece inside or outside its row as a whole)
    auto dma_params = [&](const Coord& c, int tpar) __attribute__((always_inline)) {
        const unsigned dst = lds0 + L_PRM + tpar * PRM;       // (descriptors are built here, once per tile: held for the whole kernel they cost 12 scalar registers)
        if (wave == 7) { if (p.bias) dma16(dst, lane < 16 ? (unsigned)(c.ot * 256 + lane * 16) : OOB, make_srd(p.bias, (unsigned)(p.O * 4)), 0u); }
        else if (wave == 6) {
            if (p.out_scale) dma16(dst + PRM_SCALE, lane < 16 ? (unsigned)((c.n * p.O + c.ot * 64) * 4 + lane * 16) : OOB, make_srd(p.out_scale, (unsigned)((long)p.N * p.O * 4)), 0u);
        } else if (wave >= 4 && p.noise_mode) {
            const int gy = c.ty * TH + (wave - 4) * 8 + (lane >> 3), gx = c.tx * TW + (lane & 7) * 4;
            const unsigned img = p.noise_mode == 2 ? (unsigned)c.n * (unsigned)(p.OHt * p.OWt) : 0u;
            dma16(dst + PRM_NOISE + (wave - 4) * 1024, (gy < p.OHt && gx < p.OWt) ? (img + (unsigned)(gy * p.OWt + gx)) * 4u : OOB,
                  make_srd(p.noise, (unsigned)((long)(p.noise_mode == 2 ? p.N : 1) * p.OHt * p.OWt * 4)), 0u);
        }
    };

    // ---- B-operand addresses inside a patch stage (plane 0): lane (j, kg), tap t, pixel block q -> pixel p, half kg ^ bit 3 of p
    // (NT = 9 is conv2d_f16_impl's 3x3 table: tap t reads patch row t / 3, column t % 3 -- q + row has four values, 12 addresses instead of 18)
    constexpr int BR = NT == 9 ? 4 : NT * 2, BC = NT == 9 ? 3 : 1;
    unsigned baddr_[BR][BC];
#pragma unroll
    for (int r = 0; r < BR; ++r)
#pragma unroll
        for (int c = 0; c < BC; ++c) {
            const int pp = NT == 9 ? (wave * 2 + r) * PW + j + c : (wave * 2 + (r & 1) + p.tdy[r >> 1]) * PW + j + p.tdx[r >> 1];
            baddr_[r][c] = (unsigned)(pp * 32 + ((kg ^ ((pp >> 3) & 1)) << 4));
        }
    auto baddr = [&](int t, int q) __attribute__((always_inline)) -> unsigned { return NT == 9 ? baddr_[q + t / 3][t % 3] : baddr_[t * 2 + q][0]; };

    f16x acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;

    // ---- the finished tile waiting for its stores: fp16 channel runs (4 halves) of [channel block][pixel block][row group]; the eight 16-byte
    // stores of a wave go out between the MFMAs of the NEXT step (buffer stores: a lane outside the tensor is dropped by the range check)
    u32x2 pk[2][2][4];
    unsigned pyoff[2] = {OOB, OOB};
    int p_ot = 0;
    i32x4 srd_y = make_srd(p.y, 0);
    bool pend = false;
    auto pend_addresses = [&](const Coord& c) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int gy = c.ty * TH + wave * 2 + q, gx = c.tx * TW + j;
            const int oy = gy * p.s_out + p.oy0, ox = gx * p.s_out + p.ox0;
            const bool ok = (gy < p.GH) & (gx < p.GW) & ((unsigned)oy < (unsigned)p.OHt) & ((unsigned)ox < (unsigned)p.OWt);
            pyoff[q] = ok ? (unsigned)((oy * p.OWt + ox) * p.O * 2) : OOB;
        }
        p_ot = c.ot;
        srd_y = make_srd(p.y + (long)c.n * p.OHt * p.OWt * p.O, (unsigned)(p.OHt * p.OWt * p.O * 2));
    };
    auto store_piece = [&](int k) __attribute__((always_inline)) {          // k = (q, m, gp)
        const int q = k >> 2, m = (k >> 1) & 1, gp = k & 1;
        // lanes 0-31 hold channels 8g .. 8g+3 of group g, lanes 32-63 channels 8g+4 .. 8g+7: after the half exchange the lower lanes own the
        // 16 bytes of group 2 gp, the upper lanes those of group 2 gp + 1
        u32x2 a = pk[m][q][2 * gp], b = pk[m][q][2 * gp + 1];
        auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
        u32x4 v;
        v[0] = r0[0]; v[1] = r1[0]; v[2] = r0[1]; v[3] = r1[1];
        const int o = p_ot * 64 + m * 32 + (2 * gp + kg) * 8;
        const unsigned voff = o < p.O ? pyoff[q] + (unsigned)(o * 2) : OOB;
        asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(srd_y));     // (the wait state hipcc pads behind a > 8-byte store whose data registers are rewritten next: not modelled inside asm)
    };

    // ---- one step: 18 k-iterations (tap, k-step) of four MFMAs; the operands of iteration it + 1 are read while iteration it multiplies (one
    // ds_read_b128 behind each MFMA, pinned by sched_group_barrier); requests of the next step and stores of the previous tile in between
    auto body = [&](auto w_tag, auto p_tag, int stage, unsigned chunk_off, unsigned tile_off, bool prm_next, const Coord& b_c, int b_tpar) __attribute__((always_inline)) {
        constexpr bool WITH_W = decltype(w_tag)::value, PEND = decltype(p_tag)::value;
        const unsigned char* wa = lds + L_W + stage * WSTAGE + lane * 16;
        const unsigned char* pa = lds + L_P + stage * PSTAGE;
        const int nstage = stage ^ 1;
        h8 a[2][2], b[2][2];
        a[0][0] = *(const h8*)(wa); b[0][0] = *(const h8*)(pa + baddr(0, 0));
        b[0][1] = *(const h8*)(pa + baddr(0, 1)); a[0][1] = *(const h8*)(wa + 1024);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int it = 0; it < NT * 2; ++it) {
            const int cb = it & 1, nb = cb ^ 1;
            if (it + 1 < NT * 2) {
                const int t = (it + 1) >> 1, ks = (it + 1) & 1;
                a[nb][0] = *(const h8*)(wa + ((it + 1) * 2 + 0) * 1024); b[nb][0] = *(const h8*)(pa + ks * PLANE + baddr(t, 0));
                b[nb][1] = *(const h8*)(pa + ks * PLANE + baddr(t, 1)); a[nb][1] = *(const h8*)(wa + ((it + 1) * 2 + 1) * 1024);
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cb][0], b[cb][0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cb][0], b[cb][1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cb][1], b[cb][0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cb][1], b[cb][1], acc[1][1], 0, 0, 0);
            if (it + 1 < NT * 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // slots (NT = 9: 18 iterations): patch pieces behind iterations 0 4 8 12 16, weight pieces behind 2 6 10 14 17, stores behind 1 3 .. 15
            constexpr int LAST = NT * 2 - 1;
            if (it % 4 == 0 && it / 4 < 5 && it / 4 * 4 <= LAST) dma_patch(it / 4, nstage, chunk_off);
            if (NT * 2 < 17 && it == LAST)
                for (int i = LAST / 4 + 1; i < 5; ++i) dma_patch(i, nstage, chunk_off);
            if constexpr (WITH_W) {
                if (it % 4 == 2 && it / 4 < 4) dma_weight(it / 4, nstage, tile_off);
                if (it == LAST) {
                    for (int i = (LAST >= 2 ? (LAST - 2) / 4 + 1 : 0); i < 4; ++i) dma_weight(i, nstage, tile_off);
                    dma_weight(4, nstage, tile_off);
                }
            }
            if (it == LAST && prm_next) dma_params(b_c, b_tpar);
            if constexpr (PEND) {
                if ((it & 1) && it / 2 < 8) store_piece(it / 2);
                if (NT * 2 < 16 && it == LAST)
                    for (int k = (LAST + 1) / 2; k < 8; ++k) store_piece(k);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;

    int tile = blockIdx.x;
    Coord cur = decode(tile);
    tile_addresses(cur);
    int chunk = 0, s = 0, tpar = 0;
    const bool prm_any = p.bias || p.tail, prm_per_tile = prm_any && (!resident || p.tail);
    {   // requests of step 0 (and, with resident weights and no tail, the bias of the one channel tile into both parameter slots)
#pragma unroll
        for (int i = 0; i < 5; ++i) dma_patch(i, 0, 0u);
        const unsigned tile_off = (unsigned)(cur.ot * 2 * p.wslots * c16n * 1024);
#pragma unroll
        for (int i = 0; i < 5; ++i) dma_weight(i, 0, tile_off);
        if (prm_any) { dma_params(cur, 0); if (!prm_per_tile) dma_params(cur, 1); }
    }
#ifndef SHG_RING_NO_PRIO
    // the second-dispatched half of the workgroup loses the issue arbitration of its SIMD to the older wave on every step (wave 7 multiplies
    // 5.5 k cycles beside wave 0's 3.6 k, and the step ends with the slowest wave): one static priority for that half, no per-phase flips
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    while (true) {
        // step s is in LDS once every wave's requests have landed; the same barrier says every wave has finished reading stage (s+1) & 1
        RING_TRACE(0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        RING_TRACE(1);
        int nchunk = chunk + 1, ntile = tile;
        if (nchunk == nchunks) { nchunk = 0; ntile = tile + gridDim.x; }
        const bool has_next = ntile < ntiles;
        Coord nc = cur;
        if (!has_next) { nchunk = chunk; ntile = tile; }             // (the last step requests itself once more: no special case in the body)
        else if (nchunk == 0) { nc = decode(ntile); tile_addresses(nc); }
        const unsigned chunk_off = (unsigned)(nchunk * 64), tile_off = (unsigned)((nc.ot * 2 * p.wslots * c16n + nchunk * 2) * 1024);
        const bool with_w = !(resident && s + 1 >= 2);
        const bool prm_next = prm_per_tile && nchunk == 0 && has_next;
        const int b_tpar = tpar ^ 1;
        RING_TRACE(2);
        if (with_w) {
            if (pend) body(T_{}, T_{}, s & 1, chunk_off, tile_off, prm_next, nc, b_tpar);
            else body(T_{}, F_{}, s & 1, chunk_off, tile_off, prm_next, nc, b_tpar);
        } else {
            if (pend) body(F_{}, T_{}, s & 1, chunk_off, tile_off, prm_next, nc, b_tpar);
            else body(F_{}, F_{}, s & 1, chunk_off, tile_off, prm_next, nc, b_tpar);
        }
        pend = false;
        if (chunk == nchunks - 1) {
            // C/D layout: column (pixel) = lane & 31, row (channel) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
            const unsigned char* prm = lds + L_PRM + tpar * PRM;
            if (!p.tail) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f4 bv = {0.f, 0.f, 0.f, 0.f};
                        if (p.bias) bv = *(const f4*)(prm + (m * 32 + g * 8 + kg * 4) * 4);
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            h4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = (_Float16)(acc[m][q][g * 4 + e] + bv[e]);
                            pk[m][q][g] = __builtin_bit_cast(u32x2, v);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[m][q][g * 4 + e] = 0.f;
                        }
                    }
            } else {
                // conv_f16_kernel's tail on the half-rounded result: z = half(acc) * d + noise * strength + bias -> lrelu_agc (or * gain) -> half
                float nz[2] = {0.f, 0.f};
                if (p.noise_mode)
#pragma unroll
                    for (int q = 0; q < 2; ++q) nz[q] = *(const float*)(prm + PRM_NOISE + ((wave * 2 + q) * 32 + j) * 4) * p.noise_strength;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f4 bv = {0.f, 0.f, 0.f, 0.f}, dv = {1.f, 1.f, 1.f, 1.f};
                        if (p.bias) bv = *(const f4*)(prm + (m * 32 + g * 8 + kg * 4) * 4);
                        if (p.out_scale) dv = *(const f4*)(prm + PRM_SCALE + (m * 32 + g * 8 + kg * 4) * 4);
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            h4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float z = __builtin_fmaf((float)(_Float16)acc[m][q][g * 4 + e], dv[e], nz[q]) + bv[e];
                                z = p.act ? shg_lrelu_agc(z, p.alpha, p.gain, p.clamp) : z * p.gain;
                                v[e] = (_Float16)z;
                            }
                            pk[m][q][g] = __builtin_bit_cast(u32x2, v);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[m][q][g * 4 + e] = 0.f;
                        }
                    }
            }
            pend_addresses(cur);
            pend = true;
        }
        RING_TRACE(3);
        if (!has_next) break;
        if (nchunk == 0) { tile = ntile; cur = nc; tpar ^= 1; }
        chunk = nchunk;
        ++s;
    }
    if (pend)
#pragma unroll
        for (int k = 0; k < 8; ++k) store_piece(k);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (no request may outlive the workgroup: the last step's self-request, the stores)
}

}  // namespace ring

// The ring kernel serves: patches read with stride 1 (stride-1 convolutions; span = extent of the tap offsets) of at most 18 x 34 pixels,
// 9 taps (for now), whole 8-channel output pieces; of the fused tail everything but the input scale and the residual (those stay on
// conv_f16_kernel), noise with whole 16-byte pieces per row.
static int g_f16_routes = 7;         // bit 0: stride-1 3x3 launches on the ring kernel, bit 1: transposed launches on the merged-phase kernel, bit 2: stride-2 3x3 launches on conv_f16_down.hip
int conv_f16_routes() { return g_f16_routes; }

bool conv_ring_eligible(const ConvP& p, int span_y, int span_x) {
#ifdef SHG_F16_NO_RING
    return false;
#else
    if (!(g_f16_routes & 1)) return false;
    if (p.s_in != 1 || p.ntaps != 9 || (p.O & 7) || (p.I & 31) || p.in_scale || p.residual) return false;
    if (p.tail && (p.s_out != 1 || p.oy0 || p.ox0)) return false;
    if (p.noise_mode && ((p.OWt & 3) || (reinterpret_cast<uintptr_t>(p.noise) & 15))) return false;
    if ((reinterpret_cast<uintptr_t>(p.bias) | reinterpret_cast<uintptr_t>(p.out_scale)) & 15) return false;
    return span_y == 3 && span_x == 3;                             // (the kernel's 18 x 34 patch)
#endif
}

int conv_ring_launch(const ConvP& p0, int span_y, int span_x, hipStream_t st) {
    ConvP p = p0;
    p.PH = ring::TH - 1 + span_y;
    p.PW = ring::TW - 1 + span_x;
    p.tiles_y = shg_cdiv(p.GH, ring::TH);
    p.tiles_x = shg_cdiv(p.GW, ring::TW);
    const int n_ot = (p.O + 63) / 64;
    const long ntiles = (long)p.N * p.tiles_y * p.tiles_x * n_ot;
    const int cus = shg_cu_count();
    static ShgDeviceOnce attr_once;
    const int dev_now = shg_current_device();
    if (attr_once.pending(dev_now)) {
        if (hipFuncSetAttribute((const void*)ring::conv_f16_ring_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize, ring::LDS_BYTES) != hipSuccess) {
            shg_set_error("conv2d_f16 (ring): cannot reserve %d bytes of LDS", ring::LDS_BYTES);
            return SHG_ERR_LAUNCH;
        }
        attr_once.mark(dev_now);
    }
    const unsigned grid = (unsigned)(ntiles < cus ? ntiles : cus);
    const ring::RingDiv dv{(unsigned)n_ot, (unsigned)p.tiles_x, (unsigned)p.tiles_y, 0xFFFFFFFFu / (unsigned)n_ot, 0xFFFFFFFFu / (unsigned)p.tiles_x,
                           0xFFFFFFFFu / (unsigned)p.tiles_y};
    hipLaunchKernelGGL((ring::conv_f16_ring_kernel<9>), dim3(grid), dim3(512), ring::LDS_BYTES, st, p, (int)ntiles, dv);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

}  // namespace f16

// Which kernels serve the fp16 convolutions: bit 0 = persistent ring kernel for the stride-1 3x3 launches, bit 1 = merged-phase kernel for the
// stride-2 transposed launches, bit 2 = persistent kernel for the stride-2 3x3 launches (default 7); 0 sends everything to the gather kernel
// conv_f16_kernel.  Routes 0 / 1 give the gather kernel's bits, route 2 its products in another summation order
// (tests/test_gpu_fp16_routes.py); the switch exists for that comparison and for A/B timing.  Returns the old mask.
extern "C" int shg_conv2d_f16_set_routes(int mask) {
    const int old = f16::g_f16_routes;
    f16::g_f16_routes = mask & 7;
    return old;
}

