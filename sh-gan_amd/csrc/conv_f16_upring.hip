// fp16 stride-2 transposed 3x3 convolution, all four sub-pixel phases in ONE persistent LDS-DMA kernel -- the up layers of the synthesis
// network and the input gradients of the stride-2 layers (reference: the cuDNN half kernels behind conv2d_resample.py:125-137 /
// conv2d_gradfix.py:118-135 in the `use_fp16` blocks, stylegan.py:486,660-667).  gfx950 only.
//
// conv_f16.hip launches the gather kernel once per phase (1 / 2 / 2 / 4 taps): four passes over the input, phases of one or two taps that
// are all prologue and epilogue (profiles/r05_f16_pmc_base_summary.txt: matrix pipe busy 0.08-0.28), 0.15 of the fp16 peak as a class.
// Here a step = one 32-channel chunk of a tile of 16 x 32 GRID pixels (oy', ox'): the 17 x 33 input patch is fetched once and feeds all nine
// taps -- tap (ky, kx) belongs to phase (ky & 1, kx & 1) and reads the grid pixel shifted by (-(ky >> 1), -(kx >> 1)) -- into four
// accumulator sets (phase x two pixel rows x 32 output channels per wave): 36 MFMAs per wave and step from 6 patch operands and 9 weight
// operands per k-step.  Same arithmetic as the per-phase launches in the same order per accumulator (chunk -> tap -> k-step, fp32
// accumulation, one rounding): bit-identical results.
// Structure as conv_f16_ring.hip: one 512-thread workgroup per CU walks its tiles; patch (two k-step planes, half-swizzled on the global side)
// and weight slab (18 KiB in MFMA operand order) of step s + 1 arrive by `buffer_load ... lds` while step s multiplies; `x * in_scale` of
// the inference route (half x half, stylegan.py:173) is applied to the patch operands in registers; a finished tile leaves as 16-byte
// channel runs through v_permlane32_swap.
// LDS (bytes): [0, 36 864) two weight stages | [36 864, 118 784) two patch stages of 2 planes x 640 pixels x 32 B | 2 x 2 KiB in_scale rows.
#include <type_traits>
#include "shg_common.h"
#include "conv_f16_p.h"

namespace f16 {
namespace upring {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 16, TW = 32, PW = TW + 1;       // grid pixels of a tile; patch = (TH + 1) x (TW + 1) input pixels
constexpr int PPX = 640, PLANE = PPX * 32, PSTAGE = 2 * PLANE, WSTAGE = 18 * 1024;
constexpr int L_W = 0, L_P = 2 * WSTAGE, L_S = L_P + 2 * PSTAGE, LDS_BYTES = L_S + 2 * 2048;
constexpr unsigned OOB = 0x80000000u;

struct UpP {
    ConvP c;                    // x, w, y, N, I, O, H, W, OHt, OWt, in_scale
    int crop, tiles_x, tiles_y, n_ot, ntiles;
    unsigned m_ot, m_tx, m_ty;  // floor((2^32 - 1) / divisor) of the tile decode
};

__device__ __forceinline__ i32x4 make_srd(const void* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    i32x4 s;
    s[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    s[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    s[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    s[3] = 0x00020000;
    return s;
}
__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned voff, i32x4 srd, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(srd), "s"(soff));
}
__device__ __forceinline__ unsigned fastdiv(unsigned n, unsigned d, unsigned m, unsigned& rem) {
    unsigned q = __umulhi(n, m), r = n - q * d;
    if (r >= d) { ++q; r -= d; }
    rem = r;
    return q;
}

struct Coord { int n, ty, tx, ot; };

__global__ __launch_bounds__(512) void conv_f16_upring_kernel(const UpP P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const ConvP& p = P.c;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int nchunks = p.I >> 5, c16n = p.I >> 4;
    const bool scaled = p.in_scale != nullptr;

    auto decode = [&](int tile) __attribute__((always_inline)) -> Coord {
        Coord c;
        unsigned r, t = fastdiv((unsigned)tile, (unsigned)P.n_ot, P.m_ot, r);
        c.ot = (int)r;
        t = fastdiv(t, (unsigned)P.tiles_x, P.m_tx, r);
        c.tx = (int)r;
        c.n = (int)fastdiv(t, (unsigned)P.tiles_y, P.m_ty, r);
        c.ty = (int)r;
        return c;
    };

    // ---- patch requests: wave w owns pieces 5 (w & 3) .. + 4 of plane w >> 2; lane l of a piece = patch pixel 32 piece + l / 2, 16-byte half
    // (l & 1) ^ bit 3 of the pixel index (the swizzle of conv_f16_ring.hip); patch pixel (r, c) = input pixel (16 ty - 1 + r, 32 tx - 1 + c)
    const int ks_dma = wave >> 2;
    unsigned pvoff[5];
    i32x4 srd_x = make_srd(p.x, 0);
    auto tile_addresses = [&](const Coord& c) __attribute__((always_inline)) {
        const int iy0 = c.ty * TH - 1, ix0 = c.tx * TW - 1;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int pp = ((wave & 3) * 5 + i) * 32 + (lane >> 1), py = pp / PW, px = pp - py * PW;
            const int iy = iy0 + py, ix = ix0 + px, ch = ks_dma * 16 + (((lane & 1) ^ ((pp >> 3) & 1)) << 3);
            const bool ok = (py <= TH) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            pvoff[i] = ok ? (unsigned)(((iy * p.W + ix) * p.I + ch) * 2) : OOB;
        }
        srd_x = make_srd(p.x + (long)c.n * p.H * p.W * p.I, (unsigned)(p.H * p.W * p.I * 2));
    };
    const i32x4 srd_w = make_srd(p.w, (unsigned)((long)((p.OB + 3) / 4 * 4) * 9 * c16n * 1024));
    auto dma_patch = [&](int i, int stage, unsigned chunk_off) __attribute__((always_inline)) {
        dma16(lds0 + L_P + stage * PSTAGE + (ks_dma * 20 + (wave & 3) * 5 + i) * 1024, pvoff[i], srd_x, chunk_off);
    };
    // weight pieces of a step: piece = tap * 2 + k-step (18); wave w issues pieces w, w + 8 and (waves 0, 1) w + 16
    auto dma_weight = [&](int i, int stage, unsigned tile_off) __attribute__((always_inline)) {
        const int pi = i * 8 + wave;
        if (pi < 18) dma16(lds0 + L_W + stage * WSTAGE + pi * 1024, (unsigned)(lane * 16), srd_w, (unsigned)((((pi >> 1) * c16n) + (pi & 1)) * 1024) + tile_off);
    };
    // in_scale row of sample n (I floats, at most 2 KiB): waves 6 / 7 fetch its halves
    auto dma_scale = [&](const Coord& c, int tpar) __attribute__((always_inline)) {
        if (wave >= 6) {
            const int off = (wave - 6) * 1024 + lane * 16;
            dma16(lds0 + L_S + tpar * 2048 + (wave - 6) * 1024, off < p.I * 4 ? (unsigned)(c.n * p.I * 4 + off) : OOB,
                  make_srd(p.in_scale, (unsigned)((long)p.N * p.I * 4)), 0u);
        }
    };

    // ---- patch operand addresses: lane (j, kg) reads grid pixel (2 wave + q + dy, j + dx) = patch pixel (2 wave + q + dy + 1, j + dx + 1);
    // row r = q + dy + 1 in {0, 1, 2}, column c = dx + 1 in {0, 1}: six operands per k-step feed the 18 products
    unsigned baddr[3][2];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int pp = (wave * 2 + r) * PW + j + c;
            baddr[r][c] = (unsigned)(pp * 32 + ((kg ^ ((pp >> 3) & 1)) << 4));
        }

    f16x acc[4][2];                                   // [phase 2 py + px][pixel row q]
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ph][q][r] = 0.f;

    int tile = blockIdx.x;
    Coord cur = decode(tile);
    tile_addresses(cur);
    int chunk = 0, s = 0, tpar = 0;
    {
#pragma unroll
        for (int i = 0; i < 5; ++i) dma_patch(i, 0, 0u);
        const unsigned tile_off = (unsigned)(cur.ot * 9 * c16n * 1024);
#pragma unroll
        for (int i = 0; i < 3; ++i) dma_weight(i, 0, tile_off);
        if (scaled) dma_scale(cur, 0);
    }
    while (true) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // step s landed (stores of the previous tile too); stage (s + 1) & 1 is free
        int nchunk = chunk + 1, ntile = tile;
        if (nchunk == nchunks) { nchunk = 0; ntile = tile + gridDim.x; }
        const bool has_next = ntile < P.ntiles;
        Coord nc = cur;
        if (!has_next) { nchunk = chunk; ntile = tile; }                   // (the last step requests itself once more)
        else if (nchunk == 0) { nc = decode(ntile); tile_addresses(nc); }
        const unsigned chunk_off = (unsigned)(nchunk * 64), tile_off = (unsigned)((nc.ot * 9 * c16n + nchunk * 2) * 1024);
        const int stage = s & 1, nstage = stage ^ 1;
        const unsigned char* wa = lds + L_W + stage * WSTAGE + lane * 16;
        const unsigned char* pa = lds + L_P + stage * PSTAGE;
        // x * in_scale (half x half): this lane's 8 channels of either k-step
        h8 sv[2];
        if (scaled) {
            const unsigned char* sp = lds + L_S + tpar * 2048 + (chunk * 32 + kg * 8) * 4;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const f4 lo = *(const f4*)(sp + ks * 64), hi = *(const f4*)(sp + ks * 64 + 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) { sv[ks][e] = (_Float16)lo[e]; sv[ks][4 + e] = (_Float16)hi[e]; }
            }
        }
        // the twelve patch operands of the step (two k-steps x three rows x two columns), scaled; then tap by tap, k-step by k-step -- the
        // order of conv_f16_kernel's phases (chunk -> tap -> k-step per accumulator)
        h8 b[2][3][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) b[ks][r][c] = *(const h8*)(pa + ks * PLANE + baddr[r][c]);
        if (scaled)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c) b[ks][r][c] = b[ks][r][c] * sv[ks];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t % 3, ph = (ky & 1) * 2 + (kx & 1), dy = -(ky >> 1), dx = -(kx >> 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const h8 a = *(const h8*)(wa + (t * 2 + ks) * 1024);
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[ph][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[ks][q + dy + 1][dx + 1], acc[ph][q], 0, 0, 0);
                // requests of the next step between the products: five patch pieces, then the weight pieces
                const int idx = t * 2 + ks;
                if ((idx & 1) && idx < 16) {
                    const int slot = idx >> 1;
                    if (slot < 5) dma_patch(slot, nstage, chunk_off);
                    else dma_weight(slot - 5, nstage, tile_off);
                }
            }
        }
        if (scaled && nchunk == 0 && has_next) dma_scale(nc, tpar ^ 1);
        if (chunk == nchunks - 1) {
            // C/D layout: column (pixel) = lane & 31, row (channel) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); lanes 0-31 hold channels 8g .. 8g+3 of
            // group g, lanes 32-63 channels 8g+4 .. 8g+7: after the half exchange the lower lanes own the 16 bytes of group 2 gp, the upper
            // lanes those of group 2 gp + 1
            const i32x4 srd_y = make_srd(p.y + (long)cur.n * p.OHt * p.OWt * p.O, (unsigned)(p.OHt * p.OWt * p.O * 2));
#pragma unroll
            for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int py = ph >> 1, px = ph & 1;
                    const int gy = cur.ty * TH + wave * 2 + q, gx = cur.tx * TW + j;
                    const int oy = 2 * gy + py - P.crop, ox = 2 * gx + px - P.crop;
                    const bool ok = (gy < p.H + 1 - py) & (gx < p.W + 1 - px) & ((unsigned)oy < (unsigned)p.OHt) & ((unsigned)ox < (unsigned)p.OWt);
                    const unsigned pix = ok ? (unsigned)((oy * p.OWt + ox) * p.O * 2) : OOB;
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        h4 v0, v1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] = (_Float16)acc[ph][q][(2 * gp) * 4 + e]; v1[e] = (_Float16)acc[ph][q][(2 * gp + 1) * 4 + e]; }
                        const u32x2 a = __builtin_bit_cast(u32x2, v0), b2 = __builtin_bit_cast(u32x2, v1);
                        auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b2[0], false, false);
                        auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b2[1], false, false);
                        u32x4 v;
                        v[0] = r0[0]; v[1] = r1[0]; v[2] = r0[1]; v[3] = r1[1];
                        const int o = cur.ot * 32 + (2 * gp + kg) * 8;
                        const unsigned voff = (ok && o < p.O) ? pix + (unsigned)(o * 2) : OOB;
                        asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(srd_y));     // (the wait state hipcc pads behind a > 8-byte store whose data registers are rewritten next: not modelled inside asm)
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ph][q][r] = 0.f;
                }
        }
        if (!has_next) break;
        if (nchunk == 0) { tile = ntile; cur = nc; tpar ^= 1; }
        chunk = nchunk;
        ++s;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace upring

// The merged kernel serves the stride-2 transposed 3x3 form with whole 32-channel input chunks and whole 8-channel output pieces, no bias
// (the layers that use it add theirs after the FIR), with or without the input scale.
bool convt_upring_eligible(const ConvP& p) {
#ifdef SHG_F16_NO_UPRING
    return false;
#else
    if (!(conv_f16_routes() & 2)) return false;
    return (p.I & 31) == 0 && (p.O & 7) == 0 && !p.bias && !p.tail && p.I <= 512 && ((reinterpret_cast<uintptr_t>(p.in_scale)) & 15) == 0 &&
           (long)p.H * p.W * p.I * 2 < (1L << 31) && (long)p.OHt * p.OWt * p.O * 2 < (1L << 31);
#endif
}

int convt_upring_launch(const ConvP& p0, int crop, hipStream_t st) {
    upring::UpP P{};
    P.c = p0;
    P.crop = crop;
    P.tiles_y = shg_cdiv(p0.H + 1, upring::TH);
    P.tiles_x = shg_cdiv(p0.W + 1, upring::TW);
    P.n_ot = (p0.O + 31) / 32;
    const long ntiles = (long)p0.N * P.tiles_y * P.tiles_x * P.n_ot;
    if (ntiles > 0x7fffffffL) { shg_set_error("conv2d_f16 (transposed): too many tiles"); return SHG_ERR_ARG; }
    P.ntiles = (int)ntiles;
    P.m_ot = 0xFFFFFFFFu / (unsigned)P.n_ot; P.m_tx = 0xFFFFFFFFu / (unsigned)P.tiles_x; P.m_ty = 0xFFFFFFFFu / (unsigned)P.tiles_y;
    const int cus = shg_cu_count();
    static ShgDeviceOnce attr_once;
    const int dev_now = shg_current_device();
    if (attr_once.pending(dev_now)) {
        if (hipFuncSetAttribute((const void*)upring::conv_f16_upring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, upring::LDS_BYTES) != hipSuccess) {
            shg_set_error("conv2d_f16 (transposed): cannot reserve %d bytes of LDS", upring::LDS_BYTES);
            return SHG_ERR_LAUNCH;
        }
        attr_once.mark(dev_now);
    }
    const unsigned grid = (unsigned)(ntiles < cus ? ntiles : cus);
    hipLaunchKernelGGL(upring::conv_f16_upring_kernel, dim3(grid), dim3(512), upring::LDS_BYTES, st, P);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

}  // namespace f16
