// fp16 stride-2 3x3 convolution as a persistent LDS-DMA kernel -- the down layers of the encoder and of the critic (after their low-pass:
// conv2d_resample.py:116-120 on the cuDNN half kernels of the `use_fp16` blocks, stylegan.py:660-667, comodgan.py:40-47).  gfx950 only.
//
// The gather kernel conv_f16_kernel<4, 9, 1> serves these launches at 0.18 of the fp16 peak: an 8 x 16-pixel tile per 256-thread workgroup,
// every wave loading every weight operand from L2, a prologue and an LDS-transposing epilogue per tile.  Here (structure of
// conv_f16_ring.hip / conv_f16_upring.hip): one persistent 512-thread workgroup per CU walks tiles of 8 x 32 OUTPUT pixels x 128 output
// channels; a step = one 16-channel k-step of a tile: the 17 x 65 input window (32 B per pixel) and the 36 KiB weight slab (9 taps x 4
// channel blocks in MFMA operand order) of step s + 1 arrive by `buffer_load ... lds` while step s multiplies 36 MFMAs per wave (wave = one
// output row, four channel blocks) from 9 patch operands and 36 weight operands.  The window is stored with its columns de-interleaved (even
// columns, then odd columns of a row) on the GLOBAL side of the DMA, so that the stride-2 operand reads of a tap are 32 consecutive pixels,
// with conv_f16_ring.hip's half swizzle (conflict-free ds_read_b128 for every tap).
// Same products as the gather kernel, summed k-step-major (its order is 32-channel chunk -> tap -> k-step): results agree to fp32 summation
// order -- NOT bit-identical, unlike the two other persistent kernels (tests/test_gpu_fp16_routes.py holds it against float64 instead).
// The fused layer tail (bias, out_scale, lrelu_agc; no noise / residual / input scale) is applied to the accumulators in registers.
// LDS (bytes): [0, 73 728) two weight stages | [73 728, 155 648) two patch stages of 40 pieces x 1 KiB | 2 x 2 KiB tile parameters.
#include <type_traits>
#include "shg_common.h"
#include "conv_f16_p.h"

namespace f16 {
namespace down {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 32;                    // output pixels of a tile: wave w owns row w
constexpr int PR = 2 * TH + 1, PC = 2 * TW + 1;   // window: 17 rows x 65 columns
constexpr int ROWS = PC + 1;                      // slots of a window row: 33 even columns, 33 slots for the 32 odd ones (66)
constexpr int PSTAGE = 40 * 1024, WSTAGE = 36 * 1024;
constexpr int L_W = 0, L_P = 2 * WSTAGE, L_PRM = L_P + 2 * PSTAGE, LDS_BYTES = L_PRM + 2 * 2048;
constexpr unsigned OOB = 0x80000000u;

struct DownP {
    ConvP c;
    int tiles_x, tiles_y, n_ot, ntiles, pad;
    unsigned m_ot, m_tx, m_ty;
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

__global__ __launch_bounds__(512) void conv_f16_down_kernel(const DownP P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const ConvP& p = P.c;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int nsteps = p.I >> 4, c16n = p.I >> 4;

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

    // ---- window requests: wave w owns pieces 5 w .. 5 w + 4 of the 40; lane l of a piece = slot 32 piece + l / 2, 16-byte half (l & 1) ^ bit 3
    // of the slot index; slot = row * 66 + (column & 1) * 33 + (column >> 1): input pixel (16 ty - pad + row, 64 tx - pad + column)
    unsigned pvoff[5];
    i32x4 srd_x = make_srd(p.x, 0);
    auto tile_addresses = [&](const Coord& c) __attribute__((always_inline)) {
        const int iy0 = c.ty * (2 * TH) - P.pad, ix0 = c.tx * (2 * TW) - P.pad;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int slot = (wave * 5 + i) * 32 + (lane >> 1), row = slot / ROWS, rem = slot - row * ROWS;
            const int odd = rem >= 33 ? 1 : 0, col = 2 * (rem - 33 * odd) + odd;
            const int iy = iy0 + row, ix = ix0 + col, ch = ((lane & 1) ^ ((slot >> 3) & 1)) << 3;
            const bool ok = (row < PR) & (col < PC) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            pvoff[i] = ok ? (unsigned)(((iy * p.W + ix) * p.I + ch) * 2) : OOB;
        }
        srd_x = make_srd(p.x + (long)c.n * p.H * p.W * p.I, (unsigned)(p.H * p.W * p.I * 2));
    };
    const i32x4 srd_w = make_srd(p.w, (unsigned)((long)((p.OB + 3) / 4 * 4) * 9 * c16n * 1024));
    auto dma_patch = [&](int i, int stage, unsigned step_off) __attribute__((always_inline)) {
        dma16(lds0 + L_P + stage * PSTAGE + (wave * 5 + i) * 1024, pvoff[i], srd_x, step_off);
    };
    // weight pieces of a step: piece = tap * 4 + channel block (36); wave w issues pieces w, w + 8, w + 16, w + 24 and (waves 0-3) w + 32
    auto dma_weight = [&](int i, int stage, unsigned tile_off) __attribute__((always_inline)) {
        const int pi = i * 8 + wave;
        if (pi < 36) dma16(lds0 + L_W + stage * WSTAGE + pi * 1024, (unsigned)(lane * 16), srd_w,
                           (unsigned)(((pi & 3) * 9 + (pi >> 2)) * c16n * 1024) + tile_off);
    };
    // a tile's parameters (waves 6 / 7): 128 bias values | 128 out_scale values of sample n (each its own KiB)
    auto dma_params = [&](const Coord& c, int tpar) __attribute__((always_inline)) {
        const unsigned dst = lds0 + L_PRM + tpar * 2048;
        if (wave == 7) { if (p.bias) dma16(dst, lane < 32 ? (unsigned)(c.ot * 512 + lane * 16) : OOB, make_srd(p.bias, (unsigned)(p.O * 4)), 0u); }
        else if (wave == 6) {
            if (p.out_scale) dma16(dst + 1024, lane < 32 ? (unsigned)((c.n * p.O + c.ot * 128) * 4 + lane * 16) : OOB, make_srd(p.out_scale, (unsigned)((long)p.N * p.O * 4)), 0u);
        }
    };

    // ---- patch operand addresses: lane (j, kg), tap (ky, kx): input pixel (2 wave + ky, 2 j + kx) = slot (2 wave + ky) * 66 + (kx & 1) * 33 + j + (kx >> 1)
    unsigned baddr[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int ky = t / 3, kx = t % 3, slot = (2 * wave + ky) * ROWS + (kx & 1) * 33 + j + (kx >> 1);
        baddr[t] = (unsigned)(slot * 32 + ((kg ^ ((slot >> 3) & 1)) << 4));
    }

    f16x acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    const bool prm_any = p.bias || p.out_scale;
    int tile = blockIdx.x;
    Coord cur = decode(tile);
    tile_addresses(cur);
    int step = 0, s = 0, tpar = 0;
    {
#pragma unroll
        for (int i = 0; i < 5; ++i) dma_patch(i, 0, 0u);
        const unsigned tile_off = (unsigned)(cur.ot * 4 * 9 * c16n * 1024);
#pragma unroll
        for (int i = 0; i < 5; ++i) dma_weight(i, 0, tile_off);
        if (prm_any) dma_params(cur, 0);
    }
    while (true) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        int nstep = step + 1, ntile = tile;
        if (nstep == nsteps) { nstep = 0; ntile = tile + gridDim.x; }
        const bool has_next = ntile < P.ntiles;
        Coord nc = cur;
        if (!has_next) { nstep = step; ntile = tile; }
        else if (nstep == 0) { nc = decode(ntile); tile_addresses(nc); }
        const unsigned step_off = (unsigned)(nstep * 32), tile_off = (unsigned)((nc.ot * 4 * 9 * c16n + nstep) * 1024);
        const int stage = s & 1, nstage = stage ^ 1;
        const unsigned char* wa = lds + L_W + stage * WSTAGE + lane * 16;
        const unsigned char* pa = lds + L_P + stage * PSTAGE;
        h8 b[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) b[t] = *(const h8*)(pa + baddr[t]);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const h8 a = *(const h8*)(wa + (t * 4 + m) * 1024);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[t], acc[m], 0, 0, 0);
                // requests of the next step between the products: the five window pieces behind taps 0-4, the weight pieces behind 4-8
                if (m == 1 && t < 5) dma_patch(t, nstage, step_off);
                if (m == 3 && t >= 4) dma_weight(t - 4, nstage, tile_off);
            }
        }
        if (prm_any && nstep == 0 && has_next) dma_params(nc, tpar ^ 1);
        if (step == nsteps - 1) {
            // C/D layout: column (pixel) = lane & 31, row (channel) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
            const unsigned char* prm = lds + L_PRM + tpar * 2048;
            const i32x4 srd_y = make_srd(p.y + (long)cur.n * p.OHt * p.OWt * p.O, (unsigned)(p.OHt * p.OWt * p.O * 2));
            const int oy = cur.ty * TH + wave, ox = cur.tx * TW + j;
            const bool ok = (oy < p.OHt) & (ox < p.OWt);
            const unsigned pix = ok ? (unsigned)((oy * p.OWt + ox) * p.O * 2) : OOB;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    h4 v[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int g = 2 * gp + h;
                        f4 bv = {0.f, 0.f, 0.f, 0.f}, dv = {1.f, 1.f, 1.f, 1.f};
                        if (p.bias) bv = *(const f4*)(prm + (m * 32 + g * 8 + kg * 4) * 4);
                        if (p.out_scale) dv = *(const f4*)(prm + 1024 + (m * 32 + g * 8 + kg * 4) * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (!p.tail) v[h][e] = (_Float16)(acc[m][g * 4 + e] + bv[e]);
                            else {
                                float z = __builtin_fmaf((float)(_Float16)acc[m][g * 4 + e], dv[e], 0.f) + bv[e];
                                z = p.act ? shg_lrelu_agc(z, p.alpha, p.gain, p.clamp) : z * p.gain;
                                v[h][e] = (_Float16)z;
                            }
                        }
                    }
                    const u32x2 a2 = __builtin_bit_cast(u32x2, v[0]), b2 = __builtin_bit_cast(u32x2, v[1]);
                    auto r0 = __builtin_amdgcn_permlane32_swap(a2[0], b2[0], false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(a2[1], b2[1], false, false);
                    u32x4 o4;
                    o4[0] = r0[0]; o4[1] = r1[0]; o4[2] = r0[1]; o4[3] = r1[1];
                    const int o = cur.ot * 128 + m * 32 + (2 * gp + kg) * 8;
                    const unsigned voff = (ok && o < p.O) ? pix + (unsigned)(o * 2) : OOB;
                    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(o4), "v"(voff), "s"(srd_y));
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
            }
        }
        if (!has_next) break;
        if (nstep == 0) { tile = ntile; cur = nc; tpar ^= 1; }
        step = nstep;
        ++s;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace down

// Stride-2 3x3 launches with whole 16-channel k-steps and whole 8-channel output pieces; of the fused tail bias, out_scale and the activation
// (noise / residual / input scale stay on the gather kernel).
bool conv_down_eligible(const ConvP& p) {
#ifdef SHG_F16_NO_DOWN
    return false;
#else
    if (!(conv_f16_routes() & 4)) return false;
    if (p.s_in != 2 || p.s_out != 1 || p.ntaps != 9 || (p.I & 15) || (p.O & 7) || p.in_scale || p.residual || p.noise_mode) return false;
    // thin layers only: a step is one 16-channel k-step (36 MFMAs per wave against ten DMA requests and a barrier), which pays while the layer is
    // HBM-bound -- measured at batch 8 with the fused tail: 64 -> 128 at 513^2 229 -> 170 us, 128 -> 256 at 257^2 157 -> 137 us, but
    // 256 -> 512 at 129^2 136 -> 153 us and 512 -> 512 at 65^2 68 -> 72 us (the gather kernel keeps those)
    if (p.I > 128) return false;
    if ((reinterpret_cast<uintptr_t>(p.bias) | reinterpret_cast<uintptr_t>(p.out_scale)) & 15) return false;
    return (long)p.H * p.W * p.I * 2 < (1L << 31) && (long)p.OHt * p.OWt * p.O * 2 < (1L << 31);
#endif
}

int conv_down_launch(const ConvP& p0, int pad, hipStream_t st) {
    down::DownP P{};
    P.c = p0;
    P.pad = pad;
    P.tiles_y = shg_cdiv(p0.OHt, down::TH);
    P.tiles_x = shg_cdiv(p0.OWt, down::TW);
    P.n_ot = (p0.O + 127) / 128;
    const long ntiles = (long)p0.N * P.tiles_y * P.tiles_x * P.n_ot;
    if (ntiles > 0x7fffffffL) { shg_set_error("conv2d_f16 (stride 2): too many tiles"); return SHG_ERR_ARG; }
    P.ntiles = (int)ntiles;
    P.m_ot = 0xFFFFFFFFu / (unsigned)P.n_ot; P.m_tx = 0xFFFFFFFFu / (unsigned)P.tiles_x; P.m_ty = 0xFFFFFFFFu / (unsigned)P.tiles_y;
    const int cus = shg_cu_count();
    static ShgDeviceOnce attr_once;
    const int dev_now = shg_current_device();
    if (attr_once.pending(dev_now)) {
        if (hipFuncSetAttribute((const void*)down::conv_f16_down_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, down::LDS_BYTES) != hipSuccess) {
            shg_set_error("conv2d_f16 (stride 2): cannot reserve %d bytes of LDS", down::LDS_BYTES);
            return SHG_ERR_LAUNCH;
        }
        attr_once.mark(dev_now);
    }
    const unsigned grid = (unsigned)(ntiles < cus ? ntiles : cus);
    hipLaunchKernelGGL(down::conv_f16_down_kernel, dim3(grid), dim3(512), down::LDS_BYTES, st, P);
    SHG_CHECK_LAUNCH();
    return SHG_OK;
}

}  // namespace f16
