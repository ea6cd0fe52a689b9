// Launch parameters of the fp16 gather convolutions (conv_f16.hip: conv_f16_kernel; conv_f16_ring.hip: the persistent LDS-DMA kernel for
// the stride-1-read forms).  gfx950 only.
#pragma once
#include "shg_common.h"

namespace f16 {

struct ConvP {
    const _Float16* x;
    const _Float16* w;               // MFMA operand order [OB][wslots][I/16][64 lanes][8]: element = W[slot][ob*32 + (lane & 31)][c16*16 + (lane >> 5)*8 + e]
    int OB, wslots;                  // 32-channel output blocks (O rounded up), tap slots of the weight tensor
    const float* bias;               // optional [O]
    _Float16* y;
    int N, I, O, H, W;               // input tensor
    int OHt, OWt;                    // output tensor extent
    int GH, GW;                      // extent of the computed pixel grid (oy', ox')
    int tiles_x, tiles_y;
    int s_in, s_out, oy0, ox0;
    int ntaps;
    int tdy[9], tdx[9], tw[9];       // input offset of tap t (already minus the patch origin) and its weight slot
    int org_y, org_x;                // patch origin: input row of patch row 0 for grid row 0 = org_y
    int PH, PW;                      // patch extent
    int wlds_off;                    // halves: start of the staged weight slab behind the patch / output tile (WLDS kernels)
    // fused layer tail of the inference route (all optional): x * in_scale[n,i] while the patch is staged; then
    // y = A(conv * out_scale[n,o] + noise * noise_strength + bias[o]) + residual in the store pass
    const float* in_scale; const float* out_scale; const float* noise; const _Float16* residual;
    int noise_mode, act, tail;       // noise: 0 none, 1 [OH,OW], 2 [N,OH,OW]; tail: any of the epilogue operands present
    float noise_strength, alpha, gain, clamp;
};

// conv_f16_ring.hip: true when the persistent ring kernel serves this launch (then it has been enqueued on `st`)
// (span = extent of the tap offsets: 3 for a 3x3 kernel; p carries the tap table, the patch origin and the pixel grid, not yet a tiling)
bool conv_ring_eligible(const ConvP& p, int span_y, int span_x);
int conv_ring_launch(const ConvP& p, int span_y, int span_x, hipStream_t st);

// conv_f16_upring.hip: the stride-2 transposed 3x3 form (mode 1 of conv2d_f16_impl) with all four phases in one persistent launch
int conv_f16_routes();          // conv_f16_ring.hip: the mask of shg_conv2d_f16_set_routes
bool convt_upring_eligible(const ConvP& p);
int convt_upring_launch(const ConvP& p, int crop, hipStream_t st);

// conv_f16_down.hip: the stride-2 3x3 launches (p.s_in == 2, nine taps) on a persistent kernel; pad = the convolution's padding
bool conv_down_eligible(const ConvP& p);
int conv_down_launch(const ConvP& p, int pad, hipStream_t st);

}  // namespace f16
