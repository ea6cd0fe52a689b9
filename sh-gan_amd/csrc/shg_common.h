// Shared host/device helpers for libshgan_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#define SHG_OK 0
#define SHG_ERR_ARG (-1)      // bad argument (mirrors TORCH_CHECK failures in upfirdn2d.cpp:19-36)
#define SHG_ERR_LAUNCH (-2)   // HIP launch / runtime error (mirrors AT_CUDA_CHECK, upfirdn2d.cpp:92)
#define SHG_ERR_UNSUPPORTED (-3)

extern "C" const char* shg_last_error(void);
void shg_set_error(const char* fmt, ...);

#define SHG_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            shg_set_error(__VA_ARGS__);          \
            return SHG_ERR_ARG;                  \
        }                                        \
    } while (0)

#define SHG_CHECK_LAUNCH()                                                        \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            shg_set_error("%s:%d: HIP launch failed: %s", __FILE__, __LINE__,     \
                          hipGetErrorString(e_));                                 \
            return SHG_ERR_LAUNCH;                                                \
        }                                                                         \
    } while (0)

static inline int shg_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// leaky-relu -> *gain -> clamp  (common/utils.py:135-143); clamp < 0 disables clamping.
__device__ __forceinline__ float shg_lrelu_agc(float v, float alpha, float gain, float clamp) {
    v = v < 0.f ? v * alpha : v;
    v *= gain;
    if (clamp >= 0.f) v = fminf(fmaxf(v, -clamp), clamp);
    return v;
}
