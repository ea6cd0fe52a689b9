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

// Per-device launch state.  A process may drive several devices (threaded data parallelism, a consumer that switches devices): the
// CU count and the "dynamic LDS above 64 KiB" function attribute belong to the CURRENT device, not to the process.  Lock-free: a
// racing thread recomputes the same value / sets the same attribute again.
#define SHG_MAX_DEVICES 64
static inline int shg_current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= SHG_MAX_DEVICES) d = 0;
    return d;
}
static inline int shg_cu_count() {
    static int cus[SHG_MAX_DEVICES];
    const int d = shg_current_device();
    int c = __atomic_load_n(&cus[d], __ATOMIC_RELAXED);
    if (!c) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || c <= 0) c = 256;
        __atomic_store_n(&cus[d], c, __ATOMIC_RELAXED);
    }
    return c;
}
struct ShgDeviceOnce {
    int done[SHG_MAX_DEVICES];
    bool pending(int d) const { return !__atomic_load_n(&done[d], __ATOMIC_ACQUIRE); }
    void mark(int d) { __atomic_store_n(&done[d], 1, __ATOMIC_RELEASE); }
};

// leaky-relu -> *gain -> clamp  (common/utils.py:135-143); clamp < 0 disables clamping.
__device__ __forceinline__ float shg_lrelu_agc(float v, float alpha, float gain, float clamp) {
    v = v < 0.f ? v * alpha : v;
    v *= gain;
    if (clamp >= 0.f) v = fminf(fmaxf(v, -clamp), clamp);
    return v;
}

// The same function with the per-launch decisions (activation on/off, clamp on/off) folded into four uniform constants, so that
// a kernel's tail is five VALU instructions per value without a branch: alpha = 1 and an infinite clamp make it `v * gain`.
// (fp32 MFMA and VALU instructions share the SIMD's fp32 datapath on gfx950: every VALU instruction of an epilogue is time the
// matrix pipe does not get.)  Same operations in the same order as shg_lrelu_agc: identical bits.
struct ShgAct { float alpha, gain, lo, hi; };
__device__ __forceinline__ ShgAct shg_act_make(int act, float alpha, float gain, float clamp) {
    ShgAct a;
    a.alpha = act ? alpha : 1.f;
    a.gain = gain;
    a.hi = (act && clamp >= 0.f) ? clamp : __builtin_inff();
    a.lo = -a.hi;
    return a;
}
__device__ __forceinline__ float shg_act_apply(float v, const ShgAct& a) {
    v = v < 0.f ? v * a.alpha : v;
    return __builtin_amdgcn_fmed3f(v * a.gain, a.lo, a.hi);
}
