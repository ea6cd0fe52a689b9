// Probe (round 4): semantics of `buffer_load_dwordx4 ... offen lds` on gfx950 that conv_wino4's window fetch relies on:
//   (a) M0 values above 64 KiB address the upper LDS;  (b) lanes whose offset fails the descriptor's range check write ZEROS to LDS
//   (or leave LDS untouched?);  (c) the scalar offset is not part of the range check.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const float* src, float* out, unsigned nrec, unsigned soff, unsigned ldsoff) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 40960; i += 64) L[i] = -7.f;           // sentinel over 160 KB
    __syncthreads();
    const unsigned long long b = (unsigned long long)src;
    i32x4 srd; srd[0] = (int)(unsigned)b; srd[1] = (int)(unsigned)(b >> 32); srd[2] = (int)nrec; srd[3] = 0x00020000;
    // lanes 0..31 in range (16 B each), lanes 32..47 at offsets beyond nrec, lanes 48..63 with the 0x80000000 marker
    int voff = lane < 32 ? lane * 16 : (lane < 48 ? (int)nrec + (lane - 32) * 16 : (int)0x80000000);
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)L) + ldsoff;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_waitcnt vmcnt(0)" ::"s"(lds), "v"(voff), "s"(srd), "s"(soff) : "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = L[ldsoff / 4 + i];
    // where did sentinel values change anywhere else?
    int changed = 0;
    for (int i = lane; i < 40960; i += 64) if ((i < (int)ldsoff / 4 || i >= (int)ldsoff / 4 + 256) && L[i] != -7.f) changed++;
    out[256 + lane] = (float)changed;
}

int main() {
    float *src, *out; const int N = 1 << 16;
    hipMalloc(&src, N * 4); hipMalloc(&out, 4096);
    float* h = new float[N]; for (int i = 0; i < N; ++i) h[i] = (float)i;
    hipMemcpy(src, h, N * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    struct { unsigned nrec, soff, ldsoff; const char* what; } cs[] = {
        {512, 0, 0, "nrec 512, soff 0, lds 0"}, {512, 4096, 0, "nrec 512, soff 4096 (beyond nrec), lds 0"},
        {512, 4096, 0x12000, "nrec 512, soff 4096, lds 0x12000"}, {512, 0, 0x20000, "lds 0x20000"}, {0, 0, 0x12000, "nrec 0"}};
    for (auto& c : cs) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 163840, 0, src, out, c.nrec, c.soff, c.ldsoff);
        float r[320]; hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
        printf("%s (err %d)\n", c.what, (int)hipGetLastError());
        for (int l : {0, 1, 31, 32, 33, 47, 48, 63}) printf("  lane %2d -> [%g %g %g %g]\n", l, r[l * 4], r[l * 4 + 1], r[l * 4 + 2], r[l * 4 + 3]);
        float ch = 0; for (int i = 0; i < 64; ++i) ch += r[256 + i];
        printf("  sentinel words changed elsewhere: %g\n", ch);
    }
    return 0;
}
