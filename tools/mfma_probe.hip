// How fast does v_mfma_f32_32x32x2_f32 issue with 1, 2, 4 waves per SIMD, 1 or 2 accumulators, and with
// LDS reads / VALU between the MFMAs?  (cycles per MFMA per SIMD; 64 = the pipe's rate)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int FILL, bool M16>
__global__ void probe(int iters, unsigned long long* out, float* sink) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1.f;
    __syncthreads();
    f32x16 a0, a1; f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
    float x = (float)threadIdx.x, y = 1.0f;
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (M16) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c3, 0, 0, 0);
        } else {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            if (NACC == 2) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
            else a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        }
        if (FILL == 1) { y += lds[(threadIdx.x * 4 + i) & 4095]; }            // one ds_read per 2 MFMAs (dependent operand!)
        if (FILL == 2) { x = x * 1.0001f + 0.5f; }                            // VALU
    }
    const unsigned long long t1 = clock64();
    if (a0[0] + a1[0] + c0[0] + c1[0] + c2[0] + c3[0] == 123.456f) sink[0] = a0[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}
template <int NACC, int FILL, bool M16>
void run(const char* name, int threads) {
    unsigned long long* d; float* s; hipMalloc(&d, 8); hipMalloc(&s, 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NACC, FILL, M16>), dim3(256), dim3(threads), 0, 0, iters, d, s);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<NACC, FILL, M16>), dim3(256), dim3(threads), 0, 0, iters, d, s);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int wps = threads / 256;
    const double mfma_per_simd = (double)iters * (M16 ? 4 : 2) * wps;
    const double flop = mfma_per_simd * 4.0 * 256 * (M16 ? 2048.0 : 4096.0);
    printf("%-40s %4d threads (%d waves/SIMD): %.3f ms, %.1f ns per MFMA per SIMD, %.1f TFLOP/s\n", name, threads, wps,
           ms, ms * 1e6 / mfma_per_simd, flop / (ms * 1e-3) / 1e12);
    hipFree(d); hipFree(s);
}
int main() {
    for (int th : {256, 512, 1024}) {
        run<1, 0, false>("32x32x2 one accumulator (dependent)", th);
        run<2, 0, false>("32x32x2 two accumulators", th);
        run<2, 1, false>("32x32x2 two acc + ds_read feeding B", th);
        run<2, 2, false>("32x32x2 two acc + VALU feeding A", th);
        run<1, 0, true>("16x16x4 four accumulators", th);
    }
    return 0;
}
