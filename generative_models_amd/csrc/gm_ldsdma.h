// gm_ldsdma.h -- LDS-DMA (global_load_lds_dwordx4) and the explicit waits that go with it, gfx950.
//
// Used by the weight gradients over >= 768 rows (gm_gemm.hip gemm16_dw_dma: autograd of ns_gan.py:138-139,155-156 at
// the bs=1024 shapes).  The stand-alone "slab" GEMM family these helpers were written for lives in tools/gm_slab.h
// (a measured negative result, profiles/r04_experiments.md; the product never launched it).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slab {

// 64 lanes x 16 bytes from per-lane global addresses to lds_byte + 16 * lane.  An asm statement: hipcc does not count
// it in its own s_waitcnt bookkeeping -- every wait below is explicit; M0 is written and restored inside.
__device__ __forceinline__ void glds16(const float* gsrc, uint32_t lds_byte) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// raw workgroup barrier (no vmcnt drain: LDS-DMA stays in flight across it), fenced for the compiler on both sides
__device__ __forceinline__ void sync_raw() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

}  // namespace slab
