// Shared helpers for the gfx950 kernels (internal; the public C-ABI is include/gm_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gm_hip.h"

extern "C" void gm_set_error(const char* msg);

#define GM_CHECK_ARG(cond)                                        \
    do {                                                          \
        if (!(cond)) {                                            \
            gm_set_error("bad argument: " #cond);                 \
            return GM_EINVAL;                                     \
        }                                                         \
    } while (0)

#define GM_LAUNCH_RET()                                           \
    do {                                                          \
        hipError_t e__ = hipGetLastError();                       \
        if (e__ != hipSuccess) {                                  \
            gm_set_error(hipGetErrorString(e__));                 \
            return -(int)e__;                                     \
        }                                                         \
        return 0;                                                 \
    } while (0)

// Device-side slot resolution: ((ctr ? *ctr : 0) * mul + add) % ring * stride.
__device__ __forceinline__ int64_t gm_slot_index(const gm_slot& s) {
    int64_t t = s.ctr ? *s.ctr : 0;
    int64_t i = t * (int64_t)s.mul + (int64_t)s.add;
    if (s.ring > 0) i %= (int64_t)s.ring;
    return i;
}
__device__ __forceinline__ int64_t gm_slot_offset(const gm_slot& s) {
    return gm_slot_index(s) * s.stride;
}

// wave64 reductions
__device__ __forceinline__ float gm_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double gm_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float gm_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
