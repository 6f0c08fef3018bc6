// gm_stage.h -- host ring -> device ring copies of the per-iteration draws (index batches, noise, eps: what
// ns_gan.py:218-226 / compute_noise produce on the host one step at a time): the stage-in launches of gm_ops.hip.
// (Round 5 also tried a "stage the NEXT iteration" rider inside the generator's last launch so that a run could be
// ONE graph of exact length -- measured no better over 20 steps and 4 us per iteration worse in steady state, removed:
// profiles/r05_experiments.md section 3.)
#pragma once
#include "gm_common.h"

struct StageP {
    gm_stage_seg seg[GM_STAGE_MAX_SEGS];
    int n_segs;
    gm_slot slot;
    int n_iters;
    // (fill gate, gm_stage_in_gated: a one-wave launch in front of this one waits for the host's draws -- stage_gate_wait)
    gm_slot it_slot;
    int64_t* publish;       // optional: workgroup (0,0) stores the absolute iteration of it_slot here (a
                            // stable base for a second stage-in that runs concurrently with iterations
                            // that advance the step counter)
};

// thread t of `stride` copies its share of iterations [first, first + n_iters) of one segment
__device__ __forceinline__ void stage_copy_seg(const gm_stage_seg& sg, int64_t first, int n_iters, int64_t t,
                                               int64_t stride) {
    const int m = sg.blocks > 1 ? sg.blocks : 1;
    const int64_t bb = sg.bytes_per_iter / m;                     // bytes per piece
    const int64_t ss = sg.src_block_stride ? sg.src_block_stride : bb;
    const int64_t ds = sg.dst_block_stride ? sg.dst_block_stride : bb;
    const int64_t nblk = (int64_t)n_iters * m;
    const char* src = reinterpret_cast<const char*>(sg.src) + first * m * ss;
    char* dst = reinterpret_cast<char*>(sg.dst) + first * m * ds;
    if (ss == bb && ds == bb) {                                   // dense: one flat range
        const int64_t bytes = bb * nblk;
        if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)bytes) & 15) == 0) {
            const uint4* s4 = reinterpret_cast<const uint4*>(src);
            uint4* d4 = reinterpret_cast<uint4*>(dst);
            for (int64_t i = t; i < bytes / 16; i += stride) d4[i] = s4[i];
        } else {                                      // odd test shapes: 4-byte granularity
            const uint32_t* s1 = reinterpret_cast<const uint32_t*>(src);
            uint32_t* d1 = reinterpret_cast<uint32_t*>(dst);
            for (int64_t i = t; i < bytes / 4; i += stride) d1[i] = s1[i];
        }
        return;
    }
    // strided pieces (a data-parallel rank's rows of every draw of the global batch)
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)bb | (uintptr_t)ss |
          (uintptr_t)ds) & 15) == 0) {
        const int64_t upb = bb / 16;
        for (int64_t u = t; u < nblk * upb; u += stride) {
            const int64_t q = u / upb, o = u - q * upb;
            reinterpret_cast<uint4*>(dst + q * ds)[o] = reinterpret_cast<const uint4*>(src + q * ss)[o];
        }
    } else {
        const int64_t upb = bb / 4;
        for (int64_t u = t; u < nblk * upb; u += stride) {
            const int64_t q = u / upb, o = u - q * upb;
            reinterpret_cast<uint32_t*>(dst + q * ds)[o] = reinterpret_cast<const uint32_t*>(src + q * ss)[o];
        }
    }
}

__device__ __forceinline__ void stage_copy(const StageP& p) {
    stage_copy_seg(p.seg[blockIdx.y], gm_slot_index(p.slot), p.n_iters,
                   (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

// bounded wait of ONE lane until the host's fill counter gate[0] covers `need` iterations; raises gate[1] on time-out.
// RELAXED system-scope loads: the gate and the rings are fine-grained (uncached) host memory, so nothing stale can
// sit in L2; an ACQUIRE here costs a system-scope cache invalidate per workgroup (measured: 9 -> 29 us per stage-in
// of 8 iterations).
__device__ __forceinline__ void stage_gate_wait(const int64_t* gate, int64_t need, uint64_t timeout) {
    if (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= need) return;
    const uint64_t t0 = wall_clock64();
    while (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < need) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > timeout) {
            __hip_atomic_store(const_cast<int64_t*>(gate) + 1, (int64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
}
