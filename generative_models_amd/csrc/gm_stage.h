// gm_stage.h -- host ring -> device ring copies of the per-iteration draws (index batches, noise, eps: what
// ns_gan.py:218-226 / compute_noise produce on the host one step at a time), shared by the stage-in launches of
// gm_ops.hip and by the "stage the NEXT iteration" rider of the generator's weight-gradient pair (gm_gemm.hip).
#pragma once
#include "gm_common.h"

struct StageP {
    gm_stage_seg seg[GM_STAGE_MAX_SEGS];
    int n_segs;
    gm_slot slot;
    int n_iters;
    // fill gate (gm_stage_in_gated): the graph may be launched BEFORE the host has finished writing
    // its iterations' ring slots; every workgroup waits until *gate (pinned host memory, advanced by
    // the host after each sub-chunk of draws) covers them.  Bounded: after `timeout` ticks of the
    // 100 MHz wall clock the kernel raises gate[1] and copies what is there (the host checks it).
    const int64_t* gate;
    gm_slot it_slot;
    uint64_t timeout;
    int64_t* publish;       // optional: workgroup (0,0) stores the absolute iteration of it_slot here (a
                            // stable base for a second stage-in that runs concurrently with iterations
                            // that advance the step counter)
    // pre-staging (gm_stage_in_prestaged): *range = (lo << 32) | hi, the iterations [lo, hi) an EARLIER launch on
    // another stream has already brought into the device rings.  mark == 0: this launch returns at once when its
    // own iterations are inside the range; mark == 1: this launch is such an earlier one -- its last workgroup to
    // finish (arrive) extends the range (or restarts it at its own iterations when they do not continue it).
    unsigned long long* range;
    unsigned int* arrive;
    int mark;
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

// ---- stage-AHEAD rider (round 5) -----------------------------------------------------------------------------------
// The last launch of iteration i (the generator's weight-gradient pair, gm_linear_bwd_dw_adam_pair_stage) carries a few
// extra workgroups that bring iteration i + 1's draws into the device rings while the GEMM tiles run: a graph of ANY
// number of iterations then needs only its FIRST iteration staged before it starts, so a run is no longer cut into
// small first pieces that each wait for all of their draws (the 21 - 33 us piece boundaries and the ~60 us until the
// first kernel of a cold 20-step run, profiles/r04_experiments.md section 7).
//   gate[0] = iterations the host has WRITTEN, gate[1] = time-out flag, gate[2] = iterations whose draws the host has
//   SUBMITTED: the rider of a graph's last iteration (may_skip) skips an iteration that was not submitted -- the next
//   graph's first node stages it then; inner riders always deliver (the host submits a graph's draws right behind its
//   launch).
//   range = (lo << 32) | hi: iterations in the device rings (the word gm_stage_in_prestaged checks).
// The struct lives in DEVICE memory (gm_stage_ahead_pack): the segment table is indexed by the rider's workgroup id.
struct StageAheadP {
    gm_stage_seg seg[GM_STAGE_MAX_SEGS];
    int n_segs;
    int parts;                 // workgroups per segment
    gm_slot ring_slot;         // ring slot of the NEXT iteration
    gm_slot it_slot;           // its absolute index
    const int64_t* gate;
    uint64_t timeout;
    unsigned long long* range;
    unsigned int* arrive;      // low 16 bits: rider workgroups that are through; high bits: those that copied
    int may_skip;              // 1: the rider of a graph's LAST iteration -- the next iteration belongs to another graph,
                               //    which stages it itself if the host had not even submitted its draws when this ran
                               // 0: an inner iteration of a graph: the draws WILL be written (bounded gate wait)
};

// rid: rider workgroup 0 .. n_segs * parts - 1; scratch: >= 1 int of the workgroup's LDS
//
// Every PCIe round trip of the rider (~3 us) sits inside the launch it rides in, so it makes as few as it can: the two
// gate words it may need are requested TOGETHER (one round trip), the copy is the second.  (Measured, round 5: three
// sequential round trips made the pair launch 4 us longer; riders that instead polled a range pre-staged by the host on
// a side stream were free in steady state and lost 24 us per iteration whenever the host was late -- removed.)
__device__ __forceinline__ void stage_ahead_body(const StageAheadP& sa, int rid, int* scratch) {
    const int64_t next = gm_slot_index(sa.it_slot);
    if (threadIdx.x == 0) {
        int go = 0;
        const unsigned long long r = __hip_atomic_load(sa.range, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool in = (int64_t)(r >> 32) <= next && next < (int64_t)(r & 0xffffffffull);     // already on the device
        if (!in) {
            // both loads are in flight before either is used
            const int64_t filled = __hip_atomic_load(sa.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const int64_t submitted = __hip_atomic_load(sa.gate + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (!sa.may_skip || next < submitted) {
                go = 1;
                if (filled < next + 1) stage_gate_wait(sa.gate, next + 1, sa.timeout);
            }
        }
        scratch[0] = go;
    }
    __syncthreads();
    const int go = scratch[0];
    if (go) {
        const int si = rid / sa.parts, part = rid - si * sa.parts;
        stage_copy_seg(sa.seg[si], gm_slot_index(sa.ring_slot), 1, (int64_t)part * blockDim.x + threadIdx.x,
                       (int64_t)sa.parts * blockDim.x);
        __threadfence();                                  // this workgroup's ring writes: device-visible
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int total = (unsigned int)(sa.n_segs * sa.parts);
        const unsigned int old = __hip_atomic_fetch_add(sa.arrive, 1u + ((unsigned int)go << 16), __ATOMIC_ACQ_REL,
                                                        __HIP_MEMORY_SCOPE_AGENT);
        if ((old & 0xffffu) == total - 1) {               // last rider workgroup of this launch
            __hip_atomic_store(sa.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((old >> 16) + (unsigned int)go == total) {    // every segment part was copied HERE
                const unsigned long long r = __hip_atomic_load(sa.range, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long it0 = (unsigned long long)next;
                const unsigned long long lo = ((r & 0xffffffffull) == it0) ? (r >> 32) : it0;
                __hip_atomic_store(sa.range, (lo << 32) | (it0 + 1ull), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// the rider as a launch of its own (the pair it should ride in could not share a tile shape); static: the header is
// included by two translation units
static __global__ __launch_bounds__(1024) void stage_ahead_kernel(const StageAheadP* sa) {
    __shared__ int scratch[1];
    stage_ahead_body(*sa, (int)blockIdx.x, scratch);
}
