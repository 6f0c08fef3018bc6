// gm_comm.hip -- gradient exchange of the data-parallel step as KERNELS inside the iteration's
// hipGraph (the reference has no distributed code at all).
//
// Round 1 exchanged the two gradient buckets of a D+G step (D 1.26 MB, G 1.29 MB) with two
// host-launched RCCL all-reduces between three segment graphs: 114 us per iteration on ONE rank
// against 71 us for the single-GPU graph -- the step is too short for host-launched collectives.
// Here every rank maps the other ranks' exchange regions (hipIpc handles over the xGMI peer
// mappings; fine-grained device memory).  The gradient bucket itself LIVES in the region (`in`: the
// backward kernels write dW/db straight into it -- dp.PeerComm.grad_buffer), so the all-reduce is two
// small kernels in the graph:
//   reduce: signal "in ready" to every peer (remote 8-byte stores), wait for theirs; every rank sums
//           ITS slice of the bucket over ranks 0..W-1 in rank order (identical bits everywhere)
//           into its `out`                                                     (reduce-scatter)
//   gather: signal "out ready", wait; read every slice from its owner, write the reduced gradient
//           back and -- optionally -- apply Adam to the parameters right there     (all-gather)
// (a bucket that lives elsewhere is first copied in by a third kernel, `stage`).
// xGMI is point-to-point: each rank moves 2 * (W-1)/W of the bucket over W-1 links in parallel
// (0.32 MB per link at W = 8) instead of W-1 ring hops.  Flags are monotonically increasing sequence
// numbers kept in device memory (the graph replays without arguments).  One wait per phase is
// enough, with single buffers: I overwrite `in` (next backward) only after my gather, which waited
// for every peer's "out ready", which a peer sends after its reduce kernel -- the only reader of my
// `in` -- has completed; I overwrite `out` only in my next reduce, after every peer's next "in
// ready", which it sends after its gather -- the only reader of my `out` -- has completed.
// Every wait is bounded: on expiry the kernel raises the communicator's error flag and carries on,
// so a lost peer shows up as a Python exception at the next read-back, never as a hung GPU.
#include "gm_common.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>

namespace {

constexpr int MAXW = 8;
constexpr int64_t FLAG_BYTES = 4096;             // [phase 0..7][source rank] u64, one 64-byte line each: 0 / 1 the pull
                                                 // exchange's two phases, 2 the scalar exchange, 4 / 5 the push exchange
constexpr int PUSH_PHASE_A = 4, PUSH_PHASE_B = 5;
constexpr int64_t SCAL_FLOATS = 64;              // scalar exchange: 2 parities x up to 16 values (+pad)
constexpr unsigned long long WAIT_TICKS = 10ull * 100000000ull;   // bounded waits: 10 s of the 100 MHz wall clock (default)

struct Region {            // layout of one rank's exchange region (byte offsets)
    int64_t flags, scal, in, out, stage, stage_stride, total;     // stage: [source rank][slice] of the push exchange
};
Region layout(int64_t n_floats, int world) {
    Region r;
    const int64_t nb = ((n_floats * 4 + 255) / 256) * 256;
    r.flags = 0;
    r.scal = FLAG_BYTES;
    r.in = r.scal + 2 * SCAL_FLOATS * 4 * MAXW;   // [parity][rank][16 floats]
    r.in = ((r.in + 255) / 256) * 256;
    r.out = r.in + nb;
    // push exchange: every peer deposits ITS contribution to my slice here.  A slice is ceil(n4 / world) float4 and
    // exactly `world` sources exist, so the area is about one bucket whatever the world size (it used to be
    // sized MAXW x half a bucket = 4 buckets for every communicator; every rank computes the same layout from the
    // same (n_floats, world), which gm_comm_connect relies on)
    const int64_t n4 = (n_floats + 3) / 4;
    r.stage = r.out + nb;
    r.stage_stride = world > 1 ? ((((n4 + world - 1) / world) * 16 + 16 + 255) / 256) * 256 : 0;
    r.total = r.stage + (int64_t)world * r.stage_stride;
    return r;
}

struct Comm {
    int rank, world;
    int64_t n_floats;
    Region lay;
    char* base[MAXW];          // mapped exchange regions (base[rank] = my own allocation)
    bool opened[MAXW];
    unsigned long long* seq;   // device: number of completed all-reduces; [1]: copy for the gather phase
    unsigned long long* sseq;  // device: number of completed scalar exchanges
    int* err;                  // device: set when a bounded wait expired
    unsigned* arrive;          // device: workgroups of the one-kernel exchange that have written their part of `out`
    int two_kernels;           // exchange form: 0 one kernel (pull), 1 reduce + gather launches (pull; ranks sharing ONE
                               // device), 2 one kernel, PUSH (posted remote writes only): gm_comm_set_exchange
    int coarse;                // 1: the region is plain hipMalloc memory (fine-grained allocation refused)
    int max_blocks;            // workgroups of an exchange launch: what can be co-resident on THIS device (or less)
    unsigned long long wait_ticks;   // bound of every device-side wait (gm_comm_set_wait_seconds)
};

struct CommP {
    int rank, world;
    char* base[MAXW];
    Region lay;
    unsigned long long* seq;
    unsigned long long* sseq;
    int* err;
    unsigned* arrive;
    unsigned long long wait_ticks;
};

__device__ __forceinline__ unsigned long long* flag_ptr(const CommP& c, int owner, int phase, int src) {
    return reinterpret_cast<unsigned long long*>(c.base[owner] + c.lay.flags) + (phase * MAXW + src) * 8;
}

// signal phase `phase` of sequence s to every peer (a remote 8-byte store into THEIR flag array),
// then wait until every peer's signal for the same phase has arrived in MINE.  ONE lane per
// workgroup does the fences and the polling (a system-scope fence by every thread of every workgroup
// was measured at 22 us per kernel); the workgroup barrier hands the acquire to the other lanes.
__device__ void signal_and_wait(const CommP& c, int phase, unsigned long long s) {
    if (c.world == 1) return;                     // uniform: nothing to exchange
    if (threadIdx.x == 0) {
        if (blockIdx.x == 0) {
            __atomic_thread_fence(__ATOMIC_RELEASE);      // system scope: the previous kernel's stores are out
            for (int peer = 0; peer < c.world; ++peer)
                if (peer != c.rank)
                    __hip_atomic_store(flag_ptr(c, peer, phase, c.rank), s, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // once a wait has expired the run is lost anyway: later waits return at once instead of
        // costing another time-out each (a broken peer mapping must not stall start-up for minutes)
        bool dead = __hip_atomic_load(c.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        const unsigned long long t0 = wall_clock64();
        for (int peer = 0; peer < c.world && !dead; ++peer) {
            if (peer == c.rank) continue;
            unsigned int spins = 0;
            while (__hip_atomic_load(flag_ptr(c, c.rank, phase, peer), __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_SYSTEM) < s) {
                __builtin_amdgcn_s_sleep(2);
                if ((++spins & 255u) == 0 && wall_clock64() - t0 > c.wait_ticks) {
                    atomicExch(c.err, 1);
                    dead = true;
                    break;
                }
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);          // system scope: see what the peers published
    }
    __syncthreads();
}

__device__ __forceinline__ void slice_of(int64_t n4, int world, int r, int64_t* lo, int64_t* hi) {
    const int64_t per = (n4 + world - 1) / world;
    *lo = per * r < n4 ? per * r : n4;
    *hi = per * (r + 1) < n4 ? per * (r + 1) : n4;
}

// stage: a bucket that does not live in the exchange region -> in
__global__ __launch_bounds__(256) void stage_kernel(CommP c, const float* __restrict__ g, int64_t n) {
    float4* dst = reinterpret_cast<float4*>(c.base[c.rank] + c.lay.in);
    const float4* src = reinterpret_cast<const float4*>(g);
    const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}

// reduce-scatter
__global__ __launch_bounds__(256) void reduce_kernel(CommP c, int64_t n) {
    const unsigned long long s = c.seq[0] + 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) c.seq[1] = s;        // the gather phase reads seq[1]
    signal_and_wait(c, 0, s);
    const int64_t n4 = n >> 2;
    int64_t lo, hi;
    slice_of(n4, c.world, c.rank, &lo, &hi);
    float4* out = reinterpret_cast<float4*>(c.base[c.rank] + c.lay.out);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += stride) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = 0; r < c.world; ++r) {                       // rank order: same bits on every rank
            const float4 v = reinterpret_cast<const float4*>(c.base[r] + c.lay.in)[i];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        out[i] = a;
    }
}

struct AdamP {
    float* p; float* m; float* v;
    const float* sched; gm_slot sched_slot;
    float omb1, b2, omb2, eps, wd, clamp;
    const float* lr_scale;
    int enabled;
};

// all-gather (+ Adam): every element comes from its slice owner's out[parity]
__global__ __launch_bounds__(256) void gather_kernel(CommP c, float* __restrict__ g, int64_t n, AdamP ad) {
    const unsigned long long s = c.seq[1];
    signal_and_wait(c, 1, s);
    const int64_t n4 = n >> 2;
    float step_size = 0.f, bc2_sqrt = 1.f;
    if (ad.enabled) {
        const int64_t si = gm_slot_index(ad.sched_slot);
        step_size = ad.sched[2 * si] * (ad.lr_scale ? ad.lr_scale[0] : 1.0f);
        bc2_sqrt = ad.sched[2 * si + 1];
    }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int r = 0; r < c.world; ++r) {
        int64_t lo, hi;
        slice_of(n4, c.world, r, &lo, &hi);
        const float4* src = reinterpret_cast<const float4*>(c.base[r] + c.lay.out);
        for (int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += stride) {
            const float4 G = src[i];
            reinterpret_cast<float4*>(g)[i] = G;
            if (ad.enabled) {
                float4 P = reinterpret_cast<float4*>(ad.p)[i];
                float4 M = reinterpret_cast<float4*>(ad.m)[i];
                float4 V = reinterpret_cast<float4*>(ad.v)[i];
                adam_update(P.x, G.x, M.x, V.x, step_size, bc2_sqrt, ad.omb1, ad.b2, ad.omb2, ad.eps, ad.wd, ad.clamp);
                adam_update(P.y, G.y, M.y, V.y, step_size, bc2_sqrt, ad.omb1, ad.b2, ad.omb2, ad.eps, ad.wd, ad.clamp);
                adam_update(P.z, G.z, M.z, V.z, step_size, bc2_sqrt, ad.omb1, ad.b2, ad.omb2, ad.eps, ad.wd, ad.clamp);
                adam_update(P.w, G.w, M.w, V.w, step_size, bc2_sqrt, ad.omb1, ad.b2, ad.omb2, ad.eps, ad.wd, ad.clamp);
                reinterpret_cast<float4*>(ad.p)[i] = P;
                reinterpret_cast<float4*>(ad.m)[i] = M;
                reinterpret_cast<float4*>(ad.v)[i] = V;
            }
        }
    }
    // last: publish the completed sequence number.  Nobody in THIS kernel reads seq[0]; the next
    // stage kernel does, after the kernel boundary.
    if (blockIdx.x == 0 && threadIdx.x == 0) c.seq[0] = s;
}

// wait (bounded) until every peer's flag of `phase` has reached s in MY flag array; one lane per workgroup
__device__ void wait_peers(const CommP& c, int phase, unsigned long long s) {
    bool dead = __hip_atomic_load(c.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    const unsigned long long t0 = wall_clock64();
    for (int peer = 0; peer < c.world && !dead; ++peer) {
        if (peer == c.rank) continue;
        unsigned int spins = 0;
        while (__hip_atomic_load(flag_ptr(c, c.rank, phase, peer), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < s) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 255u) == 0 && wall_clock64() - t0 > c.wait_ticks) {
                atomicExch(c.err, 1);
                dead = true;
                break;
            }
        }
    }
}

// Round 4: reduce-scatter AND all-gather (+ Adam) as ONE kernel -- the data-parallel iteration is the single-GPU
// iteration's 8 launches + one exchange per optimizer (round 3: + two).  What the kernel boundary between reduce and
// gather provided -- "every workgroup of this rank has written its part of `out`" before the peers are told -- is an
// arrival counter: every wave drains its stores, the workgroup's lane 0 releases and arrives, the LAST arriver re-arms
// the counter and stores the "out ready" flags into the peers' regions.  Nobody waits on the counter: workgroup b
// reduces and gathers the SAME indices of the rank's own slice (the thread that summed an element is the one that
// reads it back), and the peers' slices are guarded by their flags.  All workgroups are co-resident (<= 320 x 256
// threads on 256 CUs), so the flag waits cannot starve the arrivals they depend on.  One rank: the gradient is taken
// straight from `in`, no `out`, no counter.
__global__ __launch_bounds__(256) void xchg_kernel(CommP c, float* __restrict__ g, int64_t n, AdamP ad) {
    const unsigned long long s = c.seq[0] + 1;
    signal_and_wait(c, 0, s);                       // every rank's bucket is complete (kernel boundary + flags)
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c.world > 1) {
        int64_t lo, hi;
        slice_of(n4, c.world, c.rank, &lo, &hi);
        float4* out = reinterpret_cast<float4*>(c.base[c.rank] + c.lay.out);
        for (int64_t i = lo + first; i < hi; i += stride) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int r = 0; r < c.world; ++r) {                   // rank order: same bits on every rank
                const float4 v = reinterpret_cast<const float4*>(c.base[r] + c.lay.in)[i];
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            out[i] = a;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's `out` stores have left
        __syncthreads();
        if (threadIdx.x == 0) {
            __atomic_thread_fence(__ATOMIC_RELEASE);              // system scope
            const unsigned old = __hip_atomic_fetch_add(c.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == gridDim.x - 1) {                           // last of this rank: tell the peers, re-arm
                __hip_atomic_store(c.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // the sequence number advances HERE: every workgroup of this launch has read it by now (it arrived).
                // Written at the end of workgroup 0 instead, a workgroup that starts late -- several processes
                // time-sharing one GPU -- would read the NEW value and wait for flags of the next exchange.
                c.seq[0] = s;
                for (int peer = 0; peer < c.world; ++peer)
                    if (peer != c.rank)
                        __hip_atomic_store(flag_ptr(c, peer, 1, c.rank), s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            wait_peers(c, 1, s);
            __atomic_thread_fence(__ATOMIC_ACQUIRE);              // system scope: see the peers' `out`
        }
        __syncthreads();
    }
    float step_size = 0.f, bc2_sqrt = 1.f;
    if (ad.enabled) {
        const int64_t si = gm_slot_index(ad.sched_slot);
        step_size = ad.sched[2 * si] * (ad.lr_scale ? ad.lr_scale[0] : 1.0f);
        bc2_sqrt = ad.sched[2 * si + 1];
    }
    for (int r = 0; r < c.world; ++r) {
        int64_t lo, hi;
        slice_of(n4, c.world, r, &lo, &hi);
        const float4* src = reinterpret_cast<const float4*>(c.base[r] + (c.world > 1 ? c.lay.out : c.lay.in));
        for (int64_t i = lo + first; i < hi; i += stride) {
            const float4 G = src[i];
            if (c.world > 1 || reinterpret_cast<const float4*>(g) != src) reinterpret_cast<float4*>(g)[i] = G;
            if (ad.enabled) {
                float4 P = reinterpret_cast<float4*>(ad.p)[i];
                float4 M = reinterpret_cast<float4*>(ad.m)[i];
                float4 V = reinterpret_cast<float4*>(ad.v)[i];
                adam_update(P.x, G.x, M.x, V.x, step_size, bc2_sqrt, ad.omb1, ad.b2, ad.omb2, ad.eps, ad.wd, ad.clamp);
                adam_update(P.y, G.y, M.y, V.y, step_size, bc2_sqrt, ad.omb1, ad.b2, ad.omb2, ad.eps, ad.wd, ad.clamp);
                adam_update(P.z, G.z, M.z, V.z, step_size, bc2_sqrt, ad.omb1, ad.b2, ad.omb2, ad.eps, ad.wd, ad.clamp);
                adam_update(P.w, G.w, M.w, V.w, step_size, bc2_sqrt, ad.omb1, ad.b2, ad.omb2, ad.eps, ad.wd, ad.clamp);
                reinterpret_cast<float4*>(ad.p)[i] = P;
                reinterpret_cast<float4*>(ad.m)[i] = M;
                reinterpret_cast<float4*>(ad.v)[i] = V;
            }
        }
    }
}

// PUSH exchange (round 5): the same sum, moved by posted remote WRITES only.  The pull forms above
// READ the peers' `in` (reduce-scatter) and `out` (all-gather) over xGMI -- every cache line a full round trip with a
// bounded number outstanding.  Here rank r
//   A  writes its contribution to every other rank's slice into THAT rank's stage[r] (remote stores), and -- once all
//      its workgroups have done so (arrival counter 1, release) -- raises flag phase 4 at every peer;
//   B  waits for the peers' phase-4 flags, sums ITS slice over ranks 0..W-1 in rank order (its own part straight from
//      `in`, the others from its local stage: the same values in the same order as the pull forms -- bit-identical),
//      writes the result into its own `out` AND into every peer's `out` at the slice's place (remote stores); after
//      arrival counter 2, flag phase 5;
//   C  waits for the peers' phase-5 flags and reads the whole reduced bucket from its LOCAL `out` (+ Adam).
// Every element of `in` / `out` / stage is touched by one thread of one rank per step (the grid-stride index sets of
// A, B and C coincide per slice), so single buffers are safe by the same argument as above: a peer overwrites my
// stage (exchange s + 1, step A) only after its step C of s, which waited for my phase-5 flag, which follows my step
// B -- the only reader of my stage.  Needs all workgroups of a launch co-resident, like xchg_kernel.
__device__ __forceinline__ void arrive_then_flag(const CommP& c, unsigned* counter, int phase, unsigned long long s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's (remote) stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);                  // system scope
        const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gridDim.x - 1) {                               // last workgroup of this rank: tell the peers, re-arm
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (phase == PUSH_PHASE_B) c.seq[0] = s;              // (every workgroup has read it: it arrived twice)
            for (int peer = 0; peer < c.world; ++peer)
                if (peer != c.rank)
                    __hip_atomic_store(flag_ptr(c, peer, phase, c.rank), s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        wait_peers(c, phase, s);
        __atomic_thread_fence(__ATOMIC_ACQUIRE);                  // system scope: see what the peers deposited
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void push_kernel(CommP c, float* __restrict__ g, int64_t n, AdamP ad) {
    const unsigned long long s = c.seq[0] + 1;
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float4* in = reinterpret_cast<const float4*>(c.base[c.rank] + c.lay.in);
    float4* out = reinterpret_cast<float4*>(c.base[c.rank] + c.lay.out);
    if (c.world > 1) {
        // A: my contributions to the peers' slices
        for (int o = 0; o < c.world; ++o) {
            if (o == c.rank) continue;
            int64_t lo, hi;
            slice_of(n4, c.world, o, &lo, &hi);
            float4* dst = reinterpret_cast<float4*>(c.base[o] + c.lay.stage + (int64_t)c.rank * c.lay.stage_stride);
            for (int64_t i = lo + first; i < hi; i += stride) dst[i - lo] = in[i];
        }
        arrive_then_flag(c, c.arrive + 1, PUSH_PHASE_A, s);
        // B: my slice, summed in rank order, to everybody's `out`
        int64_t lo, hi;
        slice_of(n4, c.world, c.rank, &lo, &hi);
        for (int64_t i = lo + first; i < hi; i += stride) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int r = 0; r < c.world; ++r) {
                const float4 v = (r == c.rank) ? in[i]
                    : reinterpret_cast<const float4*>(c.base[c.rank] + c.lay.stage + (int64_t)r * c.lay.stage_stride)[i - lo];
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            out[i] = a;
            for (int peer = 0; peer < c.world; ++peer)
                if (peer != c.rank) reinterpret_cast<float4*>(c.base[peer] + c.lay.out)[i] = a;
        }
        arrive_then_flag(c, c.arrive + 2, PUSH_PHASE_B, s);
    }
    float step_size = 0.f, bc2_sqrt = 1.f;
    if (ad.enabled) {
        const int64_t si = gm_slot_index(ad.sched_slot);
        step_size = ad.sched[2 * si] * (ad.lr_scale ? ad.lr_scale[0] : 1.0f);
        bc2_sqrt = ad.sched[2 * si + 1];
    }
    // C: the whole reduced bucket is local now (per slice: the same indices this thread touched in A / B)
    for (int r = 0; r < c.world; ++r) {
        int64_t lo, hi;
        slice_of(n4, c.world, r, &lo, &hi);
        const float4* src = (c.world > 1) ? out : in;
        for (int64_t i = lo + first; i < hi; i += stride) {
            const float4 G = src[i];
            if (c.world > 1 || reinterpret_cast<const float4*>(g) != src) reinterpret_cast<float4*>(g)[i] = G;
            if (ad.enabled) {
                float4 P = reinterpret_cast<float4*>(ad.p)[i];
                float4 M = reinterpret_cast<float4*>(ad.m)[i];
                float4 V = reinterpret_cast<float4*>(ad.v)[i];
                adam_update(P.x, G.x, M.x, V.x, step_size, bc2_sqrt, ad.omb1, ad.b2, ad.omb2, ad.eps, ad.wd, ad.clamp);
                adam_update(P.y, G.y, M.y, V.y, step_size, bc2_sqrt, ad.omb1, ad.b2, ad.omb2, ad.eps, ad.wd, ad.clamp);
                adam_update(P.z, G.z, M.z, V.z, step_size, bc2_sqrt, ad.omb1, ad.b2, ad.omb2, ad.eps, ad.wd, ad.clamp);
                adam_update(P.w, G.w, M.w, V.w, step_size, bc2_sqrt, ad.omb1, ad.b2, ad.omb2, ad.eps, ad.wd, ad.clamp);
                reinterpret_cast<float4*>(ad.p)[i] = P;
                reinterpret_cast<float4*>(ad.m)[i] = M;
                reinterpret_cast<float4*>(ad.v)[i] = V;
            }
        }
    }
}

// Scalar exchange: vals[0..k) <- sum over ranks (rank order) of every rank's vals[0..k), k <= 16.
// One workgroup: write mine into every peer's slot array (remote stores), signal, wait, sum locally.
__global__ __launch_bounds__(64) void scalars_kernel(CommP c, float* __restrict__ vals, int k) {
    const unsigned long long s = c.sseq[0] + 1;
    const int par = (int)(s & 1);
    if ((int)threadIdx.x < k) {
        const float v = vals[threadIdx.x];
        for (int r = 0; r < c.world; ++r) {
            float* slot = reinterpret_cast<float*>(c.base[r] + c.lay.scal) + ((par * MAXW + c.rank) * 16);
            __hip_atomic_store(slot + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __syncthreads();
    signal_and_wait(c, 2, s);
    if ((int)threadIdx.x < k) {
        float a = 0.f;
        for (int r = 0; r < c.world; ++r) {
            const float* slot = reinterpret_cast<const float*>(c.base[c.rank] + c.lay.scal) + ((par * MAXW + r) * 16);
            a += __hip_atomic_load(slot + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        vals[threadIdx.x] = a;
    }
    if (threadIdx.x == 0) c.sseq[0] = s;
}

CommP params_of(const Comm* cm) {
    CommP p{};
    p.rank = cm->rank; p.world = cm->world; p.lay = cm->lay; p.seq = cm->seq; p.sseq = cm->sseq; p.err = cm->err;
    p.arrive = cm->arrive;
    p.wait_ticks = cm->wait_ticks ? cm->wait_ticks : WAIT_TICKS;
    for (int i = 0; i < MAXW; ++i) p.base[i] = cm->base[i];
    return p;
}

#define GM_HIPC(call)                                             \
    do {                                                          \
        hipError_t e__ = (call);                                  \
        if (e__ != hipSuccess) {                                  \
            gm_set_error(hipGetErrorString(e__));                 \
            return -(int)e__;                                     \
        }                                                         \
    } while (0)

}  // namespace

extern "C" int gm_comm_destroy(void* comm);

// The exchange region must be FINE-GRAINED device memory when peers on other GPUs store into it and
// a running kernel polls it (no kernel boundary between a peer's store and my load).  If the runtime
// refuses that allocation the region falls back to plain hipMalloc and the communicator is marked
// coarse (gm_comm_info): correct when every rank shares ONE device (the single-GPU multi-process
// tests), NOT across GPUs -- dp.PeerComm refuses a coarse region there and the engine drops to RCCL.
static int comm_create_impl(Comm* cm, void* handle_out64) {
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, (size_t)cm->lay.total, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        cm->coarse = 1;
        GM_HIPC(hipMalloc(&p, (size_t)cm->lay.total));
    }
    cm->base[cm->rank] = static_cast<char*>(p);           // owned from here on: gm_comm_destroy frees it
    GM_HIPC(hipMemset(p, 0, (size_t)cm->lay.total));
    void* ctr = nullptr;
    GM_HIPC(hipMalloc(&ctr, 64));
    cm->seq = static_cast<unsigned long long*>(ctr);
    GM_HIPC(hipMemset(ctr, 0, 64));
    cm->sseq = cm->seq + 2;
    cm->err = reinterpret_cast<int*>(cm->seq + 4);
    cm->arrive = reinterpret_cast<unsigned*>(cm->seq + 6);
    hipIpcMemHandle_t h;
    std::memset(&h, 0, sizeof(h));
    if (cm->world > 1) GM_HIPC(hipIpcGetMemHandle(&h, p));
    static_assert(sizeof(hipIpcMemHandle_t) <= 64, "handle travels as 64 bytes");
    std::memset(handle_out64, 0, 64);
    std::memcpy(handle_out64, &h, sizeof(h));
    GM_HIPC(hipDeviceSynchronize());
    return 0;
}

extern "C" int gm_comm_create(int rank, int world, int64_t n_floats, void** comm_out, void* handle_out64) {
    GM_CHECK_ARG(comm_out && handle_out64 && world >= 1 && world <= MAXW && rank >= 0 && rank < world && n_floats > 0);
    Comm* cm = new Comm();
    cm->rank = rank; cm->world = world; cm->n_floats = n_floats; cm->lay = layout(n_floats, world);
    cm->seq = nullptr; cm->sseq = nullptr; cm->err = nullptr; cm->coarse = 0;
    for (int i = 0; i < MAXW; ++i) { cm->base[i] = nullptr; cm->opened[i] = false; }
    const int rc = comm_create_impl(cm, handle_out64);
    if (rc) {                                             // nothing leaks on a failed construction
        (void)gm_comm_destroy(cm);
        *comm_out = nullptr;
        return rc;
    }
    *comm_out = cm;
    return 0;
}

extern "C" int gm_comm_info(void* comm, int* fine_grained_out) {
    Comm* cm = static_cast<Comm*>(comm);
    GM_CHECK_ARG(cm && fine_grained_out);
    *fine_grained_out = cm->coarse ? 0 : 1;
    return 0;
}

extern "C" int gm_comm_connect(void* comm, const void* all_handles) {
    Comm* cm = static_cast<Comm*>(comm);
    GM_CHECK_ARG(cm && (all_handles || cm->world == 1));
    for (int r = 0; r < cm->world; ++r) {
        if (r == cm->rank) continue;
        hipIpcMemHandle_t h;
        std::memcpy(&h, static_cast<const char*>(all_handles) + 64 * r, sizeof(h));
        void* p = nullptr;
        GM_HIPC(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        cm->base[r] = static_cast<char*>(p);
        cm->opened[r] = true;
    }
    return 0;
}

extern "C" int gm_comm_destroy(void* comm) {
    Comm* cm = static_cast<Comm*>(comm);
    if (!cm) return 0;
    for (int r = 0; r < cm->world; ++r)
        if (cm->opened[r]) (void)hipIpcCloseMemHandle(cm->base[r]);
    if (cm->base[cm->rank]) (void)hipFree(cm->base[cm->rank]);
    if (cm->seq) (void)hipFree(cm->seq);
    delete cm;
    return 0;
}

extern "C" int gm_comm_buffer(void* comm, void** ptr_out, int64_t* n_floats_out) {
    Comm* cm = static_cast<Comm*>(comm);
    GM_CHECK_ARG(cm && ptr_out && n_floats_out);
    *ptr_out = cm->base[cm->rank] + cm->lay.in;
    *n_floats_out = cm->n_floats;
    return 0;
}

// The one-kernel exchange keeps ~300 workgroups of every rank spinning on their peers' flags; with a GPU per rank
// that costs nothing, but ranks that SHARE one device (the single-GPU multi-process tests and dry runs) then hold
// the registers the peers' 1024-thread GEMM workgroups need to get on a CU at all -- measured: 4 ranks x 256 rows of the
// 784-400-20 model on one MI355X starve each other until the bounded waits expire.  two_kernels = 1 selects round 3's
// reduce + gather pair, whose first wait is a 77-workgroup, 18-register kernel.
extern "C" int gm_comm_set_exchange(void* comm, int two_kernels) {
    Comm* cm = static_cast<Comm*>(comm);
    GM_CHECK_ARG(cm && two_kernels >= 0 && two_kernels <= 2);
    cm->two_kernels = two_kernels;
    return 0;
}

// Every workgroup of the one-kernel exchange spins until the last workgroup of every peer has arrived, so all of a
// launch's workgroups must be co-resident.  Default: what the occupancy API says fits on this device (a CU-masked or
// partitioned device -- CPX mode, HSA_CU_MASK -- reports fewer CUs), never more than 320; a caller whose ranks share a
// device lowers it further.  The kernels loop over the bucket, so any positive count is correct.
static int resident_blocks() {
    int dev = 0, per_cu = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 64;
    // the cap must hold for WHICHEVER one-kernel exchange is launched: the smaller of the two occupancies
    int per_push = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, xchg_kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_push, push_kernel, 256, 0) != hipSuccess || per_push < 1) per_push = 1;
    if (per_push < per_cu) per_cu = per_push;
    const long cap = (long)per_cu * pr.multiProcessorCount;
    return (int)(cap < 320 ? (cap < 1 ? 1 : cap) : 320);
}

// Bound of every device-side wait of this communicator's launches from now on (default 10 s).  First contact between
// two devices -- the start-up self-check -- runs with a short one: a peer mapping that is not coherent must cost
// seconds, not minutes, before every rank falls back.
extern "C" int gm_comm_set_wait_seconds(void* comm, double seconds) {
    Comm* cm = static_cast<Comm*>(comm);
    GM_CHECK_ARG(cm && seconds > 0.0 && seconds <= 600.0);
    cm->wait_ticks = (unsigned long long)(seconds * 1.0e8);
    return 0;
}

extern "C" int gm_comm_set_max_blocks(void* comm, int max_blocks) {
    Comm* cm = static_cast<Comm*>(comm);
    GM_CHECK_ARG(cm && max_blocks >= 0);
    const int cap = resident_blocks();
    cm->max_blocks = (max_blocks == 0 || max_blocks > cap) ? cap : max_blocks;
    return 0;
}

extern "C" int gm_comm_error(void* comm, int* flag_out) {
    Comm* cm = static_cast<Comm*>(comm);
    GM_CHECK_ARG(cm && flag_out);
    GM_HIPC(hipMemcpy(flag_out, cm->err, sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

static int allreduce_impl(Comm* cm, hipStream_t s, float* buf, int64_t n, const AdamP& ad) {
    GM_CHECK_ARG(cm && buf && n > 0 && n <= cm->n_floats && n % 4 == 0);
    GM_CHECK_ARG((reinterpret_cast<uintptr_t>(buf) & 15) == 0);
    const CommP p = params_of(cm);
    // one 16-byte element per thread where the bucket allows it (a D+G step's buckets are ~80k
    // float4: 256 workgroups of 256 threads, one pass).  Waiting workgroups poll flags in THEIR OWN
    // memory (peers store remotely), so many pollers cost no xGMI traffic.
    if (cm->max_blocks <= 0) cm->max_blocks = resident_blocks();
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks > cm->max_blocks) blocks = cm->max_blocks;
    if (blocks < 1) blocks = 1;
    int rblocks = (int)((n / 4 / cm->world + 255) / 256);     // a slice per rank
    if (rblocks > cm->max_blocks) rblocks = cm->max_blocks;
    if (rblocks < 1) rblocks = 1;
    if (reinterpret_cast<char*>(buf) != cm->base[cm->rank] + cm->lay.in)    // the bucket lives elsewhere
        hipLaunchKernelGGL(stage_kernel, dim3(blocks), dim3(256), 0, s, p, buf, n);
    if (cm->two_kernels == 2) {
        hipLaunchKernelGGL(push_kernel, dim3(blocks), dim3(256), 0, s, p, buf, n, ad);
    } else if (cm->two_kernels) {
        hipLaunchKernelGGL(reduce_kernel, dim3(rblocks), dim3(256), 0, s, p, n);
        hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(256), 0, s, p, buf, n, ad);
    } else {
        hipLaunchKernelGGL(xchg_kernel, dim3(blocks), dim3(256), 0, s, p, buf, n, ad);
    }
    GM_LAUNCH_RET();
}

extern "C" int gm_allreduce_f32(void* comm, void* stream, float* buf, int64_t n) {
    AdamP ad{};
    return allreduce_impl(static_cast<Comm*>(comm), (hipStream_t)stream, buf, n, ad);
}

extern "C" int gm_allreduce_adam_f32(void* comm, void* stream, float* grad, int64_t n, float* p,
                                     float* m, float* v, const float* sched, gm_slot sched_slot,
                                     double beta1, double beta2, double eps, double weight_decay,
                                     float clamp, const float* lr_scale) {
    GM_CHECK_ARG(p && m && v && sched);
    GM_CHECK_ARG(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) |
                   reinterpret_cast<uintptr_t>(v)) & 15) == 0);
    AdamP ad{};
    ad.p = p; ad.m = m; ad.v = v; ad.sched = sched; ad.sched_slot = sched_slot;
    ad.omb1 = (float)(1.0 - beta1); ad.b2 = (float)beta2; ad.omb2 = (float)(1.0 - beta2);
    ad.eps = (float)eps; ad.wd = (float)weight_decay; ad.clamp = clamp; ad.lr_scale = lr_scale;
    ad.enabled = 1;
    return allreduce_impl(static_cast<Comm*>(comm), (hipStream_t)stream, grad, n, ad);
}

extern "C" int gm_allreduce_scalars(void* comm, void* stream, float* vals, int k) {
    Comm* cm = static_cast<Comm*>(comm);
    GM_CHECK_ARG(cm && vals && k > 0 && k <= 16);
    hipLaunchKernelGGL(scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, params_of(cm), vals, k);
    GM_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------
// RCCL collectives INSIDE the iteration's hipGraph (round 5): the fallback when peer mappings are refused on a node
// (GM_DP_COMM=rccl).  Round 1 launched torch.distributed all-reduces from the host between three segment graphs:
// 114 us per iteration on ONE rank against 71 us for the single graph.  RCCL supports stream capture, so the
// all-reduce of a gradient bucket becomes a node of the same one graph per iteration the single-GPU and the peer
// paths replay.  The library does NOT link RCCL: the symbols are resolved at first use from whatever librccl the
// process already carries (torch's bundled one -- two copies of RCCL in one process is what linking /opt/rocm's would
// risk), so the C-ABI library still loads where there is no RCCL at all.
// ------------------------------------------------------------------------------------------
namespace {
struct NcclUid { char b[128]; };                       // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*nccl_get_uid_t)(NcclUid*);
typedef int (*nccl_init_rank_t)(void**, int, NcclUid, int);
typedef int (*nccl_allreduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_destroy_t)(void*);
typedef const char* (*nccl_errstr_t)(int);

void* nccl_sym(const char* name) {
    void* f = dlsym(RTLD_DEFAULT, name);
    if (!f) {
        static void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (h) f = dlsym(h, name);
    }
    return f;
}
int nccl_fail(int rc, const char* what) {
    nccl_errstr_t es = reinterpret_cast<nccl_errstr_t>(nccl_sym("ncclGetErrorString"));
    static thread_local char msg[256];
    snprintf(msg, sizeof(msg), "%s: %s", what, es ? es(rc) : "RCCL error");
    gm_set_error(msg);
    return GM_EINVAL - 100 - rc;
}
}  // namespace

// 1 when every RCCL entry point this library calls resolves in this process, else 0 -- asked on EVERY rank before the
// collective ncclCommInitRank, so that a rank that cannot start makes all ranks fall back instead of leaving the others
// blocked inside the init.
extern "C" int gm_rccl_available(void) {
    for (const char* n : {"ncclGetUniqueId", "ncclCommInitRank", "ncclAllReduce", "ncclCommDestroy"})
        if (!nccl_sym(n)) return 0;
    return 1;
}

extern "C" int gm_rccl_unique_id(void* uid128_out) {
    GM_CHECK_ARG(uid128_out);
    nccl_get_uid_t f = reinterpret_cast<nccl_get_uid_t>(nccl_sym("ncclGetUniqueId"));
    if (!f) { gm_set_error("RCCL is not loaded in this process (ncclGetUniqueId not found)"); return GM_EINVAL; }
    const int rc = f(static_cast<NcclUid*>(uid128_out));
    return rc ? nccl_fail(rc, "ncclGetUniqueId") : 0;
}

extern "C" int gm_rccl_comm_create(int rank, int world, const void* uid128, void** comm_out) {
    GM_CHECK_ARG(uid128 && comm_out && world >= 1 && rank >= 0 && rank < world);
    nccl_init_rank_t f = reinterpret_cast<nccl_init_rank_t>(nccl_sym("ncclCommInitRank"));
    if (!f) { gm_set_error("RCCL is not loaded in this process (ncclCommInitRank not found)"); return GM_EINVAL; }
    NcclUid id;
    std::memcpy(&id, uid128, sizeof(id));
    void* c = nullptr;
    const int rc = f(&c, world, id, rank);
    if (rc) return nccl_fail(rc, "ncclCommInitRank");
    *comm_out = c;
    return 0;
}

extern "C" int gm_rccl_allreduce_f32(void* rccl_comm, void* stream, float* buf, int64_t n) {
    GM_CHECK_ARG(rccl_comm && buf && n > 0);
    static nccl_allreduce_t f = reinterpret_cast<nccl_allreduce_t>(nccl_sym("ncclAllReduce"));
    if (!f) { gm_set_error("RCCL is not loaded in this process (ncclAllReduce not found)"); return GM_EINVAL; }
    const int rc = f(buf, buf, (size_t)n, /* ncclFloat32 */ 7, /* ncclSum */ 0, rccl_comm, (hipStream_t)stream);
    return rc ? nccl_fail(rc, "ncclAllReduce") : 0;
}

extern "C" int gm_rccl_comm_destroy(void* rccl_comm) {
    if (!rccl_comm) return 0;
    nccl_destroy_t f = reinterpret_cast<nccl_destroy_t>(nccl_sym("ncclCommDestroy"));
    if (f) (void)f(rccl_comm);
    return 0;
}
