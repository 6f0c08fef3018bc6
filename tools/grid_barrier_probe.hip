// Round-2 prerequisite measurement: what does an agent-scope grid
// barrier cost on MI355X when 256 workgroups (one per CU, 8 XCDs) take part?  Standalone program:
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o tools/grid_barrier_probe.bin
// Prints us per barrier for (a) the bare barrier, (b) barrier + every workgroup writing 16 KB that a
// workgroup on another XCD reads after the barrier (the producer/consumer pattern of a persistent
// per-iteration kernel), for 1024- and 256-thread workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned nblocks, unsigned& epoch,
                                             unsigned* err) {
    __shared__ int ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                           // release this phase's writes
        const unsigned target = (++epoch) * nblocks;
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        ok = 1;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 20000000) { ok = 0; *err = 1; break; }   // never hang the box
        }
        __threadfence();                                           // acquire the others' writes
    }
    __syncthreads();
    return ok != 0;
}

// Round 4: the same barrier with the polling the exchange kernels of csrc/gm_comm.hip ended up with.
//   MODE 1: one counter, RELAXED agent-scope polling loads, ONE acquire fence after the wait (round 2 polled with
//           acquire loads: an L2 invalidate per poll);
//   MODE 2: counter + epoch flag -- the last arriver publishes the epoch, everyone else polls the flag with relaxed
//           loads (no read-modify-write traffic on the polled line);
//   MODE 3: per-XCD counters (workgroup b runs on XCD b % 8): 32 arrivals on an XCD-local line, the XCD's last
//           arriver arrives on the chip counter, the chip's last arriver publishes the epoch flag.
// ctr layout: [0] chip counter, [16] epoch flag, [32 + 16 x] XCD x's counter (64-byte lines apart).
template <int MODE>
__device__ __forceinline__ bool grid_barrier_v(unsigned* ctr, unsigned nblocks, unsigned& epoch, unsigned* err) {
    __shared__ int ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                          // the workgroup's writes of this phase (one wave's
                                                                  // L2 write-back covers them: they are past the barrier)
        ++epoch;
        ok = 1;
        int spins = 0;
        if (MODE == 1) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = epoch * nblocks;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 20000000) { ok = 0; *err = 1; break; }
            }
        } else {
            bool last;
            if (MODE == 2) {
                last = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == epoch * nblocks - 1;
            } else {
                const unsigned x = blockIdx.x & 7, per = nblocks / 8;
                last = false;
                if (__hip_atomic_fetch_add(ctr + 32 + 16 * x, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == epoch * per - 1)
                    last = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == epoch * 8 - 1;
            }
            if (last) {
                __hip_atomic_store(ctr + 16, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_load(ctr + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 20000000) { ok = 0; *err = 1; break; }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok != 0;
}

template <int MODE>
__global__ void probe_v(unsigned* ctr, unsigned* err, float* buf, int words, int phases, float* sink) {
    unsigned epoch = 0;
    const unsigned nb = gridDim.x;
    float acc = 0.f;
    for (int ph = 0; ph < phases; ++ph) {
        float* mine = buf + ((size_t)(ph & 1) * nb + blockIdx.x) * words;
        for (int i = threadIdx.x; i < words; i += blockDim.x) mine[i] = (float)(ph + i);
        if (!grid_barrier_v<MODE>(ctr, nb, epoch, err)) return;
        const float* theirs = buf + ((size_t)(ph & 1) * nb + (blockIdx.x + nb / 2 + 1) % nb) * words;
        for (int i = threadIdx.x; i < words; i += blockDim.x)
            acc += __builtin_nontemporal_load(theirs + i);        // (never a stale line of this XCD's L2)
    }
    if (acc == -1.f) sink[0] = acc;
}

// words: floats exchanged per workgroup per phase (0 = bare barrier)
__global__ void probe(unsigned* ctr, unsigned* err, float* buf, int words, int phases, float* sink) {
    unsigned epoch = 0;
    const unsigned nb = gridDim.x;
    float acc = 0.f;
    for (int ph = 0; ph < phases; ++ph) {
        float* mine = buf + ((size_t)(ph & 1) * nb + blockIdx.x) * words;
        for (int i = threadIdx.x; i < words; i += blockDim.x) mine[i] = (float)(ph + i);
        if (!grid_barrier(ctr, nb, epoch, err)) return;
        // read what the workgroup "across the chip" wrote (block b and b+nb/2 sit on different XCDs)
        const float* theirs = buf + ((size_t)(ph & 1) * nb + (blockIdx.x + nb / 2 + 1) % nb) * words;
        for (int i = threadIdx.x; i < words; i += blockDim.x) acc += theirs[i];
    }
    if (acc == -1.f) sink[0] = acc;
}

int main() {
    unsigned *ctr, *err;
    float *buf, *sink;
    const int nb = 256, maxw = 4096;
    CK(hipMalloc(&ctr, 4 * 256)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&buf, sizeof(float) * 2 * nb * maxw));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int phases = 2000;
    for (int threads : {1024, 256}) {
        for (int words : {0, 1024, 4096}) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(ctr, 0, 4)); CK(hipMemset(err, 0, 4));
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(probe, dim3(nb), dim3(threads), 0, 0, ctr, err, buf, words, phases, sink);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            unsigned h = 0;
            CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
            printf("threads/wg %4d  exchange %5d B/wg/phase : %7.3f us per phase%s\n", threads,
                   words * 4, best * 1e3f / phases, h ? "  (SPIN LIMIT HIT)" : "");
        }
    }
    for (int mode : {1, 2, 3}) {
        for (int threads : {1024, 256}) {
            for (int words : {0, 1024}) {
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemset(ctr, 0, 4 * 256)); CK(hipMemset(err, 0, 4));
                    CK(hipEventRecord(e0));
                    if (mode == 1) hipLaunchKernelGGL(probe_v<1>, dim3(nb), dim3(threads), 0, 0, ctr, err, buf, words, phases, sink);
                    else if (mode == 2) hipLaunchKernelGGL(probe_v<2>, dim3(nb), dim3(threads), 0, 0, ctr, err, buf, words, phases, sink);
                    else hipLaunchKernelGGL(probe_v<3>, dim3(nb), dim3(threads), 0, 0, ctr, err, buf, words, phases, sink);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms = 0.f;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                unsigned h = 0;
                CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
                printf("mode %d  threads/wg %4d  exchange %5d B/wg/phase : %7.3f us per phase%s\n", mode, threads,
                       words * 4, best * 1e3f / phases, h ? "  (SPIN LIMIT HIT)" : "");
            }
        }
    }
    // reference point: an empty kernel launched back to back (what a launch boundary costs eagerly)
    CK(hipEventRecord(e0));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, 0, ctr, err, buf, 0, 0, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("empty 256-workgroup kernel, back-to-back eager launches: %7.3f us each\n", ms * 1e3f / 200);
    return 0;
}
