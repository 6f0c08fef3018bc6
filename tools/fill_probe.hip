// fill_probe.hip -- how fast can ONE CU pull L2-resident bytes, as a function of waves per CU and loads in flight per
// wave?  (round 4's "one wave-wide vector-memory instruction per ~40 cycles per CU" was measured
// with four waves per CU; the shipped GEMM kernels run sixteen.)  Stand-alone: no library, no rocBLAS.
//
//   fill_probe.bin [reps]          prints one line per configuration: GB/s per CU, TB/s chip, cycles per instruction
//
// Every configuration is its own kernel instantiation (k_fill<MODE, WAVES, DEPTH, PAT>), so that a rocprofv3 --pmc
// pass attributes TA / TCP / TCC / SQ counters to it by name (tools/fill_law.sh runs the passes,
// tools/fill_law_report.py joins them with this program's timing into profiles/r05_fill_law.md).
//
// MODE  0: global_load_dwordx4 into VGPRs (consumed by an add, so the data really arrives in registers)
//       1: global_load_lds_dwordx4 (LDS-DMA; the landed bytes are not read back: the fill path alone)
// WAVES per workgroup; the launch is one workgroup per CU (256) or two (512, "x2": 2 x 8 waves)
// DEPTH loads in flight per wave (issue DEPTH, then: wait for the oldest, consume, issue the next)
// PAT   0: a wave instruction reads 1 KB contiguous
//       1: 16 rows x 64 B at a 3136-byte row stride, lane (r = lane & 15, g = lane >> 4) -> row r, bytes 16 g ..: the
//          16x16x4 fragment load of the k-contiguous 784-wide operands as shipped until round 4 (ADJACENT LANES read
//          DIFFERENT rows; the four lanes that share a row's 64 bytes are 16 lanes apart)
//       2: as 0, every workgroup inside its own 16 KB window (L1-resident: the TA/TCP/TD path without L2)
//       3: the same 16 rows x 64 B, lane -> row lane >> 2, bytes 16 (lane & 3) ..: adjacent lanes read adjacent bytes
//       4: 8 rows x 128 B (whole cache lines), lane -> row lane >> 3, bytes 16 (lane & 7) ..
//       5: the x-contiguous operand load as shipped until round 4 (gm_gemm.hip raw_xc4_16): k-row 4 (lane >> 4) +
//          (lane & 3), bytes 16 ((lane >> 2) & 3) ..: the four lanes of a QUAD read four different rows
// Footprint for PAT 0 / 1: all workgroups walk ONE shared 3 MB buffer (the D layer-1 GEMM's 2.85 MB of operands) from
// different offsets, 192 KB per pass -- L2 / MALL resident, the real kernels' situation.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../generative_models_amd/csrc/gm_ldsdma.h"

#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void gload16(f32x4& dst, const float* src) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
}
// the wait names the register it protects: the consumer below cannot be scheduled above it
template <int N> __device__ __forceinline__ void wait_for(f32x4& r) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(N) : "memory");
}

constexpr long BUF_BYTES = 3l << 20;         // shared operand buffer
constexpr int PASS_BYTES = 192 * 1024;       // (16 waves x 1 KB divides it)       // what one workgroup walks before it wraps (PAT 0 / 1)

template <int MODE, int WAVES, int DEPTH, int PAT>
__global__ __launch_bounds__(WAVES * 64) void k_fill(const float* __restrict__ src, int pieces_per_wave, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const char* base = reinterpret_cast<const char*>(src);
    // 32-bit offsets, advanced incrementally (no division in the loop: the address arithmetic must not be the bound)
    const uint32_t window = (PAT == 2) ? 16384u : (uint32_t)PASS_BYTES;
    const uint32_t wg_off = (PAT == 2) ? blockIdx.x * 16384u : (blockIdx.x * 98304u) % (uint32_t)(BUF_BYTES - PASS_BYTES);
    uint32_t off = (uint32_t)w * 1024u;          // PAT 0 / 2: byte offset of this wave's next piece inside the window
    int pc = w, pr = 0;                          // PAT 1: piece column (0..48) and row block (0..2)
    if (PAT == 1 || PAT == 3 || PAT == 5) { while (pc >= 49) { pc -= 49; pr = pr == 2 ? 0 : pr + 1; } }
    if (PAT == 4) { while (pc >= 24) { pc -= 24; pr = pr == 5 ? 0 : pr + 1; } }
    auto next = [&]() -> const float* {          // address of this wave's next piece, then advance by WAVES pieces
        uint32_t o;
        if (PAT == 1 || PAT == 3 || PAT == 5) {
            // 16 rows x 64 B; consecutive pieces walk along the rows (64 B further), 49 pieces per row block, then the
            // next 16 rows
            const int row = PAT == 1 ? (lane & 15) : PAT == 3 ? (lane >> 2) : (4 * (lane >> 4) + (lane & 3));
            const int b16 = PAT == 1 ? (lane >> 4) : PAT == 3 ? (lane & 3) : ((lane >> 2) & 3);
            o = (uint32_t)((pr * 16 + row) * 3136 + pc * 64 + b16 * 16);
            pc += WAVES;
            if (pc >= 49) { pc -= 49; pr = pr == 2 ? 0 : pr + 1; }
        } else if (PAT == 4) {
            // 8 rows x 128 B: 24 pieces per row block (3072 of the 3136 bytes), 6 row blocks of 8
            o = (uint32_t)((pr * 8 + (lane >> 3)) * 3136 + pc * 128 + (lane & 7) * 16);
            pc += WAVES;
            if (pc >= 24) { pc -= 24; pr = pr == 5 ? 0 : pr + 1; }
        } else {
            o = off + lane * 16u;
            off += WAVES * 1024u;
            if (off >= window) off -= window;
        }
        return reinterpret_cast<const float*>(base + wg_off + o);
    };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (MODE == 0) {
        f32x4 buf[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) gload16(buf[d], next());
        int i = DEPTH;
        for (; i + DEPTH <= pieces_per_wave; i += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                wait_for<DEPTH - 1>(buf[d]);
                acc += buf[d];
                gload16(buf[d], next());
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            wait_for<0>(buf[d]);
            acc += buf[d];
        }
    } else {
        // each wave owns DEPTH 1 KB landing slots
        const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds + (uint32_t)(w * DEPTH) * 1024u);   // wave-uniform (M0)
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) slab::glds16(next(), lds0 + d * 1024u);
        int i = DEPTH;
        for (; i + DEPTH <= pieces_per_wave; i += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                slab::wait_vm<DEPTH - 1>();
                slab::glds16(next(), lds0 + d * 1024u);
            }
        }
        slab::wait_vm<0>();
        acc[0] = lds[(w * DEPTH) * 256 + lane];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[blockIdx.x] = acc[0];     // never true: keeps the loads
}

struct Row { const char* name; double us, gbs_cu, tbs_chip, cyc_per_instr; };

static int g_reps = 20;
static double g_mhz = 2400.0;

template <int MODE, int WAVES, int DEPTH, int PAT>
static Row run(const char* name, const float* d_src, float* d_sink, int wgs, int kb_per_wg) {
    const int pieces_per_wave = (kb_per_wg / WAVES) / DEPTH * DEPTH;                 // 1 KB pieces
    const size_t lds_bytes = MODE == 1 ? (size_t)WAVES * DEPTH * 1024 : 0;
    auto k = k_fill<MODE, WAVES, DEPTH, PAT>;
    if (lds_bytes > 64 * 1024) HIPC(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipEvent_t e0, e1;
    HIPC(hipEventCreate(&e0)); HIPC(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(WAVES * 64), lds_bytes, 0, d_src, pieces_per_wave, d_sink);
    HIPC(hipDeviceSynchronize());
    HIPC(hipEventRecord(e0));
    for (int i = 0; i < g_reps; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(WAVES * 64), lds_bytes, 0, d_src, pieces_per_wave, d_sink);
    HIPC(hipEventRecord(e1));
    HIPC(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPC(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / g_reps;
    const double bytes_wg = (double)pieces_per_wave * WAVES * 1024.0;
    const double per_cu = bytes_wg * (wgs > 256 ? wgs / 256.0 : 1.0);               // per ACTIVE CU (256 CUs)
    Row r;
    r.name = name; r.us = us; r.gbs_cu = per_cu / us * 1e-3; r.tbs_chip = per_cu * (wgs > 256 ? 256.0 : wgs) / us * 1e-6;
    r.cyc_per_instr = us * g_mhz / (per_cu / 1024.0);
    printf("%-34s wgs %3d waves/wg %2d depth %d  %8.1f us  %6.1f GB/s per CU  %5.2f TB/s chip  %5.1f cycles per wave instruction (at %.0f MHz)\n",
           name, wgs, WAVES, DEPTH, us, r.gbs_cu, r.tbs_chip, r.cyc_per_instr, g_mhz);
    fflush(stdout);
    return r;
}

int main(int argc, char** argv) {
    if (argc > 1) g_reps = atoi(argv[1]);
    hipDeviceProp_t pr;
    HIPC(hipGetDeviceProperties(&pr, 0));
    g_mhz = pr.clockRate / 1000.0;
    printf("device %s  CUs %d  clock %.0f MHz  L2 %d KB\n", pr.gcnArchName, pr.multiProcessorCount, g_mhz, pr.l2CacheSize / 1024);
    float *d_src, *d_sink;
    HIPC(hipMalloc(&d_src, BUF_BYTES + (256 * 16384)));
    HIPC(hipMalloc(&d_sink, 4096 * sizeof(float)));
    std::vector<float> h((BUF_BYTES + 256 * 16384) / 4, 1.0f);
    HIPC(hipMemcpy(d_src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int KB = 6144;                                                            // bytes per workgroup per launch: 6 MB
#define RUN(M, W, D, P, WGS) run<M, W, D, P>(#M "/" #W "w/d" #D "/p" #P "/" #WGS, d_src, d_sink, WGS, (WGS) > 256 ? KB * 256 / (WGS) : KB)
    const bool all = argc > 2 && !strcmp(argv[2], "all");
    if (all) {
        // VGPR loads: waves per CU x depth
        RUN(0, 4, 2, 0, 256); RUN(0, 4, 4, 0, 256); RUN(0, 4, 8, 0, 256);
        RUN(0, 8, 2, 0, 256); RUN(0, 8, 4, 0, 256); RUN(0, 8, 8, 0, 256);
        RUN(0, 16, 2, 0, 256); RUN(0, 16, 4, 0, 256); RUN(0, 16, 8, 0, 256);
        // two workgroups of 8 waves per CU
        RUN(0, 8, 4, 0, 512);
        // LDS-DMA
        RUN(1, 4, 2, 0, 256); RUN(1, 4, 4, 0, 256); RUN(1, 4, 8, 0, 256);
        RUN(1, 8, 2, 0, 256); RUN(1, 8, 4, 0, 256); RUN(1, 8, 8, 0, 256);
        RUN(1, 16, 2, 0, 256); RUN(1, 16, 4, 0, 256); RUN(1, 16, 8, 0, 256);
        // L1-resident: the TA / TCP / TD path alone
        RUN(0, 4, 4, 2, 256); RUN(0, 16, 4, 2, 256); RUN(1, 16, 4, 2, 256);
        // fewer CUs active (is it a per-CU or a shared-fabric bound?)
        RUN(0, 16, 4, 0, 32); RUN(1, 16, 4, 0, 32);
    }
    // the access PATTERN of one wave instruction, 16 waves x 4 in flight = the shipped kernels' window
    RUN(0, 16, 4, 0, 256); RUN(0, 16, 4, 1, 256); RUN(0, 16, 4, 3, 256); RUN(0, 16, 4, 4, 256); RUN(0, 16, 4, 5, 256);
    RUN(1, 16, 4, 1, 256); RUN(1, 16, 4, 3, 256); RUN(1, 16, 4, 4, 256);
    RUN(0, 4, 2, 0, 256);
    return 0;
}
