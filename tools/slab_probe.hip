// slab_probe.hip -- stand-alone check + timing of the slab GEMM family (tools/gm_slab.h) against an fp64 host
// reference, the shipped split-reduction kernels (libgm_hip.so through the C-ABI) and the vendor's sgemm, all in one
// process on one box.  Build: tools/build_slab_probe.sh ; run on the GPU box: tools/slab_probe.bin [reps]
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "gm_slab.h"
#include "../include/gm_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

using namespace slab;

struct StoreEpi {
    float* C; int ldc, M, N, m0, n0;
    __device__ __forceinline__ void operator()(int row, int c4, float4 v) const {
        const int m = m0 + row, n = n0 + 4 * c4;
        if (m >= M || n >= N) return;
        float* o = C + (int64_t)m * ldc + n;
        if (n + 3 < N) { *reinterpret_cast<float4*>(o) = v; return; }
        o[0] = v.x; if (n + 1 < N) o[1] = v.y; if (n + 2 < N) o[2] = v.z;
    }
};

template <int TMW, int TNW, int NBUF, int KU, int ABL = 0>
__global__ __launch_bounds__(256) void k_dw(CoreP p, float* C, int ldc) {
    using Cf = DwCfg<TMW, TNW, NBUF, KU>;
    __shared__ __attribute__((aligned(16))) float lds[Cf::LDS_FLOATS];
    int tile, s;
    if (!map_block(p, tile, s)) return;
    StoreEpi epi{C, ldc, p.M, p.N, (tile / p.tn) * Cf::BM, (tile % p.tn) * Cf::BN};
    dw_tile<TMW, TNW, NBUF, KU, StoreEpi, NoAXf, ABL>(p, lds, tile, s, epi, NoAXf());
}
template <int TMW, int TNW, int NBUF>
__global__ __launch_bounds__(256) void k_fw(CoreP p, float* C, int ldc) {
    using Cf = FwCfg<TMW, TNW, NBUF>;
    __shared__ __attribute__((aligned(16))) float lds[Cf::LDS_FLOATS];
    int tile, s;
    if (!map_block(p, tile, s)) return;
    StoreEpi epi{C, ldc, p.M, p.N, (tile / p.tn) * Cf::BM, (tile % p.tn) * Cf::BN};
    fwd_tile<TMW, TNW, NBUF>(p, lds, tile, s, epi);
}


// ---- per-CU fill-rate microbenchmark: how fast can one workgroup pull bytes, by LDS-DMA or into VGPRs, depending on
// where they come from (footprint) and how a 1 KB piece is laid out in memory (contiguous / split over rows) -------
// KIND 0: global_load_lds_dwordx4, 1: global_load_dwordx4 into VGPRs.  Every wave keeps DEPTH batches of PP pieces in
// flight.  ROWB: bytes per contiguous row segment of a piece (1024: fully contiguous; 320: DW-like rows).
template <int KIND, int PP, int DEPTH>
__global__ __launch_bounds__(256) void k_fill(const float* src, long wg_stride_f, long wrap_f, int iters, int rowb, long row_stride_f, float* sink) {
    __shared__ __attribute__((aligned(16))) float lds[4 * DEPTH * PP * 256];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;
    const float* base = src + (long)blockIdx.x * wg_stride_f;
    // lane's float offset inside a 1 KB piece laid out as rows of rowb bytes
    const int lane_b = lane * 16;
    const long lane_off = (long)(lane_b / rowb) * row_stride_f + (lane_b % rowb) / 4;
    const long piece_f = (1024 / rowb) * row_stride_f;          // floats advanced per piece (rowb < 1024) ...
    const long piece_adv = rowb >= 1024 ? 256 : piece_f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    long pos = (long)w * PP * piece_adv;                        // this wave's running piece position (floats)
    const long wave_adv = 4L * PP * piece_adv;
    auto addr = [&](long ppos, int j) { return base + ((ppos + (long)j * piece_adv) % wrap_f) + lane_off; };
    if constexpr (KIND == 0) {
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int j = 0; j < PP; ++j) glds16(addr(pos, j), lds_base + ((w * DEPTH + d) * PP + j) * 1024u);
            pos += wave_adv;
        }
        int d = 0;
        for (int it = DEPTH; it < iters; ++it) {
            wait_vm<PP*(DEPTH - 1)>();
#pragma unroll
            for (int j = 0; j < PP; ++j) glds16(addr(pos, j), lds_base + ((w * DEPTH + d) * PP + j) * 1024u);
            pos += wave_adv;
            d = (d + 1 == DEPTH) ? 0 : d + 1;
        }
        wait_vm<0>();
        acc.x = lds[t];
    } else if constexpr (KIND >= 2) {
        // KIND = 2, 3: 8 / 12 bytes per lane (the interleaved-fragment loads of the shipped weight-gradient kernel);
        // a "piece" is still one wave instruction, now 512 / 768 bytes
        constexpr int W = KIND;
        struct __attribute__((aligned(4))) V { float v[W]; };
        V r[DEPTH][PP];
        const long lo = (long)lane * W;
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int j = 0; j < PP; ++j) r[d][j] = *reinterpret_cast<const V*>(base + ((pos + (long)j * 256) % wrap_f) + lo);
            pos += wave_adv;
        }
        for (int it = DEPTH; it < iters; it += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
                for (int j = 0; j < PP; ++j) {
#pragma unroll
                    for (int e = 0; e < W; ++e) acc.x += r[d][j].v[e];
                    r[d][j] = *reinterpret_cast<const V*>(base + ((pos + (long)j * 256) % wrap_f) + lo);
                }
                pos += wave_adv;
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int j = 0; j < PP; ++j) acc.y += r[d][j].v[0];
    } else {
        float4 r[DEPTH][PP];
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int j = 0; j < PP; ++j) r[d][j] = *reinterpret_cast<const float4*>(addr(pos, j));
            pos += wave_adv;
        }
        for (int it = DEPTH; it < iters; it += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
                for (int j = 0; j < PP; ++j) { acc = add4(acc, r[d][j]); r[d][j] = *reinterpret_cast<const float4*>(addr(pos, j)); }
                pos += wave_adv;
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int j = 0; j < PP; ++j) acc = add4(acc, r[d][j]);
    }
    if (acc.x == 123.456f) sink[t] = acc.x + acc.y + acc.z + acc.w;
}

// ---- instruction-fetch microbenchmark: the same number of dependent v_fma executed as straight-line code (every
// 64-byte line fetched once, cold after the kernel boundary) or as a loop over a 64-instruction body (hot) ---------
template <int N, bool LOOP>
__global__ __launch_bounds__(256) void k_icache(float* out, float a, float b) {
    float x = a + threadIdx.x;
    if constexpr (LOOP) {
        for (int i = 0; i < N / 64; ++i) {
#pragma unroll
            for (int j = 0; j < 64; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
        }
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    }
    if (x == 123.456f) out[threadIdx.x] = x;
}

static hipStream_t g_stream;
static int g_reps = 20;
static unsigned long long* g_trace = nullptr;   // device, 8 stamps x 2048 workgroups
static bool g_want_trace = false, g_nocheck = false;

// per-launch microseconds of `fn` captured `g_reps` times into one graph (best of 5 replays)
template <class F> static double time_graph(F fn) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(g_stream, hipStreamCaptureModeGlobal));
    for (int i = 0; i < g_reps; ++i) fn();
    CK(hipStreamEndCapture(g_stream, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) CK(hipGraphLaunch(ge, g_stream));
    CK(hipStreamSynchronize(g_stream));
    double best = 1e30;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, g_stream));
        CK(hipGraphLaunch(ge, g_stream));
        CK(hipEventRecord(e1, g_stream));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, (double)ms * 1e3 / g_reps);
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return best;
}

struct Prob {
    int mode, M, N, K;              // output M x N, reduction K
    std::vector<float> hA, hB; std::vector<double> ref;
    float *dA, *dB, *dC; float* ws; unsigned* cnt; unsigned* err;
};

static void fill(std::vector<float>& v, unsigned seed) {
    unsigned x = seed * 2654435761u + 12345u;
    for (auto& f : v) { x = x * 1664525u + 1013904223u; f = ((x >> 8) & 0xffff) / 32768.0f - 1.0f; }
}

static void make_prob(Prob& P, int mode, int M, int N, int K) {
    P.mode = mode; P.M = M; P.N = N; P.K = K;
    P.hA.resize((size_t)M * K); P.hB.resize((size_t)N * K); P.ref.assign((size_t)M * N, 0.0);
    fill(P.hA, 1 + M + K); fill(P.hB, 7 + N + K);
    // layouts: DW: A[K][M], B[K][N];  FWD: A[M][K], B[N][K]
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double acc = 0;
            if (mode == DW) for (int k = 0; k < K; ++k) acc += (double)P.hA[(size_t)k * M + m] * P.hB[(size_t)k * N + n];
            else for (int k = 0; k < K; ++k) acc += (double)P.hA[(size_t)m * K + k] * P.hB[(size_t)n * K + k];
            P.ref[(size_t)m * N + n] = acc;
        }
    CK(hipMalloc(&P.dA, P.hA.size() * 4)); CK(hipMalloc(&P.dB, P.hB.size() * 4)); CK(hipMalloc(&P.dC, (size_t)M * N * 4 + 64));
    CK(hipMemcpy(P.dA, P.hA.data(), P.hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(P.dB, P.hB.data(), P.hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&P.ws, 64u << 20)); CK(hipMalloc(&P.cnt, 4096 * 8)); CK(hipMalloc(&P.err, 4));
    CK(hipMemset(P.cnt, 0, 4096 * 8)); CK(hipMemset(P.err, 0, 4));
}

static double check(Prob& P, const char* what) {
    std::vector<float> out((size_t)P.M * P.N);
    CK(hipMemcpy(out.data(), P.dC, out.size() * 4, hipMemcpyDeviceToHost));
    double maxe = 0, maxr = 0;
    for (size_t i = 0; i < out.size(); ++i) { maxe = std::max(maxe, std::fabs(out[i] - P.ref[i])); maxr = std::max(maxr, std::fabs(P.ref[i])); }
    unsigned e; CK(hipMemcpy(&e, P.err, 4, hipMemcpyDeviceToHost));
    const double rel = maxe / maxr;
    if (rel > 3e-6 * std::sqrt((double)P.K) || e || rel != rel) printf("   !!! %s WRONG: rel err %.3e  wait-flag %u\n", what, rel, e);
    return rel;
}

template <class KF> static void run_variant(Prob& P, const char* name, KF kern, int BM, int BN, int S, int xmap) {
    CoreP p{};
    p.A = P.dA; p.B = P.dB; p.M = P.M; p.N = P.N; p.K = P.K;
    p.lda = (P.mode == DW) ? P.M : P.K; p.ldb = (P.mode == DW) ? P.N : P.K;
    p.tm = (P.M + BM - 1) / BM; p.tn = (P.N + BN - 1) / BN; p.xmap = xmap;
    p.sp.S = S; p.sp.ws = P.ws; p.sp.cnt = P.cnt; p.sp.err = P.err; p.sp.trace = nullptr;
    const int T = p.tm * p.tn;
    int grid = T * S;
    if (xmap) { const int g = 8 / S; grid = 8 * ((T + g - 1) / g); }
    if (T * S > 256) { printf("  %-34s skipped: %d workgroups\n", name, T * S); return; }
    CK(hipMemsetAsync(P.dC, 0xff, (size_t)P.M * P.N * 4, g_stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, g_stream, p, P.dC, P.N);
    CK(hipStreamSynchronize(g_stream));
    const double rel = g_nocheck ? 0 : check(P, name);
    const double us = time_graph([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, g_stream, p, P.dC, P.N); });
    const double rel2 = g_nocheck ? 0 : check(P, name);
    printf("  %-34s T=%3d S=%d wg=%3d  %7.2f us  %6.1f TF  err %.1e/%.1e\n", name, T, S, grid, us,
           2.0 * P.M * P.N * P.K / us * 1e-6, rel, rel2);
    if (g_want_trace) {
        // one traced launch: stamps relative to the earliest start, averaged over workgroups (us at 100 MHz ticks?)
        CK(hipMemset(g_trace, 0, 2048 * 8 * 8));
        p.sp.trace = g_trace;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, g_stream, p, P.dC, P.N);
        CK(hipStreamSynchronize(g_stream));
        std::vector<unsigned long long> tr(2048 * 8);
        CK(hipMemcpy(tr.data(), g_trace, tr.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < grid; ++b) if (tr[b * 8]) t0 = std::min(t0, tr[b * 8]);
        double avg[8] = {0}, mx[8] = {0}; int n = 0;
        for (int b = 0; b < grid; ++b) {
            if (!tr[b * 8]) continue;
            ++n;
            for (int i = 0; i < 8; ++i) { const double d = tr[b * 8 + i] ? (double)(tr[b * 8 + i] - t0) : 0; avg[i] += d; mx[i] = std::max(mx[i], d); }
        }
        printf("      trace (ticks since first start; avg | max over %d WGs):", n);
        for (int i = 0; i < 8; ++i) printf("  [%d] %.0f|%.0f", i, avg[i] / n, mx[i]);
        printf("\n");
        p.sp.trace = nullptr;
    }
    fflush(stdout);
}

static gm_slot noslot() { gm_slot z; memset(&z, 0, sizeof z); return z; }

static void baselines(Prob& P, rocblas_handle h) {
    // shipped kernels through the C-ABI (layer terms: rows, in, out)
    if (P.mode == DW) {
        // dW[N_layer = M][K_layer = N] from dA[rows = K][M], X[rows][N]
        auto fn = [&] { gm_linear_bwd_dw(g_stream, P.dA, P.M, P.dB, P.N, noslot(), P.dC, nullptr, P.K, P.N, P.M, 0); };
        fn(); CK(hipStreamSynchronize(g_stream));
        const double rel = check(P, "shipped dw");
        printf("  %-34s %28s %7.2f us  %6.1f TF  err %.1e\n", "shipped gm_linear_bwd_dw", "", time_graph(fn),
               2.0 * P.M * P.N * P.K / time_graph(fn) * 1e-6, rel);
    } else {
        auto fn = [&] { gm_linear_fwd(g_stream, P.dA, P.K, noslot(), P.dB, nullptr, P.dC, P.N, P.M, P.K, P.N, GM_ACT_ID); };
        fn(); CK(hipStreamSynchronize(g_stream));
        const double rel = check(P, "shipped fwd");
        printf("  %-34s %28s %7.2f us  %6.1f TF  err %.1e\n", "shipped gm_linear_fwd", "", time_graph(fn),
               2.0 * P.M * P.N * P.K / time_graph(fn) * 1e-6, rel);
    }
    const float one = 1.f, zero = 0.f;
    auto vf = [&] {
        if (P.mode == DW) rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_transpose, P.N, P.M, P.K, &one, P.dB, P.N, P.dA, P.M, &zero, P.dC, P.N);
        else rocblas_sgemm(h, rocblas_operation_transpose, rocblas_operation_none, P.N, P.M, P.K, &one, P.dB, P.K, P.dA, P.K, &zero, P.dC, P.N);
    };
    vf(); CK(hipStreamSynchronize(g_stream));
    const double rel = check(P, "rocblas");
    const double us = time_graph(vf);
    printf("  %-34s %28s %7.2f us  %6.1f TF  err %.1e\n", "vendor rocblas_sgemm", "", us, 2.0 * P.M * P.N * P.K / us * 1e-6, rel);
    fflush(stdout);
}

#define DWV(TM_, TN_, NB_, KU_, S_, X_) run_variant(P, "dw " #TM_ "x" #TN_ " nbuf" #NB_ " ku" #KU_ " S" #S_ " x" #X_, k_dw<TM_, TN_, NB_, KU_>, 16 * TM_, 16 * TN_, S_, X_)
#define FWV(TM_, TN_, NB_, S_, X_) run_variant(P, "fw " #TM_ "x" #TN_ " nbuf" #NB_ " S" #S_ " x" #X_, k_fw<TM_, TN_, NB_>, 64 * TM_, 16 * TN_, S_, X_)

int main(int argc, char** argv) {
    if (argc > 1) g_reps = atoi(argv[1]);
    const char* only = argc > 2 ? argv[2] : "";
    CK(hipStreamCreate(&g_stream));
    CK(hipMalloc(&g_trace, 2048 * 8 * 8));
    rocblas_handle h; rocblas_create_handle(&h); rocblas_set_stream(h, g_stream);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  CUs %d  clock %d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);



    if (*only && strstr(only, "trace")) {
        g_want_trace = true;
        const int rows[] = {2048, 256};
        for (int K : rows) {
            Prob P; make_prob(P, DW, 400, 784, K);
            printf("DW 400x784 over %d rows (trace)\n", K);
            DWV(5, 7, 4, 1, 7, 0); DWV(5, 7, 4, 1, 1, 0); DWV(5, 9, 4, 1, 8, 1);
            g_nocheck = true;
            run_variant(P, "dw 5x7 nbuf4 S7 NO DMA", k_dw<5, 7, 4, 1, 1>, 80, 112, 7, 0);
            run_variant(P, "dw 5x7 nbuf4 S7 NO MFMA", k_dw<5, 7, 4, 1, 2>, 80, 112, 7, 0);
            run_variant(P, "dw 5x7 nbuf4 S1 NO DMA", k_dw<5, 7, 4, 1, 1>, 80, 112, 1, 0);
            run_variant(P, "dw 5x7 nbuf4 S1 NO MFMA", k_dw<5, 7, 4, 1, 2>, 80, 112, 1, 0);
            g_nocheck = false;
        }
        // what a tick is: time a known-length kernel? report clock rates instead
        int wall = 0; hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0);
        printf("wall clock rate %d kHz, shader clock %d kHz\n", wall, prop.clockRate);
        return 0;
    }

    if (*only && strstr(only, "icache")) {
        float* sink; CK(hipMalloc(&sink, 4096));
        auto run = [&](const char* nm, auto kern, int grid) {
            auto fn = [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, g_stream, sink, 1.0f, 0.5f); };
            fn(); CK(hipStreamSynchronize(g_stream));
            printf("  %-40s grid %4d  %7.2f us\n", nm, grid, time_graph(fn)); fflush(stdout);
        };
        for (int grid : {256, 8}) {
            run("empty-ish: 64 fma straight", k_icache<64, false>, grid);
            run("512 fma straight (4 KB)", k_icache<512, false>, grid);
            run("512 fma loop", k_icache<512, true>, grid);
            run("2048 fma straight (16 KB)", k_icache<2048, false>, grid);
            run("2048 fma loop", k_icache<2048, true>, grid);
            run("8192 fma straight (64 KB)", k_icache<8192, false>, grid);
            run("8192 fma loop", k_icache<8192, true>, grid);
        }
        return 0;
    }

    if (*only && strstr(only, "small")) {
        const int rows[] = {2048, 1024, 512, 256};
        for (int K : rows) {
            Prob P; make_prob(P, DW, 400, 784, K);
            printf("DW 400x784 over %d rows, no cross-workgroup split\n", K);
            baselines(P, h);
            DWV(2, 3, 2, 4, 1, 0); DWV(2, 3, 3, 4, 1, 0); DWV(2, 3, 3, 2, 1, 0); DWV(2, 3, 4, 2, 1, 0); DWV(2, 3, 4, 1, 1, 0);
            DWV(2, 4, 3, 4, 1, 0); DWV(3, 3, 3, 4, 1, 0); DWV(3, 4, 3, 2, 1, 0); DWV(3, 4, 3, 4, 1, 0); DWV(4, 4, 3, 2, 1, 0);
            CK(hipFree(P.dA)); CK(hipFree(P.dB)); CK(hipFree(P.dC)); CK(hipFree(P.ws)); CK(hipFree(P.cnt)); CK(hipFree(P.err));
        }
        return 0;
    }
    if (*only && strstr(only, "fill")) {
        float* buf; const size_t BYTES = 1ull << 30;
        CK(hipMalloc(&buf, BYTES + (16 << 20))); CK(hipMemset(buf, 0, BYTES + (16 << 20)));
        float* sink; CK(hipMalloc(&sink, 4096));
        struct Case { const char* name; long wg_stride_b; long wrap_b; int rowb; long row_stride_b; int grid; };
        const Case cases_all[] = {
            {"own 16 KB per WG (L1/L2), contiguous", 16 << 10, 16 << 10, 1024, 0, 256},
            {"own 64 KB per WG (L2), contiguous", 64 << 10, 64 << 10, 1024, 0, 256},
            {"own 4 MB per WG (HBM stream), contiguous", 4 << 20, 4 << 20, 1024, 0, 256},
            {"ALL share 1 MB (L2 after first touch), contiguous", 0, 1 << 20, 1024, 0, 256},
            {"ALL share 8 MB, contiguous", 0, 8 << 20, 1024, 0, 256},
            {"own 64 KB per WG, rows of 320 B / 1600 B stride", 64 << 10, 64 << 10, 320, 1600, 256},
            {"ALL share 1 MB, rows of 320 B / 1600 B stride", 0, 1 << 20, 320, 1600, 256},
            {"own 64 KB per WG (L2), contiguous, 32 WGs", 64 << 10, 64 << 10, 1024, 0, 32},
            {"own 4 MB per WG (HBM), contiguous, 32 WGs", 4 << 20, 4 << 20, 1024, 0, 32},
        };
        const Case cases_rows[] = {
            {"share 2 MB, rows of 64 B, stride 3136 B", 0, 2 << 20, 64, 3136, 256},
            {"share 2 MB, rows of 128 B, stride 3136 B", 0, 2 << 20, 128, 3136, 256},
            {"share 2 MB, rows of 128 B, stride 3200 B (aligned)", 0, 2 << 20, 128, 3200, 256},
            {"share 2 MB, rows of 256 B, stride 3136 B", 0, 2 << 20, 256, 3136, 256},
            {"share 2 MB, rows of 256 B, stride 3200 B (aligned)", 0, 2 << 20, 256, 3200, 256},
            {"share 2 MB, rows of 512 B, stride 3136 B", 0, 2 << 20, 512, 3136, 256},
            {"share 2 MB, rows of 192 B, stride 3136 B", 0, 2 << 20, 192, 3136, 256},
            {"share 2 MB, contiguous", 0, 2 << 20, 1024, 0, 256},
        };
        const bool rows_only = strstr(only, "fillrows") != nullptr;
        const Case* cases = rows_only ? cases_rows : cases_all;
        const int ncases = rows_only ? 8 : 9;
        for (int ci = 0; ci < ncases; ++ci) {
            const Case& c = cases[ci];
            const int iters = 256;
            auto run = [&](const char* kn, auto kern, int PP) {
                auto fn = [&] { hipLaunchKernelGGL(kern, dim3(c.grid), dim3(256), 0, g_stream, buf, c.wg_stride_b / 4, c.wrap_b / 4, iters, c.rowb, c.row_stride_b / 4, sink); };
                fn(); CK(hipStreamSynchronize(g_stream));
                const double us = time_graph(fn);
                const double bytes = (double)c.grid * iters * 4 * PP * 1024;
                printf("  %-52s %-22s %8.2f us  %7.1f GB/s per CU  %6.2f TB/s total\n", c.name, kn, us, bytes / c.grid / us * 1e-3, bytes / us * 1e-6);
                fflush(stdout);
            };
            run("lds-dma 3pc depth2", k_fill<0, 3, 2>, 3);
            run("lds-dma 3pc depth4", k_fill<0, 3, 4>, 3);
            run("lds-dma 6pc depth4", k_fill<0, 6, 4>, 6);
            run("vgpr 3pc depth2", k_fill<1, 3, 2>, 3);
            run("vgpr 3pc depth4", k_fill<1, 3, 4>, 3);
            run("vgpr 6pc depth4", k_fill<1, 6, 4>, 6);
            if (rows_only && ci == 7) {
                auto runw = [&](const char* kn, auto kern, int PP, int W) {
                    auto fn = [&] { hipLaunchKernelGGL(kern, dim3(c.grid), dim3(256), 0, g_stream, buf, c.wg_stride_b / 4, c.wrap_b / 4, iters, c.rowb, c.row_stride_b / 4, sink); };
                    fn(); CK(hipStreamSynchronize(g_stream));
                    const double us = time_graph(fn);
                    const double bytes = (double)c.grid * iters * 4 * PP * 64 * 4 * W;
                    printf("  %-52s %-22s %8.2f us  %7.1f GB/s per CU  %6.2f TB/s total\n", c.name, kn, us, bytes / c.grid / us * 1e-3, bytes / us * 1e-6);
                };
                runw("vgpr 8 B/lane 6 instr depth4", k_fill<2, 6, 4>, 6, 2);
                runw("vgpr 12 B/lane 6 instr depth4", k_fill<3, 6, 4>, 6, 3);
                runw("vgpr 8 B/lane 3 instr depth4", k_fill<2, 3, 4>, 3, 2);
                runw("vgpr 12 B/lane 3 instr depth4", k_fill<3, 3, 4>, 3, 3);
            }
        }
        return 0;
    }
    if (!*only || strstr(only, "dw")) {
        const int rows[] = {2048, 1024, 512, 256};
        for (int K : rows) {
            Prob P; make_prob(P, DW, 400, 784, K);
            printf("DW 400x784 over %d rows\n", K);
            baselines(P, h);
            DWV(5, 7, 4, 1, 7, 0); DWV(5, 7, 4, 1, 4, 0); DWV(5, 7, 4, 1, 2, 0); DWV(5, 7, 4, 1, 1, 0);
            DWV(5, 7, 2, 1, 7, 0); DWV(5, 7, 3, 2, 7, 0); DWV(5, 7, 2, 2, 7, 0);
            DWV(5, 9, 4, 1, 8, 1); DWV(5, 9, 4, 1, 8, 0); DWV(5, 9, 3, 2, 8, 1); DWV(5, 9, 4, 1, 4, 1);
            DWV(4, 4, 4, 1, 2, 0); DWV(4, 4, 4, 2, 2, 0);
            CK(hipFree(P.dA)); CK(hipFree(P.dB)); CK(hipFree(P.dC)); CK(hipFree(P.ws)); CK(hipFree(P.cnt)); CK(hipFree(P.err));
        }
        {
            Prob P; make_prob(P, DW, 784, 400, 256);
            printf("DW 784x400 over 256 rows\n");
            baselines(P, h);
            DWV(7, 5, 4, 1, 7, 0); DWV(7, 5, 4, 1, 4, 0); DWV(7, 5, 4, 1, 2, 0);
        }
    }
    if (!*only || strstr(only, "fw")) {
        const int ms[] = {2048, 512, 256};
        for (int M : ms) {
            Prob P; make_prob(P, FWD, M, 400, 784);
            printf("FWD %dx784 -> 400\n", M);
            baselines(P, h);
            FWV(1, 5, 4, 6, 0); FWV(1, 5, 4, 4, 0); FWV(1, 5, 4, 3, 0); FWV(1, 5, 4, 2, 0); FWV(1, 5, 4, 1, 0);
            FWV(1, 5, 2, 6, 0); FWV(1, 5, 3, 6, 0);
            FWV(2, 5, 3, 3, 0); FWV(2, 5, 4, 3, 0); FWV(2, 5, 3, 2, 0); FWV(2, 5, 3, 1, 0); FWV(2, 5, 3, 6, 0); FWV(2, 5, 3, 12, 0);
            FWV(1, 5, 4, 8, 1); FWV(1, 5, 4, 4, 1); FWV(2, 5, 3, 2, 1); FWV(2, 5, 3, 4, 1);
            CK(hipFree(P.dA)); CK(hipFree(P.dB)); CK(hipFree(P.dC)); CK(hipFree(P.ws)); CK(hipFree(P.cnt)); CK(hipFree(P.err));
        }
        {
            Prob P; make_prob(P, FWD, 512, 784, 400);
            printf("FWD 512x400 -> 784\n");
            baselines(P, h);
            FWV(1, 7, 4, 4, 0); FWV(1, 7, 4, 2, 0); FWV(1, 7, 4, 4, 1); FWV(1, 7, 3, 3, 0);
        }
    }
    return 0;
}
