// slab_probe.hip -- stand-alone check + timing of the slab GEMM family (csrc/gm_slab.h) against an fp64 host
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

#include "../generative_models_amd/csrc/gm_slab.h"
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

template <int TMW, int TNW, int NBUF, int KU>
__global__ __launch_bounds__(256) void k_dw(CoreP p, float* C, int ldc) {
    using Cf = DwCfg<TMW, TNW, NBUF, KU>;
    __shared__ __attribute__((aligned(16))) float lds[Cf::LDS_FLOATS];
    int tile, s;
    if (!map_block(p, tile, s)) return;
    StoreEpi epi{C, ldc, p.M, p.N, (tile / p.tn) * Cf::BM, (tile % p.tn) * Cf::BN};
    dw_tile<TMW, TNW, NBUF, KU>(p, lds, tile, s, epi, NoAXf());
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

static hipStream_t g_stream;
static int g_reps = 20;

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
    p.sp.S = S; p.sp.ws = P.ws; p.sp.cnt = P.cnt; p.sp.err = P.err;
    const int T = p.tm * p.tn;
    int grid = T * S;
    if (xmap) { const int g = 8 / S; grid = 8 * ((T + g - 1) / g); }
    if (T * S > 256) { printf("  %-34s skipped: %d workgroups\n", name, T * S); return; }
    CK(hipMemsetAsync(P.dC, 0xff, (size_t)P.M * P.N * 4, g_stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, g_stream, p, P.dC, P.N);
    CK(hipStreamSynchronize(g_stream));
    const double rel = check(P, name);
    const double us = time_graph([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, g_stream, p, P.dC, P.N); });
    const double rel2 = check(P, name);
    printf("  %-34s T=%3d S=%d wg=%3d  %7.2f us  %6.1f TF  err %.1e/%.1e\n", name, T, S, grid, us,
           2.0 * P.M * P.N * P.K / us * 1e-6, rel, rel2);
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
    rocblas_handle h; rocblas_create_handle(&h); rocblas_set_stream(h, g_stream);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  CUs %d  clock %d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);

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
