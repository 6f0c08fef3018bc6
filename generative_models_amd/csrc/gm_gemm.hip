// gm_gemm.hip -- fp32 MFMA GEMMs for the 784<->400<->20 MLP layers (forward, dX, dW) on gfx950.
//
// One kernel body, three operand layouts:
//   fwd  (NT): Y[M,N]  = act(X[M,K] * W[N,K]^T + b)         ns_gan.py:44-45,58-59
//   dx   (NN): dX[M,K] = dA[M,N] * W[N,K] (* act'(below))   autograd of the above, ns_gan.py:138,155
//   dw   (TN): dW[N,K] = dA[M,N]^T * X[M,K], db = colsum(dA)
//
// Design for this problem (tiny, L2-resident, latency-bound -- SURVEY.md section 7 "hard parts"):
//   * 32x32 output tile per 256-thread workgroup so that even B=256 launches >= 200 workgroups;
//   * the reduction dimension of every K-step is SPLIT ACROSS THE FOUR WAVES of the workgroup
//     (wave w owns k in [w*BK/4,(w+1)*BK/4)): each wave keeps one 32x32 accumulator and issues
//     v_mfma_f32_32x32x2_f32 back to back (64-cycle issue == dependent latency, so one accumulator
//     runs the matrix pipe at full rate), then the four partial tiles are summed through LDS.
//     This cuts the dependent MFMA chain for K=784 from 392 to 98 instructions per wave;
//   * operands are staged global -> registers -> LDS in a [k][x] layout with row stride 36 floats:
//     the MFMA fragment reads (32 consecutive floats of one k row per half-wave) and the
//     transposing ds_write_b32 stores are both bank-conflict free; global loads of tile t+1 are in
//     flight while tile t is multiplied (register prefetch + double-buffered LDS, one barrier per
//     K-step);
//   * exact fp32: MFMA f32 is a k-ordered fmaf chain, no reduced-precision path exists on gfx950.
//   * db falls out of the dW GEMM for free: X is given a virtual ones-column at index K.
#include "gm_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int TM = 32, TN = 32;
constexpr int LD = 36;

enum { MODE_FWD = 0, MODE_DX = 1, MODE_DW = 2 };

struct GemmP {
    const float* A;
    const float* B;
    float* C;
    int M, N, K;              // GEMM dims: C[M,N] = sum_k A(m,k) B(k,n)
    int64_t lda, ldb, ldc;
    const float* bias;        // fwd
    const float* aux;         // dx: output of the layer below [M,N]
    int64_t ldaux;
    float* db;                // dw: bias gradient (virtual ones column n == N_real)
    int n_real;               // dw: number of real columns of B (N = n_real + 1 when db)
    int epi;
    int accumulate;
    gm_slot a_slot, b_slot;
};

// Operand element (x, k) lives at P[x*ld + k]  (k contiguous).
template <int BK, bool VEC>
struct LoaderKC {
    float4 r[BK / 32];
    __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int x0, int X,
                                         int k0, int K, int t) {
        const int x = x0 + (t & 31);
#pragma unroll
        for (int i = 0; i < BK / 32; ++i) {
            const int k = k0 + 4 * ((t >> 5) + 8 * i);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (x < X) {
                const float* q = P + (int64_t)x * ld + k;
                if (VEC && k + 3 < K) {
                    v = *reinterpret_cast<const float4*>(q);
                } else {
                    if (k + 0 < K) v.x = q[0];
                    if (k + 1 < K) v.y = q[1];
                    if (k + 2 < K) v.z = q[2];
                    if (k + 3 < K) v.w = q[3];
                }
            }
            r[i] = v;
        }
    }
    __device__ __forceinline__ void store(float* S, int t) const {
#pragma unroll
        for (int i = 0; i < BK / 32; ++i) {
            const int kr = 4 * ((t >> 5) + 8 * i);
            float* s = S + kr * LD + (t & 31);
            s[0 * LD] = r[i].x;
            s[1 * LD] = r[i].y;
            s[2 * LD] = r[i].z;
            s[3 * LD] = r[i].w;
        }
    }
};

// Operand element (x, k) lives at P[k*ld + x]  (x contiguous).  ones_col: virtual column of 1s.
template <int BK, bool VEC>
struct LoaderXC {
    float4 r[BK / 32];
    __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int x0, int X,
                                         int k0, int K, int t, int ones_col) {
        const int x = x0 + 4 * (t & 7);
#pragma unroll
        for (int i = 0; i < BK / 32; ++i) {
            const int k = k0 + (t >> 3) + 32 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < K) {
                const float* q = P + (int64_t)k * ld + x;
                if (VEC && x + 3 < X) {
                    v = *reinterpret_cast<const float4*>(q);
                } else {
                    if (x + 0 < X) v.x = q[0];
                    if (x + 1 < X) v.y = q[1];
                    if (x + 2 < X) v.z = q[2];
                    if (x + 3 < X) v.w = q[3];
                    if (ones_col >= 0) {
                        if (x + 0 == ones_col) v.x = 1.f;
                        if (x + 1 == ones_col) v.y = 1.f;
                        if (x + 2 == ones_col) v.z = 1.f;
                        if (x + 3 == ones_col) v.w = 1.f;
                    }
                }
            }
            r[i] = v;
        }
    }
    __device__ __forceinline__ void store(float* S, int t) const {
#pragma unroll
        for (int i = 0; i < BK / 32; ++i) {
            const int kr = (t >> 3) + 32 * i;
            *reinterpret_cast<float4*>(S + kr * LD + 4 * (t & 7)) = r[i];
        }
    }
};

template <int MODE, int BK, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmP p) {
    // [buf][operand][BK][LD]; the cross-wave reduction buffer (4*32*33 floats) aliases it.
    __shared__ __attribute__((aligned(16))) float smem[2 * 2 * BK * LD];
    static_assert(2 * 2 * BK * LD >= 4 * 32 * 33, "reduction buffer must fit");

    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;

    const float* A = p.A + gm_slot_offset(p.a_slot);
    const float* B = p.B + gm_slot_offset(p.b_slot);

    // A is k-contiguous for fwd/dx, m-contiguous for dw; B is k-contiguous for fwd only.
    LoaderKC<BK, VEC> a_kc;
    LoaderXC<BK, VEC> a_xc;
    LoaderKC<BK, VEC> b_kc;
    LoaderXC<BK, VEC> b_xc;
    const int b_cols = (MODE == MODE_DW) ? p.n_real : p.N;      // real columns of B
    const int ones_col = (MODE == MODE_DW && p.db) ? p.n_real : -1;

    auto load_tiles = [&](int k0) {
        if (MODE == MODE_DW) a_xc.load(A, p.lda, m0, p.M, k0, p.K, t, -1);
        else                 a_kc.load(A, p.lda, m0, p.M, k0, p.K, t);
        if (MODE == MODE_FWD) b_kc.load(B, p.ldb, n0, p.N, k0, p.K, t);
        else                  b_xc.load(B, p.ldb, n0, b_cols, k0, p.K, t, ones_col);
    };
    auto store_tiles = [&](int buf) {
        float* As = smem + buf * (2 * BK * LD);
        float* Bs = As + BK * LD;
        if (MODE == MODE_DW) a_xc.store(As, t); else a_kc.store(As, t);
        if (MODE == MODE_FWD) b_kc.store(Bs, t); else b_xc.store(Bs, t);
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    const int nt = (p.K + BK - 1) / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int it = 0; it < nt; ++it) {
        const int buf = it & 1;
        if (it + 1 < nt) load_tiles((it + 1) * BK);
        const float* As = smem + buf * (2 * BK * LD);
        const float* Bs = As + BK * LD;
        const int kw = w * (BK / 4) + (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const float a = As[(kw + 2 * kk) * LD + (lane & 31)];
            const float b = Bs[(kw + 2 * kk) * LD + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (it + 1 < nt) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // cross-wave reduction through LDS (aliases the tile buffers; all reads are done)
    float* red = smem;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        red[(w * 32 + row) * 33 + (lane & 31)] = acc[r];
    }
    __syncthreads();
    const int row = t >> 3, c4 = 4 * (t & 7);
    const int m = m0 + row;
    if (m >= p.M) return;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int o = row * 33 + c4 + j;
        v[j] = (red[o] + red[32 * 33 + o]) + (red[2 * 32 * 33 + o] + red[3 * 32 * 33 + o]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + c4 + j;
        if (n >= p.N) continue;
        float x = v[j];
        if (MODE == MODE_FWD) {
            if (p.bias) x += p.bias[n];
            if (p.epi == GM_ACT_RELU) x = fmaxf(x, 0.f);
            else if (p.epi == GM_ACT_SIGMOID) x = gm_sigmoid(x);
            p.C[(int64_t)m * p.ldc + n] = x;
        } else if (MODE == MODE_DX) {
            if (p.epi == GM_ACT_RELU) {
                x = (p.aux[(int64_t)m * p.ldaux + n] > 0.f) ? x : 0.f;
            } else if (p.epi == GM_ACT_SIGMOID) {
                const float y = p.aux[(int64_t)m * p.ldaux + n];
                x = x * (y * (1.f - y));
            }
            float* c = p.C + (int64_t)m * p.ldc + n;
            *c = p.accumulate ? (*c + x) : x;
        } else {
            float* c = (n == p.n_real) ? (p.db + m) : (p.C + (int64_t)m * p.ldc + n);
            *c = p.accumulate ? (*c + x) : x;
        }
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int MODE>
int launch(hipStream_t s, const GemmP& p, bool vec) {
    dim3 grid((p.N + TN - 1) / TN, (p.M + TM - 1) / TM);
    if (vec) hipLaunchKernelGGL((gemm_kernel<MODE, 32, true>), grid, dim3(256), 0, s, p);
    else     hipLaunchKernelGGL((gemm_kernel<MODE, 32, false>), grid, dim3(256), 0, s, p);
    GM_LAUNCH_RET();
}

inline gm_slot no_slot() { gm_slot z; z.ctr = nullptr; z.mul = 0; z.add = 0; z.ring = 0; z.stride = 0; return z; }

}  // namespace

extern "C" int gm_linear_fwd(void* stream, const float* X, int64_t ldx, gm_slot x_slot,
                             const float* W, const float* bias, float* Y, int64_t ldy, int M,
                             int K, int N, int act) {
    GM_CHECK_ARG(X && W && Y && M > 0 && K > 0 && N > 0 && ldx >= K && ldy >= N);
    GM_CHECK_ARG(act >= GM_ACT_ID && act <= GM_ACT_SIGMOID);
    GemmP p{};
    p.A = X; p.B = W; p.C = Y; p.M = M; p.N = N; p.K = K;
    p.lda = ldx; p.ldb = K; p.ldc = ldy; p.bias = bias; p.epi = act;
    p.a_slot = x_slot; p.b_slot = no_slot();
    const bool vec = aligned16(X) && aligned16(W) && (ldx % 4 == 0) && (K % 4 == 0) &&
                     (x_slot.stride % 4 == 0);
    return launch<MODE_FWD>((hipStream_t)stream, p, vec);
}

extern "C" int gm_linear_bwd_dx(void* stream, const float* dA, int64_t lda, const float* W,
                                float* dX, int64_t ldx, const float* below, int64_t ld_below,
                                int M, int K, int N, int epi) {
    GM_CHECK_ARG(dA && W && dX && M > 0 && K > 0 && N > 0 && lda >= N && ldx >= K);
    GM_CHECK_ARG(epi == GM_ACT_ID || (below && ld_below >= K));
    GemmP p{};
    // C[M, K_layer] = sum_{n} dA[m,n] * W[n,k]  => GEMM dims (M, N=K_layer, K=N_layer)
    p.A = dA; p.B = W; p.C = dX; p.M = M; p.N = K; p.K = N;
    p.lda = lda; p.ldb = K; p.ldc = ldx; p.aux = below; p.ldaux = ld_below; p.epi = epi;
    p.a_slot = no_slot(); p.b_slot = no_slot();
    const bool vec = aligned16(dA) && aligned16(W) && (lda % 4 == 0) && (K % 4 == 0);
    return launch<MODE_DX>((hipStream_t)stream, p, vec);
}

extern "C" int gm_linear_bwd_dw(void* stream, const float* dA, int64_t lda, const float* X,
                                int64_t ldx, gm_slot x_slot, float* dW, float* db, int M, int K,
                                int N, int accumulate) {
    GM_CHECK_ARG(dA && X && dW && M > 0 && K > 0 && N > 0 && lda >= N && ldx >= K);
    GemmP p{};
    // C[N_layer, K_layer(+1)] = sum_{m} dA[m,n] * X[m,k]  => GEMM dims (M=N_layer, N=K_layer(+1), K=batch)
    p.A = dA; p.B = X; p.C = dW; p.M = N; p.N = K + (db ? 1 : 0); p.K = M;
    p.lda = lda; p.ldb = ldx; p.ldc = K; p.db = db; p.n_real = K; p.accumulate = accumulate;
    p.a_slot = no_slot(); p.b_slot = x_slot;
    const bool vec = aligned16(dA) && aligned16(X) && (lda % 4 == 0) && (ldx % 4 == 0) &&
                     (x_slot.stride % 4 == 0);
    return launch<MODE_DW>((hipStream_t)stream, p, vec);
}
