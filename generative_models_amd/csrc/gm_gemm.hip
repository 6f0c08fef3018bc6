// gm_gemm.hip -- fp32 MFMA GEMMs for the 784<->400<->20 MLP layers (forward, dX, dW) on gfx950.
//
// One kernel body, three operand layouts:
//   fwd  (NT): Y[M,N]  = act(X[M,K] * W[N,K]^T + b)         ns_gan.py:44-45,58-59
//   dx   (NN): dX[M,K] = dA[M,N] * W[N,K] (* act'(below))   autograd of the above, ns_gan.py:138,155
//   dw   (TN): dW[N,K] = dA[M,N]^T * X[M,K], db = colsum(dA)
//
// Design for this problem (B=256: every GEMM is ~0.1-0.3 GFLOP, L2/MALL-resident, and the step is
// a chain of ~11 dependent GEMMs -- latency, not bandwidth or FLOPs, is the enemy):
//   * 32x32 output tile per workgroup so that even B=256 launches 100-325 workgroups;
//   * the reduction dimension is SPLIT ACROSS THE WAVES of the workgroup in round-robin 8-deep
//     chunks: 16 waves (1024 threads, one workgroup per CU) when the grid has <= 256 tiles, 8 waves
//     (two workgroups per CU) when it has more, so every launch runs in a single round.  No two
//     waves of a workgroup touch the same k => there is NO operand reuse inside the workgroup, so
//     operands are NOT staged through LDS: each wave loads its chunks straight into
//     v_mfma_f32_32x32x2_f32 fragment registers (the "load straight to VGPRs, deep unroll, late
//     vmcnt" regime of the CDNA guide).  The dependent chain for K=784 is 28 MFMAs per wave
//     instead of 392, with no barrier inside it;
//   * loads are branch-free (clamped addresses, zeroing selects deferred to the consume stage) and
//     issued G at a time back to back (G in {2,4,7} per launch = the wave's whole k-range for the
//     shapes of this model), so hipcc emits counted s_waitcnt vmcnt(n) between the MFMA groups
//     instead of a full drain behind every guarded load;
//   * fragment trick: lane (row r = lane&31, half h = lane>>5) loads 4 consecutive k
//     (k = 8c+4h+j) of its row with ONE 16-byte load; MFMA j consumes element j, i.e. the k-order
//     inside a chunk is permuted identically for A and B -- legal because a sum does not care.
//     x-contiguous operands (dX's W, dW's dA and X) use one 16-byte load of 4 consecutive x at one
//     k plus a 4x4 transpose inside each lane quad (two DPP quad_perm steps, no LDS);
//   * the partial tiles are combined through 64/32 KB of LDS (conflict-free row writes/reads), then
//     bias / activation / activation-gradient / accumulate / Adam epilogues are applied and rows
//     are stored as coalesced 128-byte segments;
//   * exact fp32 (MFMA f32 == fmaf chain); deterministic (no atomics);
//   * db falls out of the dW GEMM for free: X gets a virtual ones-column at index K;
//   * optional optimizer-in-epilogue: the thread that produces a gradient element also applies
//     Adam to the parameter (single-GPU fast path), removing the separate Adam launches.
#include "gm_common.h"
#include "gm_head.h"
#include "gm_gather.h"
#include "gm_slab.h"

#include <cstdlib>
#include <type_traits>

// Compile-time experiment switches of the 16x16x4 body (tools/build_variant.sh builds a library per
// setting; a RUNTIME flag inside the reduction loop distorts what it measures: with three such flags
// in the loop the default path's dW at 2048 rows went from 32 to 50 us -- profiles/r03_experiments.md):
//   GM_XDIRECT          x-contiguous operands as direct dword fragments: 1 everywhere, 2 in the dX GEMM only
//                       (its W operand), 0: 16-byte loads + quad transposes
//   GM_EXP_BATCH_LOADS  1: all of a wave's chunk loads issued back to back (measured slower)
//   GM_EXP_ABLATE       timing only -- 1: no MFMA, 2: no operand loads, 3: no cross-wave reduction / epilogue
#ifndef GM_XDIRECT
#define GM_XDIRECT 0
#endif
//   GM_DW_IL            1 (default): weight gradients take their x-contiguous operands as INTERLEAVED fragments
//                       (gemm16_dw_il) -- no quad transposes; 0: round 2's 16-byte loads + transposes
#ifndef GM_DW_IL
#define GM_DW_IL 1
#endif
//   GM_DW_IL_PREFETCH   1: the interleaved form loads chunk q + 1 before it consumes chunk q (measured: no change)
#ifndef GM_DW_IL_PREFETCH
#define GM_DW_IL_PREFETCH 0
#endif
#ifndef GM_EXP_BATCH_LOADS
#define GM_EXP_BATCH_LOADS 0
#endif
#ifndef GM_EXP_ABLATE
#define GM_EXP_ABLATE 0
#endif
//   GM_MFMA_INTERLEAVE  1: the four k-steps of a chunk are the OUTER loop of the MFMA block, so consecutive MFMAs of a
//                       wave go to different accumulators (no dependent back-to-back pairs); 0: accumulator by
//                       accumulator, four dependent MFMAs in a row (what hipcc emits in source order)
//   GM_EXP_PREFETCH     n > 0: weight-gradient launches whose waves own >= n chunks load chunk q + 1 before they
//                       consume chunk q (two register sets, counted waits)
#ifndef GM_EXP_PREFETCH
#define GM_EXP_PREFETCH 0
#endif
#ifndef GM_MFMA_INTERLEAVE
#define GM_MFMA_INTERLEAVE 0
#endif
//   GM_SPREAD_TAIL      1: when the reduction leaves 1..4 chunks over after every wave had its equal share (K = 784:
//                       49 chunks on 16 waves), those chunks are spread BY k-STEP over 4 waves each, loaded in the
//                       waves' last regular round -- no wave pays an extra trip to memory for a quarter of the others'
//                       work; 0: waves 0..r-1 take one more whole chunk
#ifndef GM_SPREAD_TAIL
#define GM_SPREAD_TAIL 0
#endif
//   GM_FAST_INTERIOR    1: chunks that lie wholly below K of tiles that lie wholly inside the operands skip the bounds
//                       selects of the fragment fix-up (they are most of the work: layer widths are multiples of 16)
#ifndef GM_FAST_INTERIOR
#define GM_FAST_INTERIOR 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int TM = 32, TN = 32;
constexpr int MAXW = 16;      // waves per workgroup: 16 (one workgroup per CU) or 8 (two per CU)
// G = chunks (of 8 k) loaded back to back per wave and batch: template parameter, chosen per launch
// as the smallest of {2, 4, 7} covering the wave's k-range (no redundant loads for short K).

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
    int ones_from;            // dw: the ones column is 1 for reduction rows >= ones_from, 0 before (a
                              //     stacked reduction whose first rows must not reach the bias gradient)
    int n_real;               // dw: number of real columns of B (N = n_real + 1 when db)
    int epi;
    int accumulate;
    gm_slot a_slot, b_slot;
    // XCD-aware tile mapping (0 = plain 2-D grid).  The 8 XCDs form an xr x xc grid; XCD (i,j)
    // owns m-tiles [i*tm/xr, (i+1)*tm/xr) x n-tiles [j*tn/xc, (j+1)*tn/xc), so the operand rows a
    // private L2 has to pull over the fabric shrink from "all of A and B" to 1/xr of A + 1/xc of B.
    int xr, xc, tm, tn;
    int cpw;                  // >0: wave w owns the CONTIGUOUS chunks [w*cpw, (w+1)*cpw)
    int lds_tm, lds_mpx;      // LDS macro-tile kernel: m-tiles in total / per XCD (n-tiles: tn)
    int x16;                  // gemm16_kernel: XCD-aware tile map on a 1-D grid (uses xr, xc, tm, tn)
    int il;                   // dw: interleaved fragments (gemm16_dw_il) instead of 16-byte loads + quad transposes
    int vec_epi;              // every array the epilogue touches is 16-byte aligned with rows of whole float4s (host check)
    // fwd: second output for rows m < ip_rows (WGAN-GP's x_hat written by the generator's last
    // layer): ip_out[m][n] = eps[m] * ip_x[m][n] + (1 - eps[m]) * C[m][n]      (w_gp_gan.py:197-201)
    const float* ip_eps; gm_slot ip_slot;
    const float* ip_x; int64_t ip_ldx;
    float* ip_out; int64_t ip_ldo;
    int ip_rows;
    gm_adam_epi adam;         // dw: apply Adam to the parameter right where its gradient is produced
    // fwd, folded critic head (gm_head.h): partial dots of the N = 1 layer over this launch's output,
    // one per 32-column tile and row, + a snapshot of the head parameters used
    const float* hd_w2; const float* hd_b2; float* hd_part; int64_t hd_ldp; float* hd_snap;
    // fwd, reconstruction loss in the epilogue (vae.py:203, ae.py:152: sum (x - x_hat)^2 over a sigmoid
    // output layer): sq_dA[m][n] = d loss / d (pre-sigmoid output), sq_part[m * sq_ldp + n0/32] = the
    // row's squared error inside this 32-column tile (summed later in a fixed order, gm_sum_finalize*)
    const float* sq_x; int64_t sq_ldx; float* sq_dA; int64_t sq_lda; float* sq_part; int64_t sq_ldp;
    // dx, reparameterisation backward in the epilogue (vae.py:100-106,210-212): the output IS dz; the
    // epilogue writes d loss / d [mu | log_var] = [dz + mu | dz*eps*exp(lv/2)/2 + (exp(lv) - 1)/2]
    const float* rp_ml; int64_t rp_ldml; const float* rp_eps; gm_slot rp_slot; float* rp_dml; int64_t rp_ldd;
    int rp_Z;
    // dw / dx, folded head: A is the hidden layer h; dH = dS[row] * fold_w2[col] * [h > 0] is formed
    // in registers (dS: the workgroup's LDS copy, filled by the kernel's prologue)
    const float* fold_w2;
    const float* add;         // dx: v += add_scale * add[m,n] before the activation gradient
    int64_t ldadd;
    float add_scale;
};

// Operand loads are BRANCH-FREE: out-of-range rows / k are clamped to a valid address and the
// value is zeroed by a select afterwards.  (With `if (in range) load` hipcc branches around every
// load and puts an s_waitcnt vmcnt(0) behind each one -- 14 serialized L2 round trips per wave,
// measured as a 3.5 us load phase that did not overlap the MFMA chain.)

// The load (raw_*) and the zeroing (fix_*) are separate functions: the selects run in the consume
// stage, right before the MFMAs, so that nothing forces a wait while the loads are in flight.

// k-contiguous operand: element (x, k) at P[x*ld + k]; the 4 values k = kb..kb+3.
template <bool VEC>
__device__ __forceinline__ float4 raw_kc(const float* __restrict__ P, int64_t ld, int x, int X,
                                         int kb, int K) {
    const float* row = P + (int64_t)min(x, X - 1) * ld;
    if (VEC)                        // K % 4 == 0 and 16-byte aligned rows: kb < K <=> kb+3 < K
        return *reinterpret_cast<const float4*>(row + min(kb, K - 4));
    return make_float4(row[min(kb + 0, K - 1)], row[min(kb + 1, K - 1)], row[min(kb + 2, K - 1)],
                       row[min(kb + 3, K - 1)]);
}
__device__ __forceinline__ float4 fix_kc(float4 v, int x, int X, int kb, int K) {
    const bool okx = x < X;
    v.x = (okx && kb + 0 < K) ? v.x : 0.f;
    v.y = (okx && kb + 1 < K) ? v.y : 0.f;
    v.z = (okx && kb + 2 < K) ? v.z : 0.f;
    v.w = (okx && kb + 3 < K) ? v.w : 0.f;
    return v;
}

// x-contiguous operand: element (x, k) at P[k*ld + x]; 4 coalesced dword loads (one per k).
// ones_col: virtual column of 1s (bias gradient through the dW GEMM).
__device__ __forceinline__ float4 raw_xc(const float* __restrict__ P, int64_t ld, int x, int X,
                                         int kb, int K) {
    const float* col = P + min(x, X - 1);
    return make_float4(col[(int64_t)min(kb + 0, K - 1) * ld], col[(int64_t)min(kb + 1, K - 1) * ld],
                       col[(int64_t)min(kb + 2, K - 1) * ld], col[(int64_t)min(kb + 3, K - 1) * ld]);
}
// Vector form for x-contiguous operands (ld % 4 == 0, X % 4 == 0, 16-byte aligned): ONE 16-byte load
// per lane instead of four dword loads.  Lane l = e + 4q + 32m fetches x = x0+4q..4q+3 of row
// k = 8c+4m+e (8 k-rows x 128 contiguous bytes per instruction); a 4x4 transpose inside each lane
// quad (two DPP quad_perm exchange steps, no LDS) then gives lane (r = 4q+e, m) its four k values
// -- exactly the MFMA fragment layout.
__device__ __forceinline__ float4 raw_xc4(const float* __restrict__ P, int64_t ld, int x0, int X,
                                          int c, int K, int lane) {
    const int e = lane & 3, q = (lane >> 2) & 7, m = lane >> 5;
    const int k = min(8 * c + 4 * m + e, K - 1);
    const int x = min(x0 + 4 * q, X - 4);
    return *reinterpret_cast<const float4*>(P + (int64_t)k * ld + x);
}
__device__ __forceinline__ float dpp_xor2(float v) {   // lane ^ 2 within the quad
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor1(float v) {   // lane ^ 1 within the quad
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
// new[lane e][reg j] = old[lane j][reg e] over the 4 lanes of a quad
__device__ __forceinline__ float4 quad_transpose(float4 v, int lane) {
    const bool b1 = lane & 2, b0 = lane & 1;
    // step 1: exchange 2x2 blocks with lane^2
    const float s0 = dpp_xor2(b1 ? v.x : v.z), s1 = dpp_xor2(b1 ? v.y : v.w);
    if (b1) { v.x = s0; v.y = s1; } else { v.z = s0; v.w = s1; }
    // step 2: exchange inside the 2x2 blocks with lane^1
    const float t0 = dpp_xor1(b0 ? v.x : v.y), t1 = dpp_xor1(b0 ? v.z : v.w);
    if (b0) { v.x = t0; v.z = t1; } else { v.y = t0; v.w = t1; }
    return v;
}

__device__ __forceinline__ float4 fix_xc(float4 v, int x, int X, int kb, int K, int ones_col,
                                         int ones_from = 0) {
    const bool okx = x < X, oc = (x == ones_col);
    v.x = (kb + 0 < K) ? (okx ? v.x : ((oc && kb + 0 >= ones_from) ? 1.f : 0.f)) : 0.f;
    v.y = (kb + 1 < K) ? (okx ? v.y : ((oc && kb + 1 >= ones_from) ? 1.f : 0.f)) : 0.f;
    v.z = (kb + 2 < K) ? (okx ? v.z : ((oc && kb + 2 >= ones_from) ? 1.f : 0.f)) : 0.f;
    v.w = (kb + 3 < K) ? (okx ? v.w : ((oc && kb + 3 >= ones_from) ? 1.f : 0.f)) : 0.f;
    return v;
}

// Epilogue of one finished output element C(m, n) = v (m < M, n < N checked by the caller).
// forward epilogue value: bias + activation
__device__ __forceinline__ float fwd_value(const GemmP& p, float v, int n) {
    if (p.bias) v += p.bias[n];
    if (p.epi == GM_ACT_RELU) v = fmaxf(v, 0.f);
    else if (p.epi == GM_ACT_SIGMOID) v = gm_sigmoid(v);
    return v;
}

template <int MODE>
__device__ __forceinline__ void store_element(const GemmP& p, float v, int m, int n) {
    if (MODE == MODE_FWD) {
        v = fwd_value(p, v, n);
        p.C[(int64_t)m * p.ldc + n] = v;
        if (p.ip_out && m < p.ip_rows) {
            // two roundings and an add, never contracted: torch's eps * x + (1 - eps) * g
            const float ev = (p.ip_eps + gm_slot_offset(p.ip_slot))[m];
            p.ip_out[(int64_t)m * p.ip_ldo + n] = gm_interp_unfused(ev, p.ip_x[(int64_t)m * p.ip_ldx + n], v);
        }
    } else if (MODE == MODE_DX) {
        if (p.add) v += p.add_scale * p.add[(int64_t)m * p.ldadd + n];
        if (p.epi == GM_ACT_RELU) {
            v = (p.aux[(int64_t)m * p.ldaux + n] > 0.f) ? v : 0.f;
        } else if (p.epi == GM_ACT_SIGMOID) {
            const float y = p.aux[(int64_t)m * p.ldaux + n];
            v = v * (y * (1.f - y));
        }
        float* cp = p.C + (int64_t)m * p.ldc + n;
        *cp = p.accumulate ? (*cp + v) : v;
        if (p.rp_dml) {                                       // kernel-argument uniform
            // same expressions, same order as gm_vae_reparam_bwd (bit-identical to the separate launch)
            const int Z = p.rp_Z;
            const float mu = p.rp_ml[(int64_t)m * p.rp_ldml + n], lv = p.rp_ml[(int64_t)m * p.rp_ldml + Z + n];
            const float e = (p.rp_eps + gm_slot_offset(p.rp_slot))[(int64_t)m * Z + n];
            p.rp_dml[(int64_t)m * p.rp_ldd + n] = v + 0.5f * (2.f * mu);
            p.rp_dml[(int64_t)m * p.rp_ldd + Z + n] = ((v * e) * expf(lv / 2.f)) / 2.f + 0.5f * (expf(lv) - 1.f);
        }
    } else {
        const bool is_b = (n == p.n_real);
        float* cp = is_b ? (p.db + m) : (p.C + (int64_t)m * p.ldc + n);
        if (p.accumulate) v += *cp;
        *cp = v;
        if (p.adam.enabled) {
            // optimizer fused into the gradient epilogue: every gradient element is produced by
            // exactly one thread, so Adam can run here and the separate launch disappears
            const int64_t o = is_b ? (int64_t)m : ((int64_t)m * p.ldc + n);
            float* pp = (is_b ? p.adam.pb : p.adam.pW) + o;
            float* mm = (is_b ? p.adam.mb : p.adam.mW) + o;
            float* vv = (is_b ? p.adam.vb : p.adam.vW) + o;
            // (prefetching p/m/v or the schedule scalars at kernel start was measured SLOWER: vmcnt
            // and lgkmcnt retire in order, so the early requests hold back the operand loads --
            // profiles/r01_experiments.md)
            // (loading p / m / v EARLY was measured slower twice: at kernel start, ahead of the operand loads
            // (round 1, +1.4 us on the paired launch), and right behind the last chunk's operand loads
            // (round 3: dW+head 13.8 -> 15.2 us, pair 10.9 -> 12.5 us, profiles/r03_experiments.md))
            const int64_t si = gm_slot_index(p.adam.sched_slot);
            const float step_size = p.adam.sched[2 * si], bc2_sqrt = p.adam.sched[2 * si + 1];
            float P = *pp, M = *mm, V = *vv;
            adam_update(P, v, M, V, step_size, bc2_sqrt, p.adam.omb1, p.adam.b2, p.adam.omb2,
                        p.adam.eps, p.adam.wd, p.adam.clamp);
            *pp = P; *mm = M; *vv = V;
        }
    }
}

// Four consecutive outputs C(m, n .. n+3), n % 4 == 0, all inside the real columns (p.vec_epi: every array involved is
// 16-byte aligned with a leading dimension of whole float4s).  Same arithmetic per element as store_element, in the
// same order -- bit-identical results; what changes is the number of vector-memory instructions: one CU retires about
// one wave-wide load or store per 40 cycles whatever its width (profiles/r04_experiments.md), so the element-wise
// epilogue of a 32 x 48 weight-gradient tile with Adam (3 loads + 4 stores per element on 16 waves) was ~170
// instructions = 3 us of a 9 us launch; as float4s on 4 waves it is 42.
__device__ __forceinline__ float4 ld4(const float* q) { return *reinterpret_cast<const float4*>(q); }
__device__ __forceinline__ void st4(float* q, float4 v) { *reinterpret_cast<float4*>(q) = v; }

template <int MODE>
__device__ __forceinline__ void store4(const GemmP& p, float4 v, int m, int n) {
    if (MODE == MODE_FWD) {
        if (p.bias) { const float4 b = ld4(p.bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        if (p.epi == GM_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        else if (p.epi == GM_ACT_SIGMOID) { v.x = gm_sigmoid(v.x); v.y = gm_sigmoid(v.y); v.z = gm_sigmoid(v.z); v.w = gm_sigmoid(v.w); }
        st4(p.C + (int64_t)m * p.ldc + n, v);
        if (p.ip_out && m < p.ip_rows) {
            const float ev = (p.ip_eps + gm_slot_offset(p.ip_slot))[m];
            const float4 x = ld4(p.ip_x + (int64_t)m * p.ip_ldx + n);
            st4(p.ip_out + (int64_t)m * p.ip_ldo + n,
                make_float4(gm_interp_unfused(ev, x.x, v.x), gm_interp_unfused(ev, x.y, v.y),
                            gm_interp_unfused(ev, x.z, v.z), gm_interp_unfused(ev, x.w, v.w)));
        }
    } else if (MODE == MODE_DX) {
        if (p.add) {
            const float4 a = ld4(p.add + (int64_t)m * p.ldadd + n);
            v.x += p.add_scale * a.x; v.y += p.add_scale * a.y; v.z += p.add_scale * a.z; v.w += p.add_scale * a.w;
        }
        if (p.epi == GM_ACT_RELU) {
            const float4 y = ld4(p.aux + (int64_t)m * p.ldaux + n);
            v.x = (y.x > 0.f) ? v.x : 0.f; v.y = (y.y > 0.f) ? v.y : 0.f; v.z = (y.z > 0.f) ? v.z : 0.f; v.w = (y.w > 0.f) ? v.w : 0.f;
        } else if (p.epi == GM_ACT_SIGMOID) {
            const float4 y = ld4(p.aux + (int64_t)m * p.ldaux + n);
            v.x = v.x * (y.x * (1.f - y.x)); v.y = v.y * (y.y * (1.f - y.y));
            v.z = v.z * (y.z * (1.f - y.z)); v.w = v.w * (y.w * (1.f - y.w));
        }
        float* cp = p.C + (int64_t)m * p.ldc + n;
        if (p.accumulate) { const float4 c = ld4(cp); v.x = c.x + v.x; v.y = c.y + v.y; v.z = c.z + v.z; v.w = c.w + v.w; }
        st4(cp, v);
    } else {
        const int64_t o = (int64_t)m * p.ldc + n;
        if (p.accumulate) { const float4 c = ld4(p.C + o); v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w; }
        st4(p.C + o, v);
        if (p.adam.enabled) {
            const int64_t si = gm_slot_index(p.adam.sched_slot);
            const float step_size = p.adam.sched[2 * si], bc2_sqrt = p.adam.sched[2 * si + 1];
            float4 P = ld4(p.adam.pW + o), M = ld4(p.adam.mW + o), V = ld4(p.adam.vW + o);
            adam_update(P.x, v.x, M.x, V.x, step_size, bc2_sqrt, p.adam.omb1, p.adam.b2, p.adam.omb2, p.adam.eps, p.adam.wd, p.adam.clamp);
            adam_update(P.y, v.y, M.y, V.y, step_size, bc2_sqrt, p.adam.omb1, p.adam.b2, p.adam.omb2, p.adam.eps, p.adam.wd, p.adam.clamp);
            adam_update(P.z, v.z, M.z, V.z, step_size, bc2_sqrt, p.adam.omb1, p.adam.b2, p.adam.omb2, p.adam.eps, p.adam.wd, p.adam.clamp);
            adam_update(P.w, v.w, M.w, V.w, step_size, bc2_sqrt, p.adam.omb1, p.adam.b2, p.adam.omb2, p.adam.eps, p.adam.wd, p.adam.clamp);
            st4(p.adam.pW + o, P); st4(p.adam.mW + o, M); st4(p.adam.vW + o, V);
        }
    }
}

// the float4 group (m, n .. n+3) as store4 where it is whole and real, element by element at the edges (the dW ones
// column, a ragged last group)
template <int MODE>
__device__ __forceinline__ void store_group(const GemmP& p, float4 v, int m, int n, int ncap) {
    const int nreal = (MODE == MODE_DW) ? p.n_real : p.N;
    if (n + 3 < nreal && n + 3 < ncap) { store4<MODE>(p, v, m, n); return; }
    if (n < p.N && n < ncap) store_element<MODE>(p, v.x, m, n);
    if (n + 1 < p.N && n + 1 < ncap) store_element<MODE>(p, v.y, m, n + 1);
    if (n + 2 < p.N && n + 2 < ncap) store_element<MODE>(p, v.z, m, n + 2);
    if (n + 3 < p.N && n + 3 < ncap) store_element<MODE>(p, v.w, m, n + 3);
}

// Sum the per-wave partial tiles (red[w][32][32]) and apply the epilogue of the mode.
// ncap: columns >= ncap are not this block's to store (the 16-column last block of a 48-wide tile);
// rcap: rows of the block that belong to the tile (16 for the last block of a 48-row tile)
template <int MODE, int WAVES, int ROWS = 32>
__device__ __forceinline__ void reduce_and_store(const GemmP& p, const float* red, int t, int m0,
                                                 int n0, int ncap = 0x7fffffff, int rcap = 32) {
    if (MODE == MODE_FWD && p.hd_part) {                     // kernel-argument uniform
        // Folded critic head: every row of this 32-column block also leaves its partial dot with w2.
        // The 32 lanes that hold a row (one half of a wave) sum their products in a fixed butterfly;
        // no lane leaves the loop early, out-of-range elements contribute 0.
#pragma unroll
        for (int e = 0; e < 1024 / (WAVES * 64); ++e) {
            const int row = (t >> 5) + e * (WAVES * 2), col = t & 31;
            const bool live = !(ROWS < 32 && row >= ROWS) && row < rcap;
            float v = 0.f;
            if (live) {
#pragma unroll
                for (int ww = 0; ww < WAVES; ++ww) v += red[(ww * 32 + row) * 32 + col];
            }
            const int m = m0 + row, n = n0 + col;
            const bool ok = live && m < p.M && n < p.N && n < ncap;
            float pv = 0.f;
            if (ok) {
                const float w = p.hd_w2[n];
                const float y = fwd_value(p, v, n);
                p.C[(int64_t)m * p.ldc + n] = y;
                pv = y * w;
                if (m == 0 && p.hd_snap) {                   // the head parameters this forward used
                    p.hd_snap[n] = w;
                    if (n == 0) p.hd_snap[p.N] = p.hd_b2[0];
                }
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) pv += __shfl_xor(pv, o, 64);
            if (col == 0 && live && m < p.M) p.hd_part[(int64_t)m * p.hd_ldp + (n0 >> 5)] = pv;
        }
        return;
    }
    if (MODE == MODE_FWD && p.sq_part) {                     // kernel-argument uniform
        // Reconstruction loss where x_hat is produced: x and x_hat meet in this thread.  dA exactly as
        // gm_sqerr_sigmoid_bwd writes it; the row's squared error inside this tile by the same 32-lane
        // butterfly as the folded head's partial dots.
#pragma unroll
        for (int e = 0; e < 1024 / (WAVES * 64); ++e) {
            const int row = (t >> 5) + e * (WAVES * 2), col = t & 31;
            const bool live = !(ROWS < 32 && row >= ROWS) && row < rcap;
            float v = 0.f;
            if (live) {
#pragma unroll
                for (int ww = 0; ww < WAVES; ++ww) v += red[(ww * 32 + row) * 32 + col];
            }
            const int m = m0 + row, n = n0 + col;
            const bool ok = live && m < p.M && n < p.N && n < ncap;
            float pv = 0.f;
            if (ok) {
                const float r = fwd_value(p, v, n);
                p.C[(int64_t)m * p.ldc + n] = r;
                const float d = p.sq_x[(int64_t)m * p.sq_ldx + n] - r;
                pv = d * d;
                const float go = -(2.f * d);                  // PowBackward * SubBackward
                p.sq_dA[(int64_t)m * p.sq_lda + n] = (go * (1.f - r)) * r;   // SigmoidBackward
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) pv += __shfl_xor(pv, o, 64);
            if (col == 0 && live && m < p.M) p.sq_part[(int64_t)m * p.sq_ldp + (n0 >> 5)] = pv;
        }
        return;
    }
    // (float4 groups on the first 256 threads were measured SLOWER here than one element per thread on all 1024:
    // NSGAN bs=256 70.1 -> 77.6 us, the 784 x 400 weight gradient with Adam 8.6 -> 9.9 us -- four times the epilogue
    // arithmetic behind one memory round trip on a quarter of the waves; profiles/r04_experiments.md.  The many-row
    // LDS kernel, whose threads each own eight elements, does gain: store_group there.)
#pragma unroll
    for (int e = 0; e < 1024 / (WAVES * 64); ++e) {
        const int row = (t >> 5) + e * (WAVES * 2), col = t & 31;
        if ((ROWS < 32 && row >= ROWS) || row >= rcap) continue;   // 16-row tiles / blocks: half the threads idle
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < WAVES; ++ww) v += red[(ww * 32 + row) * 32 + col];
        const int m = m0 + row, n = n0 + col;
        if (m >= p.M || n >= p.N || n >= ncap) continue;
        store_element<MODE>(p, v, m, n);
    }
}

// ------------------------------------------------------------------------------------------
// LDS-staged macro-tile kernel for launches with many rows (batch >= ~512: the forward and dX GEMMs
// of config 5, bs = 1024).  The split-reduction kernels below give every wave its own k-chunks and
// therefore re-read each operand row once per 32x32 tile (8 FLOP per L2 byte) from whichever of the
// 8 XCD L2s the tile landed on -- fine while latency bounds a B=256 launch, 5.8x the algorithmic
// fabric traffic and the bound at B=1024.  Here a workgroup owns a BM x BN tile:
//   * WM x WN waves tile it 32x32 each (one v_mfma_f32_32x32x2_f32 accumulator per wave), WK wave
//     groups split each BK-deep stage between them (2 waves per SIMD hide each other's LDS latency;
//     the WK partial tiles are summed in a fixed order through LDS at the end);
//   * operands are staged through LDS, double-buffered, ONE barrier per stage: global 16-byte loads
//     of stage s+1 are issued before the MFMAs of stage s and written to the other buffer after
//     them (the K-tail zeroing select is applied at that write, never behind the load);
//   * k-contiguous operands (X, dA; W in the forward) are stored [row][BK+4] and read back as ONE
//     ds_read_b128 per lane = the 4 consecutive k the fragment trick needs (row stride = 4 mod 32
//     words: conflict-free for the 16-lane b128 groups); x-contiguous operands (W in dX) are
//     stored [k][BN+4] and read as 4 conflict-free ds_read_b32;
//   * XCD-aware mapping: workgroup b runs on XCD b % 8 (observed placement, speed only), so XCD x
//     gets the contiguous m-tiles [x*mpx, (x+1)*mpx): its private L2 holds 1/8 of A plus B instead
//     of both in full.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 keep4(bool ok, float4 v) {      // componentwise: stays in registers
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// Instruction order of one pipeline stage: the stage OPENS with an MFMA (the matrix pipe restarts
// right behind the barrier) and the stage's memory instructions -- fragment reads of the next
// stage, global loads, LDS operand writes (last: they wait for data) -- are issued one per MFMA in
// the shadow of the MFMAs.  Measured before this: all memory instructions ahead of the MFMAs left
// the pipe idle 43 % of every stage (SQ_VALU_MFMA_BUSY_CYCLES 25.6k of 44.7k wave cycles).
template <int NMFMA, int NDSR, int NVMEM, int NDSW>
__device__ __forceinline__ void lds_stage_schedule() {
    constexpr int NMEM = NDSR + NVMEM + NDSW;
    int done = 0;
#pragma unroll
    for (int q = 0; q < NMFMA; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // one MFMA
        const int upto = (NMEM * (q + 1) + NMFMA - 1) / NMFMA;               // memory ops issued by now
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (done < upto) {
                if (done < NDSR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);               // DS read
                else if (done < NDSR + NVMEM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
                else __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                           // DS write
                ++done;
            }
        }
    }
}

// Workgroup tile BM x BN; every wave owns a WTM x WTN sub-tile (32 or 64 on a side = 1, 2 or 4
// v_mfma_f32_32x32x2_f32 accumulators) and WK wave groups split each BK-deep stage between them.
template <int MODE, int BM, int BN, int WTM, int WTN, int WK, int KU>
struct LdsCfg {
    static constexpr int WM = BM / WTM, WN = BN / WTN, NW = WM * WN * WK, NT = 64 * NW, BK = 8 * WK * KU;
    static constexpr int TI = WTM / 32, TJ = WTN / 32;
    static constexpr bool A_KC = (MODE != MODE_DW);        // A(m,k) = X[m][k] (fwd, dx) / dA[k][m] (dw)
    static constexpr bool B_KC = (MODE == MODE_FWD);       // B(k,n) = W[n][k] (fwd) / W[k][n] (dx) / X[k][n] (dw)
    static constexpr int LDK = BK + 4, LDXB = BN + 4, LDXA = BM + 4;
    static constexpr int A_SZ = A_KC ? BM * LDK : BK * LDXA;
    static constexpr int B_SZ = B_KC ? BN * LDK : BK * LDXB;
    static constexpr int STAGE = A_SZ + B_SZ;
    static constexpr int NBUF = 3;                          // LDS stage buffers (see the pipeline below)
    static constexpr int RED = WK * BM * BN;
    static constexpr int FLOATS = (NBUF * STAGE > RED) ? NBUF * STAGE : RED;
};

// Software pipeline of one workgroup (stage = BK reduction steps; one barrier per stage):
//   during the MFMAs of stage s (fragments already in registers) every wave has, in flight,
//     * the ds_read_b128 of stage s+1's fragments            (LDS buffer (s+1) % 3),
//     * the ds_write_b128 of stage s+2's operand tile         (LDS buffer (s+2) % 3),
//     * the global loads of stages s+3 .. s+2+PD              (PD register sets).
//   Nothing but the barrier stands between two stages' MFMAs.
// NS > 0: the reduction loop is fully unrolled for exactly NS stages (K = 784 / 400 with the shipped
// tile depths): every s_waitcnt is a counted one, the global addresses of a stage are immediate
// offsets from per-thread base pointers, and the K-tail clamp / zeroing exists only in the last
// stage.  NS == 0 keeps a runtime loop (full vmcnt drain on the back-edge, clamps everywhere).
template <int MODE, int BM, int BN, int WTM, int WTN, int WK, int KU, int PD, int NS>
__global__ __launch_bounds__(64 * (BM / WTM) * (BN / WTN) * WK) void gemm_lds_kernel(GemmP p) {
    using C = LdsCfg<MODE, BM, BN, WTM, WTN, WK, KU>;
    constexpr int NT = C::NT, BK = C::BK, LDK = C::LDK, LDXB = C::LDXB, TI = C::TI, TJ = C::TJ;
    static_assert(PD >= 2, "two stages are stored before the first barrier");
    __shared__ __attribute__((aligned(16))) float lds[C::FLOATS];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wk = w / (C::WM * C::WN), wmn = w % (C::WM * C::WN), wm = wmn / C::WN, wn = wmn % C::WN;
    const int b = blockIdx.x, xcd = b & 7, l = b >> 3;
    const int tile_m = xcd * p.lds_mpx + l / p.tn, tile_n = l % p.tn;
    if (tile_m >= p.lds_tm) return;                               // workgroup-uniform
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const float* A = p.A + gm_slot_offset(p.a_slot);
    const float* B = p.B + gm_slot_offset(p.b_slot);

    constexpr int KQ = BK / 4;                                    // 16-byte units per k-contiguous row
    constexpr int XQ = BN / 4, XQA = BM / 4;                      // 16-byte units per x-contiguous row
    constexpr int LDXA = C::LDXA;
    constexpr int UA = C::A_KC ? BM * KQ : BK * XQA, CA = (UA + NT - 1) / NT;
    constexpr int UB = C::B_KC ? BN * KQ : BK * XQ, CB = (UB + NT - 1) / NT;
    // dw: B's real columns end at n_real; column n_real is the virtual ones column (bias gradient),
    // 1 for reduction rows >= ones_from.  Edge tiles (and only they) pay for the column masks.
    const int b_cols = (MODE == MODE_DW) ? p.n_real : p.N;
    const int ones_col = (MODE == MODE_DW && p.db) ? p.n_real : -1;
    const bool edge_m = (MODE == MODE_DW) && (m0 + BM > p.M);
    const bool edge_n = (MODE == MODE_DW) && (n0 + BN > b_cols);
    // per-thread base address of each 16-byte unit of the stage tiles (row clamps applied once)
    const float* baseA[CA];
    const float* baseB[CB];
#pragma unroll
    for (int i = 0; i < CA; ++i) {
        const int u = min(t + i * NT, UA - 1);
        if (C::A_KC) baseA[i] = A + (int64_t)min(m0 + u / KQ, p.M - 1) * p.lda + 4 * (u % KQ);
        else baseA[i] = A + min(m0 + 4 * (u % XQA), p.M - 4);    // + k * lda per stage
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
        const int u = min(t + i * NT, UB - 1);
        if (C::B_KC) baseB[i] = B + (int64_t)min(n0 + u / KQ, p.N - 1) * p.ldb + 4 * (u % KQ);
        else baseB[i] = B + min(n0 + 4 * (u % XQ), b_cols - 4);   // + k * ldb per stage
    }
    // One 16-byte unit of stage q.  The lambdas RETURN the value (a lambda that writes a captured
    // register array makes hipcc keep the array in scratch memory, with a vmcnt(0) behind every
    // load -- measured: 1.4 us per stage).  Branch-free: `inside` (the whole stage lies below K) is
    // a compile-time constant in the unrolled kernels, otherwise the k index is clamped.
    auto load_a = [&](const float* base, int i, int q, bool inside) -> float4 {
        if (!C::A_KC) {
            const int kk = q * BK + min(t + i * NT, UA - 1) / XQA;
            return *reinterpret_cast<const float4*>(base + (int64_t)(inside ? kk : min(kk, p.K - 1)) * p.lda);
        }
        if (inside) return *reinterpret_cast<const float4*>(base + q * BK);
        const int kq4 = 4 * (min(t + i * NT, UA - 1) % KQ);
        return *reinterpret_cast<const float4*>(base - kq4 + min(q * BK + kq4, p.K - 4));
    };
    auto load_b = [&](const float* base, int i, int q, bool inside) -> float4 {
        const int u = min(t + i * NT, UB - 1);
        if (C::B_KC) {
            if (inside) return *reinterpret_cast<const float4*>(base + q * BK);
            const int kq4 = 4 * (u % KQ);
            return *reinterpret_cast<const float4*>(base - kq4 + min(q * BK + kq4, p.K - 4));
        }
        const int kk = q * BK + u / XQ;
        return *reinterpret_cast<const float4*>(base + (int64_t)(inside ? kk : min(kk, p.K - 1)) * p.ldb);
    };
    // the K-tail zeroing select happens here, at the LDS write, never right behind the load
    auto store_a = [&](int buf, int i, int q, bool inside, float4 v) {
        const int u = t + i * NT;
        if (UA % NT != 0 && u >= UA) return;
        if (!C::A_KC) {
            const int kr = u / XQA, xq = u % XQA;
            const bool ok = (inside || q * BK + kr < p.K) && (!edge_m || m0 + 4 * xq < p.M);
            *reinterpret_cast<float4*>(&lds[buf * C::STAGE + kr * LDXA + 4 * xq]) = keep4(ok, v);
            return;
        }
        const int row = u / KQ, kq = u % KQ;
        *reinterpret_cast<float4*>(&lds[buf * C::STAGE + row * LDK + 4 * kq]) = inside ? v : keep4(q * BK + 4 * kq < p.K, v);
    };
    auto store_b = [&](int buf, int i, int q, bool inside, float4 v) {
        const int u = t + i * NT;
        if (UB % NT != 0 && u >= UB) return;
        float* Bs = lds + buf * C::STAGE + C::A_SZ;
        if (C::B_KC) {
            const int row = u / KQ, kq = u % KQ;
            *reinterpret_cast<float4*>(&Bs[row * LDK + 4 * kq]) = inside ? v : keep4(q * BK + 4 * kq < p.K, v);
        } else {
            const int kr = u / XQ, xq = u % XQ;
            if (MODE == MODE_DW && edge_n) {
                const int k = q * BK + kr, c0 = n0 + 4 * xq;
                const bool kin = inside || k < p.K;
                const float one = (kin && k >= p.ones_from) ? 1.f : 0.f;
                v.x = (c0 + 0 < b_cols) ? (kin ? v.x : 0.f) : ((c0 + 0 == ones_col) ? one : 0.f);
                v.y = (c0 + 1 < b_cols) ? (kin ? v.y : 0.f) : ((c0 + 1 == ones_col) ? one : 0.f);
                v.z = (c0 + 2 < b_cols) ? (kin ? v.z : 0.f) : ((c0 + 2 == ones_col) ? one : 0.f);
                v.w = (c0 + 3 < b_cols) ? (kin ? v.w : 0.f) : ((c0 + 3 == ones_col) ? one : 0.f);
                *reinterpret_cast<float4*>(&Bs[kr * LDXB + 4 * xq]) = v;
                return;
            }
            *reinterpret_cast<float4*>(&Bs[kr * LDXB + 4 * xq]) = inside ? v : keep4(q * BK + kr < p.K, v);
        }
    };
    // MFMA fragments of one 8-deep k group: lane (r, h) holds k = kb + 4h + j, j = 0..3
    auto frag_a = [&](int buf, int ku, int ti) -> float4 {
        if (!C::A_KC) {
            const float* q = &lds[buf * C::STAGE + ((wk * KU + ku) * 8 + 4 * h) * LDXA + wm * WTM + ti * 32 + r];
            return make_float4(q[0], q[LDXA], q[2 * LDXA], q[3 * LDXA]);
        }
        return *reinterpret_cast<const float4*>(
            &lds[buf * C::STAGE + (wm * WTM + ti * 32 + r) * LDK + (wk * KU + ku) * 8 + 4 * h]);
    };
    auto frag_b = [&](int buf, int ku, int tj) -> float4 {
        const float* Bs = lds + buf * C::STAGE + C::A_SZ;
        const int kb = (wk * KU + ku) * 8, col = wn * WTN + tj * 32 + r;
        if (C::B_KC) return *reinterpret_cast<const float4*>(&Bs[col * LDK + kb + 4 * h]);
        const float* q = &Bs[(kb + 4 * h) * LDXB + col];
        return make_float4(q[0], q[LDXB], q[2 * LDXB], q[3 * LDXB]);
    };

    constexpr int NDSR = KU * ((C::A_KC ? TI : 4 * TI) + (C::B_KC ? TJ : 4 * TJ));   // LDS read instructions per stage
    // a 32x32 wave tile alternates two accumulators MFMA by MFMA (consecutive MFMAs must not depend
    // on each other); bigger wave tiles have 2 or 4 accumulators anyway
    constexpr int NA = (TI * TJ == 1) ? 2 : 1;
    float4 ra[PD][CA], rb[PD][CB];                               // global -> LDS staging registers
    float4 fa[2][KU][TI], fb[2][KU][TJ];                         // fragments: this stage / next stage
    f32x16 acc[NA][TI][TJ];
#pragma unroll
    for (int c = 0; c < NA; ++c)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[c][ti][tj][i] = 0.f;
    // stage q lies wholly below K in the unrolled kernels iff q < NS - 1
#define GM_LDS_INSIDE(q) (NS > 0 && (q) < NS - 1)
#define GM_LDS_GLOAD(j, q)                                                         \
    _Pragma("unroll") for (int i = 0; i < CA; ++i) ra[j][i] = load_a(baseA[i], i, (q), GM_LDS_INSIDE(q)); \
    _Pragma("unroll") for (int i = 0; i < CB; ++i) rb[j][i] = load_b(baseB[i], i, (q), GM_LDS_INSIDE(q));
#define GM_LDS_LSTORE(j, buf, q)                                                   \
    _Pragma("unroll") for (int i = 0; i < CA; ++i) store_a((buf), i, (q), GM_LDS_INSIDE(q), ra[j][i]); \
    _Pragma("unroll") for (int i = 0; i < CB; ++i) store_b((buf), i, (q), GM_LDS_INSIDE(q), rb[j][i]);
#define GM_LDS_FRAGS(P, buf)                                                       \
    _Pragma("unroll") for (int ku = 0; ku < KU; ++ku) {                            \
        _Pragma("unroll") for (int ti = 0; ti < TI; ++ti) fa[P][ku][ti] = frag_a((buf), ku, ti); \
        _Pragma("unroll") for (int tj = 0; tj < TJ; ++tj) fb[P][ku][tj] = frag_b((buf), ku, tj); \
    }
#define GM_LDS_MFMA1(P, ku, comp, cidx)                                            \
    _Pragma("unroll") for (int ti = 0; ti < TI; ++ti)                              \
        _Pragma("unroll") for (int tj = 0; tj < TJ; ++tj)                          \
            acc[(NA == 2) ? (cidx) : 0][ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(   \
                fa[P][ku][ti].comp, fb[P][ku][tj].comp, acc[(NA == 2) ? (cidx) : 0][ti][tj], 0, 0, 0);
#define GM_LDS_MFMA(P)                                                             \
    _Pragma("unroll") for (int ku = 0; ku < KU; ++ku) {                            \
        GM_LDS_MFMA1(P, ku, x, 0) GM_LDS_MFMA1(P, ku, y, 1) GM_LDS_MFMA1(P, ku, z, 0) GM_LDS_MFMA1(P, ku, w, 1) \
    }
    // stage s: J = s % PD and P = s & 1 are compile-time after unrolling.  Loads and LDS writes are
    // UNCONDITIONAL (past-the-end stages read clamped addresses and write zeros into an idle
    // buffer): a guard would make hipcc drain vmcnt to 0 at its join.
#define GM_LDS_STAGE(s_, J, P)                                                     \
    {                                                                              \
        GM_LDS_FRAGS((P) ^ 1, ((s_) + 1) % 3)                                      \
        GM_LDS_LSTORE(((J) + 2) % PD, ((s_) + 2) % 3, (s_) + 2)                    \
        GM_LDS_GLOAD(((J) + 2) % PD, (s_) + 2 + PD)                                \
        GM_LDS_MFMA(P)                                                             \
        lds_stage_schedule<4 * KU * TI * TJ, NDSR, CA + CB, CA + CB>();            \
        __builtin_amdgcn_sched_barrier(0);                                         \
        __syncthreads();                                                           \
        __builtin_amdgcn_sched_barrier(0);                                         \
    }

    const int S = (p.K + BK - 1) / BK;
    GM_LDS_GLOAD(0, 0)
    GM_LDS_GLOAD(1 % PD, 1)
    GM_LDS_LSTORE(0, 0, 0)
    GM_LDS_LSTORE(1 % PD, 1, 1)
#pragma unroll
    for (int q = 2; q < PD + 2; ++q) { GM_LDS_GLOAD(q % PD, q) }
    __syncthreads();
    GM_LDS_FRAGS(0, 0)
    if constexpr (NS > 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) GM_LDS_STAGE(s, s % PD, s & 1)
    } else {
        constexpr int U = (PD % 2 == 0) ? PD : 2 * PD;           // lcm(2, PD)
        for (int s0 = 0; s0 < S; s0 += U) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if (s0 + j >= S) break;                           // workgroup-uniform
                GM_LDS_STAGE(s0 + j, j % PD, j & 1)
            }
        }
    }
#undef GM_LDS_STAGE
#undef GM_LDS_MFMA
#undef GM_LDS_MFMA1
#undef GM_LDS_FRAGS
#undef GM_LDS_GLOAD
#undef GM_LDS_LSTORE
#undef GM_LDS_INSIDE
    // partial tiles of the WK wave groups -> LDS (every stage buffer is idle after the last barrier)
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                float v = acc[0][ti][tj][i];
                if (NA == 2) v += acc[NA - 1][ti][tj][i];
                lds[(wk * BM + wm * WTM + ti * 32 + row) * BN + wn * WTN + tj * 32 + r] = v;
            }
    __syncthreads();
    if (p.vec_epi && !(MODE == MODE_DX && p.rp_dml)) {          // kernel-argument uniform: float4 groups (see store4)
        for (int e = t; e < BM * (BN / 4); e += NT) {
            const int row = e / (BN / 4), c4 = e % (BN / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int kk = 0; kk < WK; ++kk) {
                const float4 x = *reinterpret_cast<const float4*>(&lds[(kk * BM + row) * BN + 4 * c4]);
                v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
            }
            const int m = m0 + row, n = n0 + 4 * c4;
            if (m < p.M && n < p.N) store_group<MODE>(p, v, m, n, 0x7fffffff);
        }
        return;
    }
    for (int e = t; e < BM * BN; e += NT) {
        const int row = e / BN, col = e % BN;
        float v = 0.f;
#pragma unroll
        for (int kk = 0; kk < WK; ++kk) v += lds[(kk * BM + row) * BN + col];
        const int m = m0 + row, n = n0 + col;
        if (m < p.M && n < p.N) store_element<MODE>(p, v, m, n);
    }
}

// Tile shape of the LDS kernel for an M x N output: the fewest workgroup rounds on 256 CUs weighted
// by tile area (time per round), the larger tile on ties.  1: 64x64, 2: 32x64.  Measured on MI355X
// (profiles/r02_experiments.md): 128x64 tiles lose to two rounds of 64x64 on the 2048x784 output.
inline int lds_pick_cfg(int M, int N) {
    static int forced = -1;
    if (forced < 0) { const char* e = getenv("GM_LDS_CFG"); forced = e ? atoi(e) : 0; }
    if (forced) return forced;
    const int bm[2] = {64, 32}, bn[2] = {64, 64}, id[2] = {1, 2};
    long best = -1; int pick = 0;
    for (int i = 0; i < 2; ++i) {
        const long tiles = (long)((M + bm[i] - 1) / bm[i]) * ((N + bn[i] - 1) / bn[i]);
        const long cost = ((tiles + 255) / 256) * bm[i] * bn[i];
        if (best < 0 || cost < best) { best = cost; pick = id[i]; }
    }
    return pick;
}

template <int MODE, int BM, int BN, int WTM, int WTN, int WK, int KU, int PD>
int launch_lds_cfg(hipStream_t s, GemmP p) {
    using C = LdsCfg<MODE, BM, BN, WTM, WTN, WK, KU>;
    const int tm = (p.M + BM - 1) / BM, tn = (p.N + BN - 1) / BN;
    p.lds_tm = tm; p.tn = tn; p.lds_mpx = (tm + 7) / 8;
    const dim3 grid(8 * p.lds_mpx * tn), block(C::NT);
    // fully unrolled instantiations for the reduction lengths of this model (784, 400: image and
    // hidden widths); anything else takes the runtime loop
    constexpr int NS784 = (784 + C::BK - 1) / C::BK, NS400 = (400 + C::BK - 1) / C::BK;
    const int S = (p.K + C::BK - 1) / C::BK;
    static int unroll_on = -1;
    if (unroll_on < 0) { const char* e = getenv("GM_LDS_UNROLL"); unroll_on = e ? atoi(e) : 1; }
    if constexpr (MODE == MODE_DW) {
        // weight gradients reduce over the batch rows: fully unrolled stage counts for the row counts of
        // this model's steps (B, 2B for B = 256 ... 1024): counted waits and static LDS addressing, as for
        // the forward's K = 784 / 400 (the runtime loop drains vmcnt on its back-edge)
        if (unroll_on && S * C::BK == p.K) {
            switch (S) {
#define GM_LDS_DW_NS(n) case n: hipLaunchKernelGGL((gemm_lds_kernel<MODE, BM, BN, WTM, WTN, WK, KU, PD, n>), grid, block, 0, s, p); GM_LAUNCH_RET();
            GM_LDS_DW_NS(8) GM_LDS_DW_NS(16) GM_LDS_DW_NS(32) GM_LDS_DW_NS(64)
#undef GM_LDS_DW_NS
            default: break;
            }
        }
        hipLaunchKernelGGL((gemm_lds_kernel<MODE, BM, BN, WTM, WTN, WK, KU, PD, 0>), grid, block, 0, s, p);
        GM_LAUNCH_RET();
    }
    if (unroll_on && S == NS784) hipLaunchKernelGGL((gemm_lds_kernel<MODE, BM, BN, WTM, WTN, WK, KU, PD, NS784>), grid, block, 0, s, p);
    else if (unroll_on && S == NS400) hipLaunchKernelGGL((gemm_lds_kernel<MODE, BM, BN, WTM, WTN, WK, KU, PD, NS400>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_lds_kernel<MODE, BM, BN, WTM, WTN, WK, KU, PD, 0>), grid, block, 0, s, p);
    GM_LAUNCH_RET();
}

// Tile configuration of the LDS kernel for this launch, or 0 when the launch is not for it (the caller
// continues with the split-reduction kernels).  1: 64x64 tile = 2 x 1 waves of 32x64 (two accumulators
// each) x 4 reduction groups, BK = 32; 2: 32x64 tile = 1 x 2 waves of 32x32 x 4 reduction groups.
template <int MODE>
int lds_cfg_for(const GemmP& p, bool vec, bool xv) {
    if constexpr (MODE == MODE_DW) {
        // weight gradient: both operands are k-major (dA[k][m], X[k][n]); staged through LDS once per
        // workgroup instead of DPP-transposed in every wave.  CORRECT BUT SLOWER than the
        // split-reduction kernel in this form (2048 rows: 43.0 vs 31.5 us, vendor 20.6; 512 rows: 13.9
        // vs 10.6, vendor 8.1 -- profiles/r02_experiments.md 4c), so it is off unless asked for.
        static int min_k = -1;
        if (min_k < 0) { const char* e = getenv("GM_LDS_DW_MIN_K"); min_k = e ? atoi(e) : (1 << 30); }
        if (p.K < min_k || !xv || p.M < 32 || p.n_real < 32 || p.M % 4 != 0 || p.n_real % 4 != 0) return 0;
    } else {
        static int min_m = -1;
        if (min_m < 0) { const char* e = getenv("GM_LDS_MIN_M"); min_m = e ? atoi(e) : 1024; }
        if (p.M < min_m || p.K < 64 || !vec || (MODE == MODE_DX && !xv) || p.N < 32) return 0;
    }
    const int cfg = lds_pick_cfg(p.M, p.N);
    return (cfg == 1 || cfg == 2) ? cfg : 0;
}

template <int MODE>
int launch_lds(hipStream_t s, const GemmP& p, int cfg) {
    if (cfg == 1) return launch_lds_cfg<MODE, 64, 64, 32, 64, 4, 1, 4>(s, p);
    return launch_lds_cfg<MODE, 32, 64, 32, 32, 4, 1, 4>(s, p);
}

template <int MODE, bool VEC, int WAVES, int G, bool XV>
__global__ __launch_bounds__(WAVES * 64) void gemm_kernel(GemmP p) {
    __shared__ __attribute__((aligned(16))) float red[WAVES * 32 * 32];      // 64 / 32 KB: one 32x32 partial tile per wave

    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    int tile_m = blockIdx.y, tile_n = blockIdx.x;
    if (p.xr > 0) {
        // block b is dispatched to XCD b % 8 (observed placement; only speed depends on it)
        const int b = blockIdx.x, xcd = b & 7, l = b >> 3;
        const int xi = xcd / p.xc, xj = xcd % p.xc;
        const int mlo = (xi * p.tm) / p.xr, mhi = ((xi + 1) * p.tm) / p.xr;
        const int nlo = (xj * p.tn) / p.xc, nhi = ((xj + 1) * p.tn) / p.xc;
        const int nn = nhi - nlo;
        if (nn <= 0 || l >= (mhi - mlo) * nn) return;
        tile_m = mlo + l / nn;
        tile_n = nlo + l % nn;
    }
    const int m0 = tile_m * TM, n0 = tile_n * TN;

    const float* A = p.A + gm_slot_offset(p.a_slot);
    const float* B = p.B + gm_slot_offset(p.b_slot);
    const int b_cols = (MODE == MODE_DW) ? p.n_real : p.N;       // real columns of B
    const int ones_col = (MODE == MODE_DW && p.db) ? p.n_real : -1;
    const int nchunks = (p.K + 7) >> 3;

    auto load_a = [&](int c) -> float4 {
        const int kb = 8 * c + 4 * h;
        if (MODE == MODE_DW) {
            if (XV) return raw_xc4(A, p.lda, m0, p.M, c, p.K, lane);
            return raw_xc(A, p.lda, m0 + r, p.M, kb, p.K);
        }
        return raw_kc<VEC>(A, p.lda, m0 + r, p.M, kb, p.K);
    };
    auto load_b = [&](int c) -> float4 {
        const int kb = 8 * c + 4 * h;
        if (MODE == MODE_FWD) return raw_kc<VEC>(B, p.ldb, n0 + r, p.N, kb, p.K);
        if (XV) return raw_xc4(B, p.ldb, n0, b_cols, c, p.K, lane);
        return raw_xc(B, p.ldb, n0 + r, b_cols, kb, p.K);
    };
    auto fix_a = [&](float4 v, int c) -> float4 {
        const int kb = 8 * c + 4 * h;
        if (MODE == MODE_DW) return fix_xc(XV ? quad_transpose(v, lane) : v, m0 + r, p.M, kb, p.K, -1);
        return fix_kc(v, m0 + r, p.M, kb, p.K);
    };
    auto fix_b = [&](float4 v, int c) -> float4 {
        const int kb = 8 * c + 4 * h;
        if (MODE == MODE_FWD) return fix_kc(v, n0 + r, p.N, kb, p.K);
        return fix_xc(XV ? quad_transpose(v, lane) : v, n0 + r, b_cols, kb, p.K, ones_col, p.ones_from);
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    // chunk schedule of this wave: position q = 0,1,2,... -> chunk id.  Round-robin (q*16 + w)
    // spreads adjacent chunks over waves; blocked (w*cpw + q) gives each wave a contiguous k-range
    // so that its consecutive 16-byte loads fall into the same 128-byte lines.
    const int cstep = p.cpw > 0 ? 1 : WAVES;
    const int cbase = p.cpw > 0 ? w * p.cpw : w;
    const int cend = p.cpw > 0 ? min(nchunks, (w + 1) * p.cpw) : nchunks;
    // Batches of G chunks per wave: G UNCONDITIONAL loads back to back (positions past the wave's
    // range re-read its last chunk; addresses are always valid), then the MFMAs consume them in
    // order as they land (the compiler emits counted s_waitcnt vmcnt(2*(G-1-i))).  For K = 784 and
    // 16 waves one batch is the wave's whole k-range.  History (profiles/r01_experiments.md): with
    // guarded loads hipcc put s_waitcnt vmcnt(0) behind every load -- a 3.5 us load phase made of
    // 14 serialized round trips that did not overlap the 2.75 us MFMA chain.
    const int nq = (cend - cbase + cstep - 1) / cstep;      // chunk positions owned by this wave
    for (int q0 = 0; q0 < nq; q0 += G) {
        float4 ra[G], rb[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int cc = cbase + min(q0 + i, nq - 1) * cstep;
#if defined(GM_ABLATE) && GM_ABLATE == 2      // experiment: no operand loads (MFMA chain only)
            ra[i] = make_float4(1.f, 2.f, 3.f, 4.f); rb[i] = make_float4(1.f, 1.f, 1.f, 1.f);
#else
            ra[i] = load_a(cc); rb[i] = load_b(cc);
#endif
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int q = q0 + i;
            if (q < nq) {                                    // wave-uniform
#if defined(GM_ABLATE) && GM_ABLATE == 1      // experiment: loads only (keep them live, no MFMA)
                asm volatile("" ::"v"(ra[i].x), "v"(ra[i].y), "v"(ra[i].z), "v"(ra[i].w),
                             "v"(rb[i].x), "v"(rb[i].y), "v"(rb[i].z), "v"(rb[i].w));
#else
                const int cq = cbase + q * cstep;
                const float4 fa = fix_a(ra[i], cq), fb = fix_b(rb[i], cq);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc, 0, 0, 0);
#endif
            }
        }
    }

    // cross-wave reduction through LDS
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        red[(w * 32 + row) * 32 + r] = acc[i];
    }
    __syncthreads();
    reduce_and_store<MODE, WAVES>(p, red, t, m0, n0);
}

// ------------------------------------------------------------------------------------------
// Variant on v_mfma_f32_16x16x4_f32 (default; GM_MFMA16=0 selects the 32x32x2 kernel above): same
// decomposition, but a wave's
// operand fragments are 16 rows x 16 k per instruction -- lane (i = lane&15, g = lane>>4) loads the
// 4 consecutive k = 16c+4g..+3 of row i, so ONE load instruction touches 16 cache lines with 64
// useful bytes each (the 32x32x2 form touches 32 lines with 32 bytes each): half the L1 tag
// lookups per byte.  The wave keeps four independent 16x16 accumulators (2x2 sub-tiles of the
// 32x32 tile), so the 40-cycle dependent latency of the 16x16x4 MFMA is always covered.
// ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 raw_xc4_16(const float* __restrict__ P, int64_t ld, int x0, int X,
                                             int c, int K, int lane) {
    const int e = lane & 3, q = (lane >> 2) & 3, g = lane >> 4;
    const int k = min(16 * c + 4 * g + e, K - 1);
    const int x = min(x0 + 4 * q, X - 4);
    return *reinterpret_cast<const float4*>(P + (int64_t)k * ld + x);
}

// DIRECT fragment of an x-contiguous operand (GM_XDIRECT): lane (i = lane & 15, g = lane >> 4) loads the four
// dwords P[(kb + j) * ld + x], j = 0..3, kb = 16c + 4g -- exactly the element the j-th MFMA of the chunk wants
// from this lane, so there is nothing to transpose (each load instruction is 4 k-rows x 64 coalesced bytes).
// The 16-byte + quad-transpose form above costs ~30 VALU instructions per fragment; ISA count of the dW loop:
// 250 VALU instructions per 32 MFMAs, and the ablation without operand loads still took 28 us of the 36 us
// K = 2048 launch (pipe time of its MFMAs: 14.4 us) -- the transposes and the MFMAs do not overlap.
__device__ __forceinline__ float4 raw_xd(const float* __restrict__ P, int64_t ld, int x, int X, int kb, int K) {
    const float* col = P + min(x, X - 1);
    return make_float4(col[(int64_t)min(kb + 0, K - 1) * ld], col[(int64_t)min(kb + 1, K - 1) * ld],
                       col[(int64_t)min(kb + 2, K - 1) * ld], col[(int64_t)min(kb + 3, K - 1) * ld]);
}

// ------------------------------------------------------------------------------------------
// Weight gradient with INTERLEAVED fragments.  Both operands of dW[m][n] = sum_k dA[k][m] X[k][n] are
// contiguous along the OUTPUT index, while a 16x16x4 MFMA wants from lane (i = lane & 15, g = lane >> 4) the
// element (row i, k = g).  The 16-byte-load form above loads 4 output indices of one k per lane and transposes
// 4x4 blocks across lane quads (250 VALU instructions per 32 MFMAs, and they do not overlap the MFMA pipe:
// profiles/r03_experiments.md 4b).  Here lane (i, g) loads the W consecutive elements x0 + W*i .. + W-1 of row
// k = 16c + 4s + g (one W-dword load per operand per k-step s) and the j-th of them feeds MFMA (.., j): output
// sub-tile (e, f) then holds rows m0 + MI*i' + e and columns n0 + NI*j' + f -- every output exactly once, in an
// interleaved order that only the write into the reduction buffer has to know.  No cross-lane traffic at all.
// Lanes whose W elements do not all exist (last tile of a row / column, the virtual ones column) take a per-
// element path; rows k >= K are zeroed in the A fragment only.
// ------------------------------------------------------------------------------------------
template <int W> struct __attribute__((aligned(4))) ILV { float v[W]; };

// element j of a W-dword load, j a per-lane (loop-invariant) index; j >= W: `other`
template <int W>
__device__ __forceinline__ float il_pick(const ILV<W>& t, int j, float other) {
    float v = other;
#pragma unroll
    for (int e = 0; e < W; ++e) v = (j == e) ? t.v[e] : v;
    return v;
}

// Two consecutive weight-gradient outputs C(m, n), C(m, n + 1), n even, both real columns: store_element's arithmetic
// per element, 8-byte accesses (p.vec_epi: C and the Adam arrays are 16-byte aligned, ldc % 4 == 0).
__device__ __forceinline__ void store2_dw(const GemmP& p, float2 v, int m, int n) {
    const int64_t o = (int64_t)m * p.ldc + n;
    float2* cp = reinterpret_cast<float2*>(p.C + o);
    if (p.accumulate) { const float2 c = *cp; v.x += c.x; v.y += c.y; }
    *cp = v;
    if (p.adam.enabled) {
        const int64_t si = gm_slot_index(p.adam.sched_slot);
        const float step_size = p.adam.sched[2 * si], bc2_sqrt = p.adam.sched[2 * si + 1];
        float2* pp = reinterpret_cast<float2*>(p.adam.pW + o);
        float2* mm = reinterpret_cast<float2*>(p.adam.mW + o);
        float2* vv = reinterpret_cast<float2*>(p.adam.vW + o);
        float2 P = *pp, M = *mm, V = *vv;
        adam_update(P.x, v.x, M.x, V.x, step_size, bc2_sqrt, p.adam.omb1, p.adam.b2, p.adam.omb2, p.adam.eps, p.adam.wd, p.adam.clamp);
        adam_update(P.y, v.y, M.y, V.y, step_size, bc2_sqrt, p.adam.omb1, p.adam.b2, p.adam.omb2, p.adam.eps, p.adam.wd, p.adam.clamp);
        *pp = P; *mm = M; *vv = V;
    }
}

// Weight-gradient tiles of more than one 32 x 32 block (32 x 48, 48 x 32, 32 x 64 ...) in ONE pass: every wave leaves
// its whole partial tile in LDS (16 images of 16 MI x 16 NI floats: 96 KB for the 48-wide tiles), one barrier, and each
// thread sums one PAIR of neighbouring columns over the sixteen images (wave order, as before: bit-identical) and runs
// the epilogue once.  The block-by-block form (reduce_and_store) makes two or three trips -- barrier, sum, Adam state
// in, parameters out -- one behind the other, and each trip is a memory round trip.  ILO: interleaved accumulator
// layout of gemm16_dw_il / gemm16_dw_dma.
template <int MI, int NI, bool ILO>
__device__ __forceinline__ void dw_reduce_onepass(const GemmP& p, float* red, f32x4 (&acc)[MI][NI], int m0, int n0,
                                                  bool sync_first) {
    constexpr int RT = 16 * MI, CT = 16 * NI, IMG = RT * CT, G = IMG / 2;
    static_assert(G <= 1024, "one pair of columns per thread");
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g4 = lane >> 4;
    if (sync_first) __syncthreads();
    float* img = red + w * IMG;
#pragma unroll
    for (int e = 0; e < MI; ++e)
#pragma unroll
        for (int f = 0; f < NI; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = ILO ? MI * (4 * g4 + r) + e : 16 * e + 4 * g4 + r;
                const int col = ILO ? NI * i16 + f : 16 * f + i16;
                img[row * CT + col] = acc[e][f][r];
            }
    __syncthreads();
    if (t >= G) return;
    const int row = t / (CT / 2), c2 = t % (CT / 2);
    float2 v = make_float2(0.f, 0.f);
#pragma unroll
    for (int ww = 0; ww < 16; ++ww) {
        const float2 x = *reinterpret_cast<const float2*>(&red[ww * IMG + row * CT + 2 * c2]);
        v.x += x.x; v.y += x.y;
    }
    const int m = m0 + row, n = n0 + 2 * c2;
    if (m >= p.M) return;
    if (n + 1 < p.n_real) { store2_dw(p, v, m, n); return; }
    if (n < p.N) store_element<MODE_DW>(p, v.x, m, n);       // the ones column (bias gradient) / a ragged edge
    if (n + 1 < p.N) store_element<MODE_DW>(p, v.y, m, n + 1);
}

// Cross-wave reduction + epilogue shared by the interleaved-fragment weight-gradient bodies (gemm16_dw_il,
// gemm16_dw_dma): accumulator (e, f) register r of lane (i16, g4) is output (row MI*(4*g4 + r) + e, column
// NI*i16 + f) of the tile; one 32x32 block of the tile at a time through the first 64 KB of `red`.
// sync_first: the buffer was in use inside the reduction loop (DMA rings): everybody must be out of it first.
template <int MI, int NI>
__device__ __forceinline__ void dw_il_reduce(const GemmP& p, float* red, f32x4 (&acc)[MI][NI], int m0, int n0,
                                             bool sync_first) {
    constexpr int WAVES = 16;
    if (p.vec_epi) {                                         // kernel-argument uniform
        dw_reduce_onepass<MI, NI, true>(p, red, acc, m0, n0, sync_first);
        return;
    }
    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g4 = lane >> 4;
    if (sync_first) __syncthreads();
#pragma unroll
    for (int bm = 0; bm < (MI + 1) / 2; ++bm)
#pragma unroll
        for (int bnk = 0; bnk < (NI + 1) / 2; ++bnk) {
            if (bm + bnk > 0) __syncthreads();               // previous block fully consumed
#pragma unroll
            for (int e = 0; e < MI; ++e)
#pragma unroll
                for (int f = 0; f < NI; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = MI * (4 * g4 + r) + e, col = NI * i16 + f;
                        if ((row >> 5) == bm && (col >> 5) == bnk)
                            red[(w * 32 + (row & 31)) * 32 + (col & 31)] = acc[e][f][r];
                    }
            __syncthreads();
            reduce_and_store<MODE_DW, WAVES, (MI > 1 ? 32 : 16)>(p, red, t, m0 + 32 * bm, n0 + 32 * bnk,
                                                                 (2 * bnk + 1 < NI) ? 0x7fffffff : n0 + 32 * bnk + 16,
                                                                 (2 * bm + 1 < MI) ? 32 : 16);
        }
}

template <int MI, int NI, bool OF, int FOLD>
__device__ __forceinline__ void gemm16_dw_il(const GemmP& p, float* red, int bx, int by, float* sds,
                                             const FoldP* fold) {
    constexpr int WAVES = 16;
    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g4 = lane >> 4;
    const int m0 = by * (16 * MI), n0 = bx * (16 * NI);
    const int b_cols = p.n_real;
    const int ones_col = p.db ? p.n_real : -1;
    const int nchunks = (p.K + 15) >> 4;
    const int lda = (int)p.lda, ldb = (int)p.ldb;           // (K - 1) * ld + width < 2^31: checked by the host
    // This lane's W elements start at column am / bn; the load itself starts at a column clamped into the row
    // (always legal), `shift` columns to the left of the wanted one: 0 everywhere except in the last tile.
    const int am = m0 + MI * i16, bn = n0 + NI * i16;
    const int a_x = max(min(am, p.M - MI), 0), b_x = max(min(bn, b_cols - NI), 0);
    const int a_shift = am - a_x, b_shift = bn - b_x;
    const float* pa = p.A + gm_slot_offset(p.a_slot) + a_x;
    const float* pb = p.B + gm_slot_offset(p.b_slot) + b_x;
    const int a_last = (p.K - 1) * lda, b_last = (p.K - 1) * ldb;
    // workgroup-uniform: a tile whose every lane is in range over a reduction of whole chunks needs no fix-up
    const bool edge = (m0 + 16 * MI > p.M) || (n0 + 16 * NI > b_cols) || (p.K & 15) || OF;

    float fw[FOLD == 1 ? MI : 1];                            // folded head: w2 of this lane's A columns
    if constexpr (FOLD == 1) {
#pragma unroll
        for (int e = 0; e < MI; ++e) fw[e] = (am + e < p.M) ? p.fold_w2[min(am + e, p.M - 1)] : 0.f;
    }

    f32x4 acc[MI][NI];
#pragma unroll
    for (int e = 0; e < MI; ++e)
#pragma unroll
        for (int f = 0; f < NI; ++f) acc[e][f] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nq = (nchunks - w + WAVES - 1) / WAVES;        // chunks w, w+16, ... of this wave
    auto load_chunk = [&](int c, ILV<MI> (&ra)[4], ILV<NI> (&rb)[4]) {
        const int k0 = 16 * c + g4;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            ra[s] = *reinterpret_cast<const ILV<MI>*>(pa + min((k0 + 4 * s) * lda, a_last));
            rb[s] = *reinterpret_cast<const ILV<NI>*>(pb + min((k0 + 4 * s) * ldb, b_last));
        }
        // (measured: a sched_barrier here, which keeps the chunk's eight loads together ahead of the MFMAs -- left
        // alone the scheduler sinks each pair next to its first use, four waits per chunk -- is SLOWER: 26.1 -> 27.5 us
        // at 2048 rows; so is the double-buffered loop below with it, 28.3)
    };
    auto mfma_step = [&](const float (&fa)[MI], const float (&fb)[NI]) {
#pragma unroll
        for (int e = 0; e < MI; ++e)
#pragma unroll
            for (int f = 0; f < NI; ++f)
                acc[e][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[e], fb[f], acc[e][f], 0, 0, 0);
    };
    // EDGE (workgroup-uniform, its own copy of the loop): select the wanted element of the clamped load, zero what
    // does not exist, put the ones column in; the interior copy feeds the loaded registers straight to the MFMAs.
    auto consume = [&](const ILV<MI> (&ra)[4], const ILV<NI> (&rb)[4], int c, auto edge_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = 16 * c + 4 * s + g4;
            float fa[MI], fb[NI];
#pragma unroll
            for (int e = 0; e < MI; ++e) {
                float v = ra[s].v[e];
                if constexpr (EDGE) v = il_pick<MI>(ra[s], (am + e < p.M) ? e + a_shift : MI, 0.f);
                if constexpr (FOLD == 1) v = (v > 0.f) ? sds[min(k, p.K - 1)] * fw[e] : 0.f;
                if constexpr (EDGE) v = (k < p.K) ? v : 0.f;
                fa[e] = v;
            }
#pragma unroll
            for (int f = 0; f < NI; ++f) {
                float v = rb[s].v[f];
                if constexpr (EDGE) {
                    float one = 1.f;
                    if constexpr (OF) one = (k >= p.ones_from) ? 1.f : 0.f;
                    v = il_pick<NI>(rb[s], (bn + f < b_cols) ? f + b_shift : NI, (bn + f == ones_col) ? one : 0.f);
                }
                fb[f] = v;
            }
            mfma_step(fa, fb);
        }
    };
    auto run = [&](auto edge_tag) {
        int q_first = 0;
        if constexpr (FOLD == 1) {
            // the workgroup's dS rows are rebuilt BEHIND the first chunk's operand loads (gm_head.h); waves
            // without a chunk still take the barrier
            ILV<MI> ra[4]; ILV<NI> rb[4];
            const bool have = nq > 0;
            if (have) load_chunk(w, ra, rb);
            fold_fill_lds(*fold, sds, fold->R);
            if (have) consume(ra, rb, w, edge_tag);
            q_first = 1;
        }
#if GM_DW_IL_PREFETCH
        // Double-buffered: chunk q + 1's loads are in flight while chunk q's MFMAs run (two chunks per trip, static
        // buffers; past-the-end positions re-load the wave's last chunk -- a valid, unused read).  With the
        // transposes gone the loop's VALU + MFMA work is ~12 us of a 26 us launch at 2048 rows: the rest is exposed
        // load latency, one round trip per chunk.
        if (nq - q_first >= 2) {                              // wave uniform
            ILV<MI> ra0[4], ra1[4]; ILV<NI> rb0[4], rb1[4];
            const int last = w + (nq - 1) * WAVES;
            load_chunk(w + q_first * WAVES, ra0, rb0);
            int q = q_first;
            for (; q + 1 < nq; q += 2) {
                load_chunk(w + (q + 1) * WAVES, ra1, rb1);
                consume(ra0, rb0, w + q * WAVES, edge_tag);
                load_chunk(min(w + (q + 2) * WAVES, last), ra0, rb0);
                consume(ra1, rb1, w + (q + 1) * WAVES, edge_tag);
            }
            if (q < nq) consume(ra0, rb0, w + q * WAVES, edge_tag);
            return;
        }
#endif
        for (int q = q_first; q < nq; ++q) {
            ILV<MI> ra[4]; ILV<NI> rb[4];
            const int cc = w + q * WAVES;
            load_chunk(cc, ra, rb);
            consume(ra, rb, cc, edge_tag);
        }
    };
    if (edge) run(std::true_type{}); else run(std::false_type{});

    dw_il_reduce<MI, NI>(p, red, acc, m0, n0, false);
}

// ------------------------------------------------------------------------------------------
// Weight gradient with the operand chunks brought in by LDS-DMA (round 4).  What round 3 could not explain --
// "a 32 x 48 tile streams 655 KB through its CU at ~25 GB/s whatever the instruction mix" -- is a price PER VECTOR
// MEMORY INSTRUCTION: one CU retires about one wave-wide load per 40 cycles, whatever it carries (measured,
// tools/slab_probe fill: 8 B per lane 22-24 GB/s per CU, 12 B 32-34, 16 B 43-45, global_load_lds_dwordx4 53-55;
// profiles/r04_slab_probe.md).  The interleaved form above loads 8 / 12 bytes per lane -- eight instructions per 5 KB
// chunk.  Here a wave's chunk (16 reduction rows x the tile's 16 MI + 16 NI columns = MI + NI KB) arrives as MI + NI
// pieces of 64 lanes x 16 bytes, straight into the wave's PRIVATE LDS buffer (no VGPR round trip, no workgroup
// barrier: the issuing wave's own vmcnt orders its reads), and the fragments are read back in the same interleaved
// order as gemm16_dw_il (lane i takes MI / NI consecutive outputs of row k and feeds element j to sub-tile j), so the
// accumulator layout, the reduction and every epilogue are shared with it.  One buffer per wave is enough: the chunk's
// fragments are in registers (4 k-steps x (MI + NI) values) before the next chunk's pieces are issued into the same
// buffer, and they land under this chunk's MFMAs; the other three waves of the SIMD cover the rest.
// Out-of-range columns are clamped into the row in whole 16-byte groups (M, n_real are multiples of 4: a clamped
// group never holds a real column) and only feed outputs nobody stores; the virtual ones column and rows past K are
// selects on the fragment, compiled into their own copy of the loop for the (workgroup-uniform) tiles that need them.
// ------------------------------------------------------------------------------------------
template <int MI, int NI, bool OF, int FOLD>
__device__ __forceinline__ void gemm16_dw_dma(const GemmP& p, float* red, int bx, int by, float* sds,
                                              const FoldP* fold) {
    constexpr int WAVES = 16, NP = MI + NI, CH = NP * 256;   // pieces / floats of a wave's chunk buffer
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i16 = lane & 15, g4 = lane >> 4;
    const int m0 = by * (16 * MI), n0 = bx * (16 * NI);
    const int b_cols = p.n_real;
    const int ones_col = p.db ? p.n_real : -1;
    const int nchunks = (p.K + 15) >> 4;
    float* buf = red + w * CH;
    const uint32_t buf_b = (uint32_t)(uintptr_t)buf;
    const int am = m0 + MI * i16, bn = n0 + NI * i16;        // this lane's first A / B column

    // piece j: float4 units [64 j, 64 j + 64) of the chunk image [16][16 MI] ++ [16][16 NI]
    const float* src[NP]; int srow[NP], sld[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const bool isA = j < MI;
        const int idx = (isA ? j : j - MI) * 64 + lane;
        const int row = isA ? idx / (4 * MI) : idx / (4 * NI);
        const int c4 = isA ? idx % (4 * MI) : idx % (4 * NI);
        const int col = isA ? min(m0 + 4 * c4, p.M - 4) : min(n0 + 4 * c4, b_cols - 4);
        src[j] = (isA ? p.A + gm_slot_offset(p.a_slot) : p.B + gm_slot_offset(p.b_slot)) + col;
        sld[j] = isA ? (int)p.lda : (int)p.ldb;
        srow[j] = row;
    }
    auto issue = [&](int c) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int k = min(16 * c + srow[j], p.K - 1);
            slab::glds16(src[j] + (int64_t)k * sld[j], buf_b + j * 1024u);
        }
    };

    float fw[FOLD == 1 ? MI : 1];                            // folded head: w2 of this lane's A columns
    if constexpr (FOLD == 1) {
#pragma unroll
        for (int e = 0; e < MI; ++e) fw[e] = (am + e < p.M) ? p.fold_w2[min(am + e, p.M - 1)] : 0.f;
    }
    f32x4 acc[MI][NI];
#pragma unroll
    for (int e = 0; e < MI; ++e)
#pragma unroll
        for (int f = 0; f < NI; ++f) acc[e][f] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nq = (nchunks - w + WAVES - 1) / WAVES;        // chunks w, w+16, ... of this wave
    // workgroup-uniform: does this tile hold the ones column / does the reduction end inside a chunk?
    const bool special = (ones_col >= n0 && ones_col < n0 + 16 * NI) || (p.K & 15) || OF;
    auto run = [&](auto special_tag) {
        constexpr bool SPECIAL = decltype(special_tag)::value;
        if (nq > 0) issue(w);
        if constexpr (FOLD == 1) fold_fill_lds(*fold, sds, fold->R);   // behind the first chunk's loads; ends with a barrier
        for (int q = 0; q < nq; ++q) {
            const int c = w + q * WAVES;
            slab::wait_vm<0>();                              // this wave's pieces have landed
            ILV<MI> ra[4]; ILV<NI> rb[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int row = 4 * s + g4;
                ra[s] = *reinterpret_cast<const ILV<MI>*>(buf + row * (16 * MI) + MI * i16);
                rb[s] = *reinterpret_cast<const ILV<NI>*>(buf + 256 * MI + row * (16 * NI) + NI * i16);
            }
            slab::wait_lgkm0();                              // ... and are in registers: the buffer is free
            if (q + 1 < nq) issue(c + WAVES);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = 16 * c + 4 * s + g4;
                float fa[MI], fb[NI];
#pragma unroll
                for (int e = 0; e < MI; ++e) {
                    float v = ra[s].v[e];
                    if constexpr (FOLD == 1) v = (v > 0.f) ? sds[min(k, p.K - 1)] * fw[e] : 0.f;
                    if constexpr (SPECIAL) v = (k < p.K) ? v : 0.f;
                    fa[e] = v;
                }
#pragma unroll
                for (int f = 0; f < NI; ++f) {
                    float v = rb[s].v[f];
                    if constexpr (SPECIAL) {
                        float one = 1.f;
                        if constexpr (OF) one = (k >= p.ones_from) ? 1.f : 0.f;
                        v = (bn + f == ones_col) ? one : v;
                    }
                    fb[f] = v;
                }
#pragma unroll
                for (int e = 0; e < MI; ++e)
#pragma unroll
                    for (int f = 0; f < NI; ++f)
                        acc[e][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[e], fb[f], acc[e][f], 0, 0, 0);
            }
        }
    };
    if (special) run(std::true_type{}); else run(std::false_type{});
    dw_il_reduce<MI, NI>(p, red, acc, m0, n0, true);         // the ring shares `red`: everybody out of the loop first
}

// LDS floats of the 16-wave kernels: the 64 KB block-by-block reduction buffer; for weight gradients of multi-block
// tiles the sixteen whole-tile images of dw_reduce_onepass; the DMA form's sixteen chunk buffers
template <int MODE, int IL, int MI, int NI> struct RedSize {
    static constexpr int base = 16 * 32 * 32;
    static constexpr int onepass = (MODE == MODE_DW && MI * NI > 4) ? 16 * 256 * MI * NI : 0;
    static constexpr int dma = (MODE == MODE_DW && IL == 2) ? 16 * (MI + NI) * 256 : 0;
    static constexpr int m1 = base > onepass ? base : onepass;
    static constexpr int value = m1 > dma ? m1 : dma;
};

// MI x NI = number of 16-row / 16-column sub-tiles per wave: (2,2) is the 32x32 tile; (2,4) and
// (4,2) are 32x64 / 64x32 tiles used when a launch would otherwise have more tiles than CUs (two
// rounds of one workgroup per CU): one round, 6 fragment loads per 32 MFMAs instead of 4 per 16.
// OF: the ones column starts at reduction row p.ones_from (WGAN-GP's stacked dW only); compiled out
// otherwise -- the four extra compares per fragment cost the generator's dW pair 1.2 us when they ran
// unconditionally.
// FOLD (folded critic head, gm_head.h): 1 = weight gradient whose A operand dH[k][x] is formed from
// h[k][x], sds[k] (dS of reduction row k) and w2[x]; 2 = input gradient whose A operand dH[m][k] is
// formed from h[m][k], sds[m - m0] and w2[k].  sds: the workgroup's LDS copy of dS.
template <int MODE, bool VEC, int WAVES, int G, bool XV, int MI, int NI, bool OF = false, int FOLD = 0, int IL = 0>
__device__ __forceinline__ void gemm16_body(const GemmP& p, float* red, int bx, int by,
                                            float* sds = nullptr, const FoldP* fold = nullptr) {
    static_assert(FOLD == 0 || (FOLD == 1 && MODE == MODE_DW && XV) || (FOLD == 2 && MODE == MODE_DX && VEC),
                  "folded head: 16-byte operand paths only");
    // IL: the launch chose the interleaved-fragment form (its own kernel instantiations: as a run-time branch
    // inside the shared kernels it cost the bs=256 step 1.5 us in registers and code it never runs)
    // IL: 1 = interleaved fragments loaded into VGPRs, 2 = the same fragments through LDS-DMA
    if constexpr (IL == 2 && MODE == MODE_DW && XV && WAVES == 16 && FOLD != 2) {
        gemm16_dw_dma<MI, NI, OF, FOLD>(p, red, bx, by, sds, fold);
        return;
    }
    if constexpr (IL == 1 && GM_DW_IL && MODE == MODE_DW && XV && WAVES == 16 && FOLD != 2) {
        gemm16_dw_il<MI, NI, OF, FOLD>(p, red, bx, by, sds, fold);
        return;
    }
    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g4 = lane >> 4;
    const int m0 = by * (16 * MI), n0 = bx * (16 * NI);

    const float* A = p.A + gm_slot_offset(p.a_slot);
    const float* B = p.B + gm_slot_offset(p.b_slot);
    const int b_cols = (MODE == MODE_DW) ? p.n_real : p.N;
    const int ones_col = (MODE == MODE_DW && p.db) ? p.n_real : -1;
    const int nchunks = (p.K + 15) >> 4;
    constexpr bool xdirect = (GM_XDIRECT == 1) || (GM_XDIRECT == 2 && MODE == MODE_DX);

    // folded head: what stays fixed per lane across the reduction
    float4 fw[FOLD == 1 ? MI : 1];
    float fds[FOLD == 2 ? MI : 1];
    if constexpr (FOLD == 1) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)                       // w2 of this lane's four A columns
            fw[mi] = *reinterpret_cast<const float4*>(
                p.fold_w2 + min(m0 + 16 * mi + 4 * ((lane >> 2) & 3), p.M - 4));
    }
    auto load_a = [&](int c, int mi) -> float4 {
        const int kb = 16 * c + 4 * g4, x0 = m0 + 16 * mi;
        if (MODE == MODE_DW) {
            if (XV && xdirect && FOLD == 0) return raw_xd(A, p.lda, x0 + i16, p.M, kb, p.K);
            if (XV) return raw_xc4_16(A, p.lda, x0, p.M, c, p.K, lane);
            return raw_xc(A, p.lda, x0 + i16, p.M, kb, p.K);
        }
        return raw_kc<VEC>(A, p.lda, x0 + i16, p.M, kb, p.K);
    };
    auto load_b = [&](int c, int ni) -> float4 {
        const int kb = 16 * c + 4 * g4, x0 = n0 + 16 * ni;
        if (MODE == MODE_FWD) return raw_kc<VEC>(B, p.ldb, x0 + i16, p.N, kb, p.K);
        if (XV && xdirect) return raw_xd(B, p.ldb, x0 + i16, b_cols, kb, p.K);
        if (XV) return raw_xc4_16(B, p.ldb, x0, b_cols, c, p.K, lane);
        return raw_xc(B, p.ldb, x0 + i16, b_cols, kb, p.K);
    };
    auto fix_a = [&](float4 v, int c, int mi, float4 wk) -> float4 {
        const int kb = 16 * c + 4 * g4, x = m0 + 16 * mi + i16;
        if constexpr (FOLD == 1)                              // the loaded row is k = 16c + 4g + e (clamped)
            v = fold_dh4(v, sds[min(16 * c + 4 * g4 + (lane & 3), p.K - 1)], fw[mi]);
        if constexpr (FOLD == 2) v = fold_dh4(v, fds[mi], wk);
        if (MODE == MODE_DW) return fix_xc((XV && !(xdirect && FOLD == 0)) ? quad_transpose(v, lane) : v, x, p.M, kb, p.K, -1);
        return fix_kc(v, x, p.M, kb, p.K);
    };
    auto fix_b = [&](float4 v, int c, int ni) -> float4 {
        const int kb = 16 * c + 4 * g4, x = n0 + 16 * ni + i16;
        if (MODE == MODE_FWD) return fix_kc(v, x, p.N, kb, p.K);
        return fix_xc((XV && !xdirect) ? quad_transpose(v, lane) : v, x, b_cols, kb, p.K, ones_col, OF ? p.ones_from : 0);
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    int nq = (nchunks - w + WAVES - 1) / WAVES;              // chunks w, w+WAVES, ... of this wave
#if GM_SPREAD_TAIL
    const int nfull = nchunks / WAVES, rem = nchunks - nfull * WAVES;
    const bool spread = FOLD == 0 && WAVES == 16 && rem > 0 && rem <= 4 && nfull >= 1;    // kernel uniform
    const bool tail_mine = spread && w < 4 * rem;            // this wave takes k-step (w & 3) of leftover chunk w >> 2
    if (spread) nq = nfull;
#endif
    // G = 1: load a chunk's fragments, consume them, next chunk.  The four waves of a SIMD drift
    // apart and overlap each other's loads and MFMAs; batching G chunks of loads ahead of their
    // MFMAs (the 32x32x2 kernel's scheme) keeps the waves in lockstep -- load phase, then MFMA phase
    // -- and measured slower here (fwd 512x784x400: G=4 8.8 us, G=2 7.7, G=1 7.4-7.6); a rolling
    // prefetch of the next chunk was slower still (iteration 71.7 -> 76.3 us).  G stays a template
    // parameter (= 1) so that kernel names keep their shape across rounds.
#if GM_FAST_INTERIOR
    // workgroup-uniform: every row / column this tile touches exists in both operands (the dW ones column
    // makes the last column tile an edge tile)
    const bool tile_inside = (m0 + 16 * MI <= p.M) && (n0 + 16 * NI <= b_cols);
#endif
    auto consume = [&](const float4 (&ra)[MI], const float4 (&rb)[NI], float4 wk, int q) {
        const int cq = w + q * WAVES;
        float4 fa[MI], fb[NI];
#if GM_FAST_INTERIOR
        if (FOLD == 0 && tile_inside && 16 * cq + 16 <= p.K) {           // wave uniform: nothing to zero
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                fa[mi] = (MODE == MODE_DW && XV && !xdirect) ? quad_transpose(ra[mi], lane) : ra[mi];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                fb[ni] = (MODE != MODE_FWD && XV && !xdirect) ? quad_transpose(rb[ni], lane) : rb[ni];
        } else
#endif
        {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[mi] = fix_a(ra[mi], cq, mi, wk);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fb[ni] = fix_b(rb[ni], cq, ni);
        }
#if GM_EXP_ABLATE == 1                                       // experiment: operands arrive and are fixed up, no MFMA
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
            asm volatile("" ::"v"(fa[mi].x), "v"(fa[mi].y), "v"(fa[mi].z), "v"(fa[mi].w));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
            asm volatile("" ::"v"(fb[ni].x), "v"(fb[ni].y), "v"(fb[ni].z), "v"(fb[ni].w));
        return;
#endif
#if GM_MFMA_INTERLEAVE
#define GM_MFMA_STEP(comp)                                                                         \
        _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                          \
            _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                      \
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mi].comp, fb[ni].comp, acc[mi][ni], 0, 0, 0);
        GM_MFMA_STEP(x) GM_MFMA_STEP(y) GM_MFMA_STEP(z) GM_MFMA_STEP(w)
#undef GM_MFMA_STEP
#else
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                f32x4 c4 = acc[mi][ni];
                c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mi].x, fb[ni].x, c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mi].y, fb[ni].y, c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mi].z, fb[ni].z, c4, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mi].w, fb[ni].w, c4, 0, 0, 0);
                acc[mi][ni] = c4;
            }
#endif
    };
    static_assert(G == 1, "the 16x16x4 kernel runs the per-chunk schedule only");
    auto load_wk = [&](int cc) -> float4 {                    // FOLD == 2: w2 of the chunk's four reduction columns
        if constexpr (FOLD == 2) return *reinterpret_cast<const float4*>(p.fold_w2 + min(16 * cc + 4 * g4, p.K - 4));
        return make_float4(0.f, 0.f, 0.f, 0.f);
    };
    int q_first = 0;
    if constexpr (FOLD != 0) {
        // Folded head: the workgroup's dS rows are rebuilt from the forward's partial dots BEHIND the
        // first chunk's operand loads -- both trips to the fabric are in flight together (the prologue
        // in front of the loop cost a second, serial round trip per launch: in the real step, where
        // every kernel's inputs are fresh from other XCDs, the folded step was only 1.2 us faster than
        // the unfolded one although two launches were gone).  Waves without a chunk still take the barrier.
        float4 ra[MI], rb[NI];
        float4 wk = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool have = nq > 0;
        if (have) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) ra[mi] = load_a(w, mi);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) rb[ni] = load_b(w, ni);
            wk = load_wk(w);
        }
        if constexpr (FOLD == 1) {
            fold_fill_lds(*fold, sds, fold->R);               // every reduction row (ends with the barrier)
        } else {
            if (t < 16 * MI) {                                // the tile's own rows
                float s_, ds_, l_;
                fold_row(*fold, min(m0 + t, fold->R - 1), s_, ds_, l_);
                sds[t] = ds_;
            }
            __syncthreads();
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) fds[mi] = sds[16 * mi + i16];     // dS of this lane's A rows
        }
        if (have) consume(ra, rb, wk, 0);
        q_first = 1;
    }
#if GM_EXP_BATCH_LOADS
    // Experiment: a wave with at most BMAX chunks issues ALL its operand loads back to back (unconditional,
    // clamped chunk index) and then consumes them in order.  Measured inside the real step: 75.1 -> 87.9 us.
    constexpr int BMAX = (MI * NI <= 4) ? 4 : 2;
    if (FOLD == 0 && nq <= BMAX) {                            // wave uniform
        float4 ra[BMAX][MI], rb[BMAX][NI];
#pragma unroll
        for (int q = 0; q < BMAX; ++q) {
            const int cc = w + min(q, max(nq - 1, 0)) * WAVES;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) ra[q][mi] = load_a(min(cc, nchunks - 1), mi);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) rb[q][ni] = load_b(min(cc, nchunks - 1), ni);
        }
#pragma unroll
        for (int q = 0; q < BMAX; ++q)
            if (q < nq) consume(ra[q], rb[q], make_float4(0.f, 0.f, 0.f, 0.f), q);
        q_first = nq;
    }
#endif
#if GM_EXP_PREFETCH > 0
    if (MODE == MODE_DW && FOLD == 0 && nq >= GM_EXP_PREFETCH) {        // wave uniform
        float4 ra[2][MI], rb[2][NI];
        auto issue = [&](int buf, int q) {
            const int cc = min(w + q * WAVES, nchunks - 1);              // past-the-end: a valid, unused reload
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) ra[buf][mi] = load_a(cc, mi);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) rb[buf][ni] = load_b(cc, ni);
        };
        issue(0, 0);
        int q = 0;
        for (; q + 1 < nq; q += 2) {                                      // two chunks per trip: static buffers
            issue(1, q + 1);
            consume(ra[0], rb[0], make_float4(0.f, 0.f, 0.f, 0.f), q);
            issue(0, q + 2);
            consume(ra[1], rb[1], make_float4(0.f, 0.f, 0.f, 0.f), q + 1);
        }
        if (q < nq) consume(ra[0], rb[0], make_float4(0.f, 0.f, 0.f, 0.f), q);
        q_first = nq;
    }
#endif
#if GM_SPREAD_TAIL
    if (spread) {
        // all rounds but the last as usual; the last one also brings in this wave's share of a leftover chunk
        for (int q = 0; q + 1 < nq; ++q) {
            float4 ra[MI], rb[NI];
            const int cc = w + q * WAVES;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) ra[mi] = load_a(cc, mi);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) rb[ni] = load_b(cc, ni);
            consume(ra, rb, make_float4(0.f, 0.f, 0.f, 0.f), q);
        }
        float4 ra[MI], rb[NI], ta[MI], tb[NI];
        const int cc = w + (nq - 1) * WAVES, ct = nfull * WAVES + (w >> 2);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) ra[mi] = load_a(cc, mi);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) rb[ni] = load_b(cc, ni);
        if (tail_mine) {                                      // wave uniform
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) ta[mi] = load_a(ct, mi);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) tb[ni] = load_b(ct, ni);
        }
        consume(ra, rb, make_float4(0.f, 0.f, 0.f, 0.f), nq - 1);
        if (tail_mine) {
            const int j = w & 3;
            auto pick = [&](float4 v) -> float { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); };
            float a1[MI], b1[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a1[mi] = pick(fix_a(ta[mi], ct, mi, make_float4(0.f, 0.f, 0.f, 0.f)));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b1[ni] = pick(fix_b(tb[ni], ct, ni));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[mi], b1[ni], acc[mi][ni], 0, 0, 0);
        }
        q_first = nq;
    }
#endif
    for (int q = q_first; q < nq; ++q) {
        float4 ra[MI], rb[NI];
        const int cc = w + q * WAVES;
#if GM_EXP_ABLATE == 2                                       // experiment: no operand loads, MFMA chain only
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) ra[mi] = make_float4(1.f + lane, 2.f, 3.f + cc, 4.f);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) rb[ni] = make_float4(1.f, 1.f + cc, 1.f, 1.f + lane);
#else
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) ra[mi] = load_a(cc, mi);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) rb[ni] = load_b(cc, ni);
#endif
        const float4 wk = load_wk(cc);
        consume(ra, rb, wk, q);
    }
#if GM_EXP_ABLATE == 3                                       // experiment: no cross-wave reduction / epilogue
    {
        float sink = 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) sink += acc[mi][ni][0] + acc[mi][ni][1] + acc[mi][ni][2] + acc[mi][ni][3];
        if (sink == 123456.789f) red[t] = sink;
        return;
    }
#endif
    if constexpr (MODE == MODE_DW && WAVES == 16 && MI * NI > 4) {
        if (p.vec_epi) {                                     // kernel-argument uniform
            dw_reduce_onepass<MI, NI, false>(p, red, acc, m0, n0, false);
            return;
        }
    }
    // Cross-wave reduction, one 32x32 block of the tile at a time through the same 64 KB buffer.
    // C layout of the 16x16 forms: col = lane & 15, row = (lane >> 4) * 4 + reg.
    // NI odd (32x48 tiles): the last block is 16 columns wide -- its right half of the buffer is stale and the
    // stores are capped at the tile's own columns (the neighbour tile owns the next ones).
    // MI odd > 1 (48x32 tiles): the last block is 16 rows tall.
#pragma unroll
    for (int bm = 0; bm < (MI + 1) / 2; ++bm)
#pragma unroll
        for (int bn = 0; bn < (NI + 1) / 2; ++bn) {
            if (bm + bn > 0) __syncthreads();                // previous block fully consumed
#pragma unroll
            for (int rgi = 0; rgi < 4; ++rgi) {
                const int row = g4 * 4 + rgi;
                red[(w * 32 + row) * 32 + i16] = acc[2 * bm][2 * bn][rgi];
                if (2 * bn + 1 < NI) red[(w * 32 + row) * 32 + 16 + i16] = acc[2 * bm][(2 * bn + 1 < NI) ? 2 * bn + 1 : 0][rgi];
                if (2 * bm + 1 < MI) {
                    constexpr int MIX = MI > 1 ? MI - 1 : 0;             // (keeps the index in range for MI == 1)
                    const int mu = (2 * bm + 1 < MI) ? 2 * bm + 1 : MIX;
                    red[(w * 32 + 16 + row) * 32 + i16] = acc[mu][2 * bn][rgi];
                    if (2 * bn + 1 < NI)
                        red[(w * 32 + 16 + row) * 32 + 16 + i16] = acc[mu][(2 * bn + 1 < NI) ? 2 * bn + 1 : 0][rgi];
                }
            }
            __syncthreads();
            reduce_and_store<MODE, WAVES, (MI > 1 ? 32 : 16)>(p, red, t, m0 + 32 * bm, n0 + 32 * bn,
                                                              (2 * bn + 1 < NI) ? 0x7fffffff : n0 + 32 * bn + 16,
                                                              (2 * bm + 1 < MI) ? 32 : 16);
        }
}

// Interleaved fragments for weight gradients whose reduction has >= this many rows (0: never).  Measured
// (profiles/r03_experiments.md 4c): 2048 rows 27.3 -> 26.3 us, 1024 rows 15.7 -> 15.3, but 512 rows 9.1 -> 9.7 and
// 256 rows 6.3 -> 7.6 (one chunk per wave: the extra load instructions and the predicated reduction fill cost more
// than the transposes they replace).
static inline int dw_il_min_k() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("GM_DW_IL_MIN_K"); v = e ? atoi(e) : 1024; }
    return v;
}
template <int MODE, bool VEC, int WAVES, int G, bool XV, int MI, int NI, int IL = 0>
__global__ __launch_bounds__(WAVES * 64) void gemm16_kernel(GemmP p) {
    __shared__ __attribute__((aligned(16))) float red[(WAVES == 16) ? RedSize<MODE, IL, MI, NI>::value : WAVES * 32 * 32];
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.x16) {
        // workgroup b runs on XCD b % 8 (observed placement; only speed depends on it): XCD (xi, xj) of
        // the xr x xc arrangement owns a pr x pc block of tiles, so its private L2 pulls 1/xr of the
        // A rows and 1/xc of the B rows over the fabric instead of all of both
        const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
        const int pr = (p.tm + p.xr - 1) / p.xr, pc = (p.tn + p.xc - 1) / p.xc;
        by = (xcd / p.xc) * pr + j / pc;
        bx = (xcd % p.xc) * pc + j % pc;
        if (j >= pr * pc || by >= p.tm || bx >= p.tn) return;   // workgroup-uniform
    }
    gemm16_body<MODE, VEC, WAVES, G, XV, MI, NI, false, 0, IL>(p, red, bx, by);
}

// The weight-gradient GEMM with the critic head's backward workgroups riding in the same grid:
// rows [0, hrows) of the grid are head workgroups (dispatched first), the rest are GEMM tiles.  The
// two touch disjoint outputs and neither reads what the other writes (gm_hip.h), so the launch
// boundary -- and its ~2 us of idle machine inside a graph -- between them disappears.
template <int MODE, bool VEC, int G, bool XV, int MI, int NI, bool OF = false, bool FOLDED = false, int IL = 0>
__device__ __forceinline__ void gemm16_with_head(const GemmP& p, const HeadBwdP& hp, int hrows,
                                                 int hblocks) {
    __shared__ __attribute__((aligned(16))) float red[RedSize<MODE, IL, MI, NI>::value];
    __shared__ float sds[FOLDED ? FOLD_MAX_ROWS : 1];
    constexpr int FOLD = FOLDED ? (MODE == MODE_DW ? 1 : 2) : 0;
    // (folded head: the dS prologue runs inside the bodies, behind their first operand loads)
    if ((int)blockIdx.y < hrows) {                           // workgroup-uniform
        const int bid = blockIdx.y * gridDim.x + blockIdx.x;
        if (bid < hblocks) head_bwd_body(hp, bid, sds);
        return;
    }
    gemm16_body<MODE, VEC, 16, G, XV, MI, NI, OF, FOLD, IL>(p, red, blockIdx.x, blockIdx.y - hrows, sds, &hp.fold);
}

template <bool VEC, int G, bool XV, int MI, int NI, bool OF = false, bool FOLDED = false, int IL = 0>
__global__ __launch_bounds__(1024) void gemm16_dw_head_kernel(GemmP p, HeadBwdP hp, int hrows,
                                                              int hblocks) {
    gemm16_with_head<MODE_DW, VEC, G, XV, MI, NI, OF, FOLDED, IL>(p, hp, hrows, hblocks);
}

// The generator step's dX GEMM carrying the one scalar workgroup of the head (loss + tick): the
// generator-mode head_bwd has nothing else to do once head_fwd_loss wrote dH.
template <int G, int MI, int NI, bool FOLDED = false>
__global__ __launch_bounds__(1024) void gemm16_dx_head_kernel(GemmP p, HeadBwdP hp, int hrows,
                                                              int hblocks) {
    gemm16_with_head<MODE_DX, true, G, true, MI, NI, false, FOLDED>(p, hp, hrows, hblocks);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int xcd_mode() {
    static int mode = -1;
    if (mode < 0) {
        const char* e = getenv("GM_XCD_MAP");
        mode = e ? atoi(e) : 0;
    }
    return mode;
}

// The forward GEMM with the batch gather's workgroups riding in the same grid (rows [0, grows) of
// the grid; 16 image rows per workgroup): the gather only needs the index ring and the resident
// dataset, so it costs no launch of its own when the generator's first layer carries it.
template <bool VEC, int G, int MI, int NI>
__global__ __launch_bounds__(1024) void gemm16_fwd_gather_kernel(GemmP p, GatherP gp, int grows,
                                                                 int gblocks) {
    __shared__ __attribute__((aligned(16))) float red[16 * 32 * 32];
    if ((int)blockIdx.y < grows) {                           // workgroup-uniform
        const int bid = blockIdx.y * gridDim.x + blockIdx.x;
        if (bid < gblocks) gather_body(gp, bid);
        return;
    }
    gemm16_body<MODE_FWD, VEC, 16, G, false, MI, NI>(p, red, blockIdx.x, blockIdx.y - grows);
}

// Two weight-gradient GEMMs over the same batch rows (same reduction length, same tile shape) as
// ONE launch: workgroups [0, na) are tiles of the first, the rest tiles of the second.  The
// generator step's dW2 (784x401) and dW1 (400x21) are independent once dH is known.
template <int G, bool XV, int MI, int NI, int IL = 0>
__global__ __launch_bounds__(1024) void gemm16_dw_pair_kernel(GemmP pa, GemmP pb, int na, int tna,
                                                              int tnb) {
    __shared__ __attribute__((aligned(16))) float red[RedSize<MODE_DW, IL, MI, NI>::value];
    const int id = blockIdx.x;
    if (id < na) gemm16_body<MODE_DW, false, 16, G, XV, MI, NI, false, 0, IL>(pa, red, id % tna, id / tna);
    else gemm16_body<MODE_DW, false, 16, G, XV, MI, NI, false, 0, IL>(pb, red, (id - na) % tnb, (id - na) / tnb);
}

// May the epilogue move whole float4s (store4)?  Every array it touches 16-byte aligned, leading dimensions in
// whole float4s.  GM_VEC_EPI=0: element-wise epilogues everywhere (round 3).
template <int MODE>
int vec_epi_ok(const GemmP& p) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("GM_VEC_EPI"); on = e ? atoi(e) : 1; }
    if (!on || !aligned16(p.C) || p.ldc % 4 != 0) return 0;
    if (MODE == MODE_FWD) {
        if (p.bias && !aligned16(p.bias)) return 0;
        if (p.ip_out && !(aligned16(p.ip_out) && aligned16(p.ip_x) && p.ip_ldo % 4 == 0 && p.ip_ldx % 4 == 0)) return 0;
    } else if (MODE == MODE_DX) {
        if (p.epi != GM_ACT_ID && !(aligned16(p.aux) && p.ldaux % 4 == 0)) return 0;
        if (p.add && !(aligned16(p.add) && p.ldadd % 4 == 0)) return 0;
    } else {
        if (p.adam.enabled && !(aligned16(p.adam.pW) && aligned16(p.adam.mW) && aligned16(p.adam.vW))) return 0;
    }
    return 1;
}

// Work that rides in (or pairs with) a GEMM launch.
struct Rider {
    const HeadBwdP* head = nullptr;      // MODE_DW: critic-head backward workgroups
    const GatherP* gather = nullptr;     // MODE_FWD: batch-gather workgroups
    const GemmP* pair = nullptr;         // MODE_DW: a second weight-gradient GEMM
    bool pair_xvec = false;
};

template <int MODE>
int launch(hipStream_t s, const GemmP& p_in, bool vec, bool xvec = false, const Rider& rider = Rider()) {
    const HeadBwdP* head = rider.head;
    // folded critic head: the head workgroups AND the GEMM's A operand depend on the fold -- only the
    // riding 16x16x4 launches below implement it; every other configuration is refused, never
    // silently run without it
    const bool folded = head && head->fold.enabled;
    GemmP p = p_in;
    p.vec_epi = vec_epi_ok<MODE>(p);
    const int tm = (p.M + TM - 1) / TM, tn = (p.N + TN - 1) / TN;
    dim3 grid(tn, tm);
    p.xr = 0;
    // (operands at least one fragment wide, 32-bit element offsets -- else the 16-byte form)
    {
        // weight gradients on 16-byte aligned operands: 2 = chunks by LDS-DMA (gemm16_dw_dma, reductions of >=
        // GM_DW_DMA_MIN_K rows), 1 = interleaved fragments in VGPRs (round 3; GM_DW_DMA_MIN_K=0 brings it back for
        // reductions >= GM_DW_IL_MIN_K)
        // Measured (profiles/r04_experiments.md): 2048 rows 26.0 -> 24.2 us, 1024 rows 15.3 -> 15.2, 768 rows 12.2 -> 11.9;
        // below that the extra hop through LDS costs more than the load instructions it saves (512 rows 9.0 -> 9.2,
        // 256 rows 6.3 -> 7.0): the 16-byte + quad-transpose form keeps the short reductions.
        static int dma_min_k = -1;
        if (dma_min_k < 0) { const char* e = getenv("GM_DW_DMA_MIN_K"); dma_min_k = e ? atoi(e) : 768; }
        const bool fits = MODE == MODE_DW && p.M >= 4 && p.n_real >= 4 &&
                          ((int64_t)p.K + 64) * (p.lda > p.ldb ? p.lda : p.ldb) < (1ll << 31);
        p.il = !fits ? 0 : (dma_min_k > 0 && xvec && p.K >= dma_min_k) ? 2
                         : (dw_il_min_k() > 0 && p.K >= dw_il_min_k()) ? 1 : 0;
    }
    {
        static int blocked = -1;
        if (blocked < 0) { const char* e = getenv("GM_CHUNK_BLOCKED"); blocked = e ? atoi(e) : 0; }
        const int nchunks = (p.K + 7) / 8;
        p.cpw = blocked ? (nchunks + MAXW - 1) / MAXW : 0;
    }
    // GM_WAVES8=1: 8-wave workgroups (two per CU) when there are more tiles than CUs, =2: always.
    // Measured with the 16x16x4 kernel: 16 waves everywhere is fastest (0.1059 vs 0.1086 / 0.1123 ms
    // per iteration), so the default is 0; the 32x32x2 kernel preferred =1.
    static int w8 = -1;
    if (w8 < 0) { const char* e = getenv("GM_WAVES8"); w8 = e ? atoi(e) : 0; }
    const bool use8 = (w8 == 2 || (w8 && (tm * tn > 256))) && p.cpw == 0;   // GM_WAVES8=2: always
    if (xcd_mode() && tm * tn >= 16) {
        // pick the XCD grid xr x xc (xr*xc == 8) minimising per-XCD operand rows tm/xr + tn/xc
        int best = 1 << 30, bxr = 8;
        for (int xr = 1; xr <= 8; xr <<= 1) {
            const int xc = 8 / xr;
            if (xr > tm || xc > tn) continue;
            const int cost = (tm + xr - 1) / xr + (tn + xc - 1) / xc;
            if (cost < best) { best = cost; bxr = xr; }
        }
        if (best < (1 << 30)) {
            p.xr = bxr; p.xc = 8 / bxr; p.tm = tm; p.tn = tn;
            const int per = ((tm + p.xr - 1) / p.xr) * ((tn + p.xc - 1) / p.xc);
            grid = dim3(8 * per, 1);
        }
    }
    {
        // many-row launches: LDS-staged macro tiles.  Riders get their own launch first (a head /
        // gather workgroup set is microseconds next to a >= 1024-row GEMM).
        static int xvq = -1;
        if (xvq < 0) { const char* e = getenv("GM_XVEC"); xvq = e ? atoi(e) : 1; }
        const bool xv_l = xvec && xvq && MODE != MODE_FWD;
        if (!rider.pair && p.xr == 0 && p.cpw == 0 && !folded && !p.hd_part && !p.sq_part) {
            const int cfg = lds_cfg_for<MODE>(p, vec, xv_l);
            if (cfg) {
                if (head) hipLaunchKernelGGL(head_bwd_kernel, dim3(gm_head_bwd_blocks(*head)), dim3(1024), 0, s, *head);
                if (rider.gather)
                    hipLaunchKernelGGL(gather_rows_kernel, dim3(gm_gather_blocks(*rider.gather, 4)), dim3(256), 0, s, *rider.gather);
                return launch_lds<MODE>(s, p, cfg);
            }
        }
    }
    const int nw = use8 ? 8 : 16;
    const int per_wave = ((p.K + 7) / 8 + nw - 1) / nw;      // chunk positions of the busiest wave
    // batch depth: fewest loaded chunks, with a penalty per extra (serialized) batch
    int g = 7, best_cost = 1 << 30;
    for (int cand : {2, 4, 7}) {
        const int batches = (per_wave + cand - 1) / cand;
        const int cost = batches * cand + 2 * (batches - 1);
        if (cost <= best_cost) { best_cost = cost; g = cand; }
    }
    static int xv_on = -1;
    if (xv_on < 0) { const char* e = getenv("GM_XVEC"); xv_on = e ? atoi(e) : 1; }
    const bool xv = xvec && xv_on && MODE != MODE_FWD;
    static int mfma16 = -1;
    if (mfma16 < 0) { const char* e = getenv("GM_MFMA16"); mfma16 = e ? atoi(e) : 1; }
    if (mfma16 && p.xr == 0 && p.cpw == 0) {
        // more 32x32 tiles than CUs: widen the tile along the longer grid axis (one round)
        static int wide_on = -1;
        if (wide_on < 0) { const char* e = getenv("GM_WIDE_TILES"); wide_on = e ? atoi(e) : 1; }
        int wide = 0;                                         // 0: 32x32, 1: 32x64, 2: 64x32
        // measured (profiles/r01_experiments.md): pays only when a wave still has >= 2 chunks of
        // reduction work per tile (dW over 2B = 512 rows: 11.65 -> 10.37 us), loses otherwise
        static int wide_min = -1;
        if (wide_min < 0) { const char* e = getenv("GM_WIDE_MIN_CHUNKS"); wide_min = e ? atoi(e) : 32; }
        if (wide_on && !use8 && tm * tn > 256 && (p.K + 15) / 16 >= wide_min) wide = (tn >= tm) ? 1 : 2;
        // fewer than half as many 32x32 tiles as CUs: 16-row tiles double the workgroups and halve
        // each one's A-fragment loads and MFMA chain (the critic pass over B = 256 rows: 104 tiles)
        static int narrow_on = -1;
        if (narrow_on < 0) { const char* e = getenv("GM_NARROW_TILES"); narrow_on = e ? atoi(e) : 1; }
        static int narrow_max = -1;
        if (narrow_max < 0) { const char* e = getenv("GM_NARROW_MAX_TILES"); narrow_max = e ? atoi(e) : 128; }
        if (narrow_on && !use8 && !wide && tm * tn <= narrow_max && p.M > 16) wide = 3;   // 3: 16x32
        // GM_DW_NI3=1 (experiment): 32x48 tiles where 32x64 would be chosen for a weight gradient (221 instead of
        // 169 workgroups on the 400 x 785 output: 86 % instead of 66 % of the CUs, three quarters of the MFMA chain each)
        static int t48_on = -1;
        if (t48_on < 0) { const char* e = getenv("GM_DW_TILE48"); t48_on = e ? atoi(e) : 2; }
        if (MODE == MODE_DW && t48_on) {
            // 32x48 / 48x32 instead of 32x64 / 64x32 where the narrower tile still fits one round: 221 instead
            // of 169 workgroups on the 400 x 785 / 784 x 401 outputs (86 % instead of 66 % of the CUs, three
            // quarters of the MFMA chain each).  Measured: dW at 512 rows 10.6 -> 9.1 us, 2048 rows 31.5 -> 27.4.
            if (wide == 1 && tm * ((p.N + 47) / 48) <= 256) wide = 4;
            if (wide == 2 && tn * ((p.M + 47) / 48) <= 256) wide = 5;
            // (default 2) also for SHORT reductions (< 32 chunks, where the 64-wide tiles lose): more 32x32 tiles
            // than CUs, but one round of 48-wide / 48-tall ones (the generator's 784 x 401 gradient at B = 256:
            // 325 tiles -> 221; 6.99 -> 6.44 us).  GM_DW_TILE48=1: long reductions only; 0: round 2's shapes
            if (t48_on >= 2 && wide == 0 && !use8 && tm * tn > 256) {
                if (tn >= tm && tm * ((p.N + 47) / 48) <= 256) wide = 4;
                else if (tn < tm && tn * ((p.M + 47) / 48) <= 256) wide = 5;
            }
        }
        // forward over 3B = 768 rows (WGAN-GP / DRAGAN: D's hidden layer on [x_hat ; x ; G(z)] as one launch): 312
        // tiles of 32x32 would be two rounds and 156 of 64x32 leave 40 % of the CUs idle -- 48x32 tiles: 208
        static int f48_on = -1;
        if (f48_on < 0) { const char* e = getenv("GM_FWD_TILE48"); f48_on = e ? atoi(e) : 1; }
        if (MODE == MODE_FWD && f48_on && wide == 2 && !p.hd_part && !p.sq_part && tn * ((p.M + 47) / 48) <= 256) wide = 5;
        if (wide == 1) grid = dim3((p.N + 63) / 64, tm);
        if (wide == 4) grid = dim3((p.N + 47) / 48, tm);
        if (wide == 5) grid = dim3(tn, (p.M + 47) / 48);
        if (wide == 2) grid = dim3(tn, (p.M + 63) / 64);
        if (wide == 3) grid = dim3(tn, (p.M + 15) / 16);
        if constexpr (MODE == MODE_DW) {
            if (head && !use8) {
                const int hblocks = gm_head_bwd_blocks(*head);
                const int hrows = (hblocks + (int)grid.x - 1) / (int)grid.x;
                const dim3 hgrid(grid.x, grid.y + hrows);
#define GM_LH(V, GG, X, OFV, FD) do {                                                              \
        if (wide == 1) hipLaunchKernelGGL((gemm16_dw_head_kernel<V, GG, X, 2, 4, OFV, FD>), hgrid, dim3(1024), 0, s, p, *head, hrows, hblocks); \
        else if (wide == 4 && p.il == 2 && (X)) hipLaunchKernelGGL((gemm16_dw_head_kernel<V, GG, X, 2, 3, OFV, FD, ((X) ? 2 : 0)>), hgrid, dim3(1024), 0, s, p, *head, hrows, hblocks); \
        else if (wide == 5 && p.il == 2 && (X)) hipLaunchKernelGGL((gemm16_dw_head_kernel<V, GG, X, 3, 2, OFV, FD, ((X) ? 2 : 0)>), hgrid, dim3(1024), 0, s, p, *head, hrows, hblocks); \
        else if (wide == 4 && p.il == 1 && !(FD)) hipLaunchKernelGGL((gemm16_dw_head_kernel<V, GG, X, 2, 3, OFV, false, 1>), hgrid, dim3(1024), 0, s, p, *head, hrows, hblocks); \
        else if (wide == 5 && p.il == 1 && !(FD)) hipLaunchKernelGGL((gemm16_dw_head_kernel<V, GG, X, 3, 2, OFV, false, 1>), hgrid, dim3(1024), 0, s, p, *head, hrows, hblocks); \
        else if (wide == 4) hipLaunchKernelGGL((gemm16_dw_head_kernel<V, GG, X, 2, 3, OFV, FD>), hgrid, dim3(1024), 0, s, p, *head, hrows, hblocks); \
        else if (wide == 5) hipLaunchKernelGGL((gemm16_dw_head_kernel<V, GG, X, 3, 2, OFV, FD>), hgrid, dim3(1024), 0, s, p, *head, hrows, hblocks); \
        else if (wide == 2) hipLaunchKernelGGL((gemm16_dw_head_kernel<V, GG, X, 4, 2, OFV, FD>), hgrid, dim3(1024), 0, s, p, *head, hrows, hblocks); \
        else if (wide == 3) hipLaunchKernelGGL((gemm16_dw_head_kernel<V, GG, X, 1, 2, OFV, FD>), hgrid, dim3(1024), 0, s, p, *head, hrows, hblocks); \
        else hipLaunchKernelGGL((gemm16_dw_head_kernel<V, GG, X, 2, 2, OFV, FD>), hgrid, dim3(1024), 0, s, p, *head, hrows, hblocks); } while (0)
#define GM_LH_G(V, X) do { if (p.ones_from > 0) GM_LH(V, 1, X, true, false); else GM_LH(V, 1, X, false, false); } while (0)
                if (folded) {
                    if (!xv || p.ones_from > 0) {
                        gm_set_error("folded head: the weight gradient needs 16-byte aligned operands and no stacked rows");
                        return GM_EINVAL;
                    }
                    GM_LH(false, 1, true, false, true);
                } else if (xv) GM_LH_G(false, true); else GM_LH_G(false, false);   // VEC is a k-contiguous notion
#undef GM_LH_G
#undef GM_LH
                GM_LAUNCH_RET();
            }
        }
        if constexpr (MODE == MODE_DX) {
            if (head && vec && xv && !use8 && (wide == 0 || wide == 3)) {
                const int hblocks = gm_head_bwd_blocks(*head);
                const int hrows = (hblocks + (int)grid.x - 1) / (int)grid.x;
                const dim3 hgrid(grid.x, grid.y + hrows);
#define GM_LXH(GG, FD) do {                                                                        \
        if (wide == 3) hipLaunchKernelGGL((gemm16_dx_head_kernel<GG, 1, 2, FD>), hgrid, dim3(1024), 0, s, p, *head, hrows, hblocks); \
        else hipLaunchKernelGGL((gemm16_dx_head_kernel<GG, 2, 2, FD>), hgrid, dim3(1024), 0, s, p, *head, hrows, hblocks); } while (0)
                if (folded) GM_LXH(1, true); else GM_LXH(1, false);
#undef GM_LXH
                GM_LAUNCH_RET();
            }
        }
        if (folded) {
            gm_set_error("folded head: this launch configuration cannot carry it (needs the riding 16x16x4 "
                         "kernels: 16-byte aligned operands, default tile maps)");
            return GM_EINVAL;
        }
        if (head) {      // this configuration cannot carry the head workgroups: separate launch
            hipLaunchKernelGGL(head_bwd_kernel, dim3(gm_head_bwd_blocks(*head)), dim3(1024), 0, s, *head);
        }
        if constexpr (MODE == MODE_FWD) {
            if (rider.gather) {
                const GatherP& gp = *rider.gather;
                if (vec && !use8 && (wide == 0 || wide == 3)) {
                    const int gblocks = gm_gather_blocks(gp, 16);
                    const int grows = (gblocks + (int)grid.x - 1) / (int)grid.x;
                    const dim3 ggrid(grid.x, grid.y + grows);
#define GM_LG(GG) do {                                                                             \
        if (wide == 3) hipLaunchKernelGGL((gemm16_fwd_gather_kernel<true, GG, 1, 2>), ggrid, dim3(1024), 0, s, p, gp, grows, gblocks); \
        else hipLaunchKernelGGL((gemm16_fwd_gather_kernel<true, GG, 2, 2>), ggrid, dim3(1024), 0, s, p, gp, grows, gblocks); } while (0)
                    GM_LG(1);
#undef GM_LG
                    GM_LAUNCH_RET();
                }
                hipLaunchKernelGGL(gather_rows_kernel, dim3(gm_gather_blocks(gp, 4)), dim3(256), 0, s, gp);
            }
        }
        if constexpr (MODE == MODE_DW) {
            if (rider.pair) {
                GemmP pb = *rider.pair;
                pb.vec_epi = vec_epi_ok<MODE_DW>(pb);
                pb.il = (pb.M >= 4 && pb.n_real >= 4 && (p.il != 2 || rider.pair_xvec) &&
                         ((int64_t)pb.K + 64) * (pb.lda > pb.ldb ? pb.lda : pb.ldb) < (1ll << 31)) ? p.il : 0;
                if (xv && rider.pair_xvec && !use8 && wide != 3 && pb.K == p.K && p.xr == 0) {
                    const int mi = (wide == 2) ? 4 : (wide == 5 ? 3 : 2), ni = (wide == 1) ? 4 : (wide == 4 ? 3 : 2);
                    const int tna = (int)grid.x, na = (int)(grid.x * grid.y);
                    const int tnb = (pb.N + 16 * ni - 1) / (16 * ni), tmb = (pb.M + 16 * mi - 1) / (16 * mi);
                    const dim3 pgrid(na + tnb * tmb);
#define GM_LP(GG) do {                                                                             \
        if (wide == 1) hipLaunchKernelGGL((gemm16_dw_pair_kernel<GG, true, 2, 4>), pgrid, dim3(1024), 0, s, p, pb, na, tna, tnb); \
        else if (wide == 4 && p.il == 2 && pb.il == 2) hipLaunchKernelGGL((gemm16_dw_pair_kernel<GG, true, 2, 3, 2>), pgrid, dim3(1024), 0, s, p, pb, na, tna, tnb); \
        else if (wide == 5 && p.il == 2 && pb.il == 2) hipLaunchKernelGGL((gemm16_dw_pair_kernel<GG, true, 3, 2, 2>), pgrid, dim3(1024), 0, s, p, pb, na, tna, tnb); \
        else if (wide == 4 && p.il == 1 && pb.il == 1) hipLaunchKernelGGL((gemm16_dw_pair_kernel<GG, true, 2, 3, 1>), pgrid, dim3(1024), 0, s, p, pb, na, tna, tnb); \
        else if (wide == 5 && p.il == 1 && pb.il == 1) hipLaunchKernelGGL((gemm16_dw_pair_kernel<GG, true, 3, 2, 1>), pgrid, dim3(1024), 0, s, p, pb, na, tna, tnb); \
        else if (wide == 4) hipLaunchKernelGGL((gemm16_dw_pair_kernel<GG, true, 2, 3>), pgrid, dim3(1024), 0, s, p, pb, na, tna, tnb); \
        else if (wide == 5) hipLaunchKernelGGL((gemm16_dw_pair_kernel<GG, true, 3, 2>), pgrid, dim3(1024), 0, s, p, pb, na, tna, tnb); \
        else if (wide == 2) hipLaunchKernelGGL((gemm16_dw_pair_kernel<GG, true, 4, 2>), pgrid, dim3(1024), 0, s, p, pb, na, tna, tnb); \
        else hipLaunchKernelGGL((gemm16_dw_pair_kernel<GG, true, 2, 2>), pgrid, dim3(1024), 0, s, p, pb, na, tna, tnb); } while (0)
                    GM_LP(1);
#undef GM_LP
                    GM_LAUNCH_RET();
                }
                // not pairable in this configuration: the second GEMM gets its own launch afterwards
                Rider none;
                const int rc = launch<MODE_DW>(s, p_in, vec, xvec, none);
                if (rc) return rc;
                return launch<MODE_DW>(s, pb, false, rider.pair_xvec, none);
            }
        }
        {
            static int x16_on = -1;
            if (x16_on < 0) { const char* e = getenv("GM_XCD16"); x16_on = e ? atoi(e) : 0; }
            const int gtm = (int)grid.y, gtn = (int)grid.x;                  // tiles of the chosen shape
            if (x16_on && gtm * gtn >= 64) {
                const int th = (wide == 2) ? 64 : (wide == 3 ? 16 : (wide == 5 ? 48 : 32)), tw = (wide == 1) ? 64 : (wide == 4 ? 48 : 32);
                int best = 1 << 30, bxr = 0;
                for (int xr = 1; xr <= 8; xr <<= 1) {
                    const int xc = 8 / xr;
                    if (xr > gtm || xc > gtn) continue;
                    const int cost = ((gtm + xr - 1) / xr) * th + ((gtn + xc - 1) / xc) * tw;   // operand rows per XCD
                    if (cost < best) { best = cost; bxr = xr; }
                }
                if (bxr) {
                    p.x16 = 1; p.xr = bxr; p.xc = 8 / bxr; p.tm = gtm; p.tn = gtn;
                    grid = dim3(8 * ((gtm + p.xr - 1) / p.xr) * ((gtn + p.xc - 1) / p.xc), 1);
                }
            }
        }
#define GM_L16(V, W, GG, X) do {                                                                   \
        if (wide == 1) hipLaunchKernelGGL((gemm16_kernel<MODE, V, W, GG, X, 2, 4>), grid, dim3(W * 64), 0, s, p); \
        else if (wide == 4 && MODE == MODE_DW && p.il == 2 && (X) && W == 16) hipLaunchKernelGGL((gemm16_kernel<MODE, V, W, GG, X, 2, 3, ((MODE == MODE_DW && (X) && W == 16) ? 2 : 0)>), grid, dim3(W * 64), 0, s, p); \
        else if (wide == 5 && MODE == MODE_DW && p.il == 2 && (X) && W == 16) hipLaunchKernelGGL((gemm16_kernel<MODE, V, W, GG, X, 3, 2, ((MODE == MODE_DW && (X) && W == 16) ? 2 : 0)>), grid, dim3(W * 64), 0, s, p); \
        else if (wide == 4 && MODE == MODE_DW && p.il == 1 && (X) && W == 16) hipLaunchKernelGGL((gemm16_kernel<MODE, V, W, GG, X, 2, 3, ((MODE == MODE_DW && (X) && W == 16) ? 1 : 0)>), grid, dim3(W * 64), 0, s, p); \
        else if (wide == 5 && MODE == MODE_DW && p.il == 1 && (X) && W == 16) hipLaunchKernelGGL((gemm16_kernel<MODE, V, W, GG, X, 3, 2, ((MODE == MODE_DW && (X) && W == 16) ? 1 : 0)>), grid, dim3(W * 64), 0, s, p); \
        else if (wide == 4) hipLaunchKernelGGL((gemm16_kernel<MODE, V, W, GG, X, 2, 3>), grid, dim3(W * 64), 0, s, p); \
        else if (wide == 5) hipLaunchKernelGGL((gemm16_kernel<MODE, V, W, GG, X, 3, 2>), grid, dim3(W * 64), 0, s, p); \
        else if (wide == 2) hipLaunchKernelGGL((gemm16_kernel<MODE, V, W, GG, X, 4, 2>), grid, dim3(W * 64), 0, s, p); \
        else if (wide == 3) hipLaunchKernelGGL((gemm16_kernel<MODE, V, W, GG, X, 1, 2>), grid, dim3(W * 64), 0, s, p); \
        else hipLaunchKernelGGL((gemm16_kernel<MODE, V, W, GG, X, 2, 2>), grid, dim3(W * 64), 0, s, p); } while (0)
#define GM_L16_G(V, W, X) GM_L16(V, W, 1, X)
#define GM_L16_W(V, X) do { if (use8) GM_L16_G(V, 8, X); else GM_L16_G(V, 16, X); } while (0)
        if (MODE == MODE_FWD) { if (vec) GM_L16_W(true, false); else GM_L16_W(false, false); }
        else if (xv)          { if (vec) GM_L16_W(true, true); else GM_L16_W(false, true); }
        else                  { if (vec) GM_L16_W(true, false); else GM_L16_W(false, false); }
#undef GM_L16_W
#undef GM_L16_G
#undef GM_L16
        GM_LAUNCH_RET();
    }
    if (folded) {
        gm_set_error("folded head: not available with GM_MFMA16=0 / XCD tile maps / blocked chunks");
        return GM_EINVAL;
    }
    if (head) hipLaunchKernelGGL(head_bwd_kernel, dim3(gm_head_bwd_blocks(*head)), dim3(1024), 0, s, *head);
    if (rider.gather)
        hipLaunchKernelGGL(gather_rows_kernel, dim3(gm_gather_blocks(*rider.gather, 4)), dim3(256), 0, s, *rider.gather);
    if (rider.pair) {
        Rider none;
        const int rc = launch<MODE>(s, p_in, vec, xvec, none);
        if (rc) return rc;
        return launch<MODE>(s, *rider.pair, false, rider.pair_xvec, none);
    }
#define GM_LAUNCH(V, W, GG, X) hipLaunchKernelGGL((gemm_kernel<MODE, V, W, GG, X>), grid, dim3(W * 64), 0, s, p)
#define GM_LAUNCH_G(V, W, X) do { if (g == 2) GM_LAUNCH(V, W, 2, X); else if (g == 4) GM_LAUNCH(V, W, 4, X); else GM_LAUNCH(V, W, 7, X); } while (0)
#define GM_LAUNCH_W(V, X) do { if (use8) GM_LAUNCH_G(V, 8, X); else GM_LAUNCH_G(V, 16, X); } while (0)
    if (MODE == MODE_FWD) { if (vec) GM_LAUNCH_W(true, false); else GM_LAUNCH_W(false, false); }
    else if (xv)          { if (vec) GM_LAUNCH_W(true, true); else GM_LAUNCH_W(false, true); }
    else                  { if (vec) GM_LAUNCH_W(true, false); else GM_LAUNCH_W(false, false); }
#undef GM_LAUNCH_W
#undef GM_LAUNCH_G
#undef GM_LAUNCH
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

extern "C" int gm_linear_fwd_interp(void* stream, const float* X, int64_t ldx, gm_slot x_slot,
                                    const float* W, const float* bias, float* Y, int64_t ldy, int M,
                                    int K, int N, int act, const float* eps, gm_slot eps_slot,
                                    const float* x_real, int64_t ld_real, float* x_hat, int64_t ld_hat,
                                    int rows) {
    GM_CHECK_ARG(X && W && Y && M > 0 && K > 0 && N > 0 && ldx >= K && ldy >= N);
    GM_CHECK_ARG(act >= GM_ACT_ID && act <= GM_ACT_SIGMOID);
    GM_CHECK_ARG(eps && x_real && x_hat && rows > 0 && rows <= M && ld_real >= N && ld_hat >= N);
    GemmP p{};
    p.A = X; p.B = W; p.C = Y; p.M = M; p.N = N; p.K = K;
    p.lda = ldx; p.ldb = K; p.ldc = ldy; p.bias = bias; p.epi = act;
    p.a_slot = x_slot; p.b_slot = no_slot();
    p.ip_eps = eps; p.ip_slot = eps_slot; p.ip_x = x_real; p.ip_ldx = ld_real;
    p.ip_out = x_hat; p.ip_ldo = ld_hat; p.ip_rows = rows;
    const bool vec = aligned16(X) && aligned16(W) && (ldx % 4 == 0) && (K % 4 == 0) &&
                     (x_slot.stride % 4 == 0);
    return launch<MODE_FWD>((hipStream_t)stream, p, vec);
}

extern "C" int gm_linear_fwd_headpart(void* stream, const float* X, int64_t ldx, gm_slot x_slot,
                                      const float* W, const float* bias, float* Y, int64_t ldy, int M,
                                      int K, int N, int act, const float* w2, const float* b2,
                                      float* part, int64_t ldp, float* snap) {
    GM_CHECK_ARG(X && W && Y && M > 0 && K > 0 && N > 0 && ldx >= K && ldy >= N);
    GM_CHECK_ARG(act >= GM_ACT_ID && act <= GM_ACT_SIGMOID);
    GM_CHECK_ARG(w2 && b2 && part && snap && ldp >= (N + 31) / 32 && ldp % 4 == 0);
    GM_CHECK_ARG(part != Y && snap != Y && (const float*)part != X && (const float*)snap != X);
    GemmP p{};
    p.A = X; p.B = W; p.C = Y; p.M = M; p.N = N; p.K = K;
    p.lda = ldx; p.ldb = K; p.ldc = ldy; p.bias = bias; p.epi = act;
    p.a_slot = x_slot; p.b_slot = no_slot();
    p.hd_w2 = w2; p.hd_b2 = b2; p.hd_part = part; p.hd_ldp = ldp; p.hd_snap = snap;
    const bool vec = aligned16(X) && aligned16(W) && (ldx % 4 == 0) && (K % 4 == 0) &&
                     (x_slot.stride % 4 == 0);
    return launch<MODE_FWD>((hipStream_t)stream, p, vec);
}

extern "C" int gm_linear_fwd_sqerr(void* stream, const float* X, int64_t ldx, const float* W,
                                  const float* bias, float* Y, int64_t ldy, int M, int K, int N,
                                  const float* target, int64_t ld_target, float* dA, int64_t lda,
                                  float* part, int64_t ldp) {
    GM_CHECK_ARG(X && W && Y && M > 0 && K > 0 && N > 0 && ldx >= K && ldy >= N);
    GM_CHECK_ARG(target && dA && part && ld_target >= N && lda >= N && ldp >= (N + 31) / 32);
    GM_CHECK_ARG(dA != Y && part != Y && part != dA && (const float*)dA != X && (const float*)part != X &&
                 (const float*)Y != target && (const float*)dA != target);
    GemmP p{};
    p.A = X; p.B = W; p.C = Y; p.M = M; p.N = N; p.K = K;
    p.lda = ldx; p.ldb = K; p.ldc = ldy; p.bias = bias; p.epi = GM_ACT_SIGMOID;
    p.a_slot = no_slot(); p.b_slot = no_slot();
    p.sq_x = target; p.sq_ldx = ld_target; p.sq_dA = dA; p.sq_lda = lda; p.sq_part = part; p.sq_ldp = ldp;
    const bool vec = aligned16(X) && aligned16(W) && (ldx % 4 == 0) && (K % 4 == 0);
    return launch<MODE_FWD>((hipStream_t)stream, p, vec);
}

static int fwd_gather_impl(void* stream, const float* X, int64_t ldx, gm_slot x_slot,
                           const float* W, const float* bias, float* Y, int64_t ldy, int M,
                           int K, int N, int act, const GatherP& g, float* out);

extern "C" int gm_linear_fwd_gather(void* stream, const float* X, int64_t ldx, gm_slot x_slot,
                                    const float* W, const float* bias, float* Y, int64_t ldy, int M,
                                    int K, int N, int act, const float* data, int64_t n_rows,
                                    const int64_t* idx, gm_slot idx_slot, float* out,
                                    int64_t ld_out, int B, int row_elems) {
    GatherP g{};
    const int rc = gm_gather_fill(data, n_rows, idx, idx_slot, out, ld_out, B, row_elems, &g);
    if (rc) return rc;
    return fwd_gather_impl(stream, X, ldx, x_slot, W, bias, Y, ldy, M, K, N, act, g, out);
}

extern "C" int gm_linear_fwd_gather_bits(void* stream, const float* X, int64_t ldx, gm_slot x_slot,
                                         const float* W, const float* bias, float* Y, int64_t ldy, int M,
                                         int K, int N, int act, const uint32_t* bits, int words_per_row,
                                         int64_t n_rows, const int64_t* idx, gm_slot idx_slot, float* out,
                                         int64_t ld_out, int B, int row_elems) {
    GatherP g{};
    const int rc = gm_gather_fill_bits(bits, words_per_row, n_rows, idx, idx_slot, out, ld_out, B, row_elems, &g);
    if (rc) return rc;
    return fwd_gather_impl(stream, X, ldx, x_slot, W, bias, Y, ldy, M, K, N, act, g, out);
}

static int fwd_gather_impl(void* stream, const float* X, int64_t ldx, gm_slot x_slot,
                           const float* W, const float* bias, float* Y, int64_t ldy, int M,
                           int K, int N, int act, const GatherP& g, float* out) {
    GM_CHECK_ARG(X && W && Y && M > 0 && K > 0 && N > 0 && ldx >= K && ldy >= N);
    GM_CHECK_ARG(act >= GM_ACT_ID && act <= GM_ACT_SIGMOID);
    // the gathered rows must not be an operand or the output of this GEMM
    GM_CHECK_ARG(out != Y && out != X);
    GemmP p{};
    p.A = X; p.B = W; p.C = Y; p.M = M; p.N = N; p.K = K;
    p.lda = ldx; p.ldb = K; p.ldc = ldy; p.bias = bias; p.epi = act;
    p.a_slot = x_slot; p.b_slot = no_slot();
    const bool vec = aligned16(X) && aligned16(W) && (ldx % 4 == 0) && (K % 4 == 0) &&
                     (x_slot.stride % 4 == 0);
    Rider r;
    r.gather = &g;
    return launch<MODE_FWD>((hipStream_t)stream, p, vec, false, r);
}

static int dx_impl(void* stream, const float* dA, int64_t lda, const float* W, float* dX, int64_t ldx,
                   const float* below, int64_t ld_below, int M, int K, int N, int epi,
                   const float* add, int64_t ldadd, float add_scale, const HeadBwdP* head = nullptr,
                   const float* fold_w2 = nullptr);

extern "C" int gm_linear_bwd_dx_head(void* stream, const float* dA, int64_t lda, const float* W,
                                     float* dX, int64_t ldx, const float* below, int64_t ld_below,
                                     int M, int K, int N, int epi, const gm_head_bwd_args* head) {
    GM_CHECK_ARG(head);
    // the head workgroups only read dS / rowloss / H: none of them may be this GEMM's output
    GM_CHECK_ARG((const float*)dX != head->H && (const float*)dX != head->dS &&
                 (const float*)dX != head->rowloss && dX != head->dH);
    HeadBwdP hp{};
    const int rc = gm_head_from_args(*head, &hp);
    if (rc) return rc;
    return dx_impl(stream, dA, lda, W, dX, ldx, below, ld_below, M, K, N, epi, nullptr, 0, 0.f, &hp);
}

extern "C" int gm_linear_bwd_dx_reparam(void* stream, const float* dA, int64_t lda, const float* W,
                                        float* dZ, int64_t ldz, int M, int Z, int N, const float* ml,
                                        int64_t ldml, const float* eps, gm_slot eps_slot, float* dml,
                                        int64_t ldd) {
    GM_CHECK_ARG(dA && W && dZ && M > 0 && Z > 0 && N > 0 && lda >= N && ldz >= Z);
    GM_CHECK_ARG(ml && eps && dml && ldml >= 2 * Z && ldd >= 2 * Z && dml != dZ && (const float*)dml != ml &&
                 (const float*)dml != dA);
    GemmP p{};
    p.A = dA; p.B = W; p.C = dZ; p.M = M; p.N = Z; p.K = N;
    p.lda = lda; p.ldb = Z; p.ldc = ldz; p.epi = GM_ACT_ID;
    p.a_slot = no_slot(); p.b_slot = no_slot();
    p.rp_ml = ml; p.rp_ldml = ldml; p.rp_eps = eps; p.rp_slot = eps_slot; p.rp_dml = dml; p.rp_ldd = ldd;
    p.rp_Z = Z;
    const bool vec = aligned16(dA) && (lda % 4 == 0) && (N % 4 == 0);
    const bool xvec = aligned16(W) && (Z % 4 == 0);
    return launch<MODE_DX>((hipStream_t)stream, p, vec, xvec);
}

extern "C" int gm_linear_bwd_dx_head_fold(void* stream, const float* H, int64_t ldh, const float* W,
                                          float* dX, int64_t ldx, const float* below, int64_t ld_below,
                                          int M, int K, int N, int epi, const gm_head_bwd_args* head,
                                          const gm_head_fold_args* fold) {
    GM_CHECK_ARG(head && fold && head->gen_mode && head->H == H && head->B == M && head->Hd == N);
    GM_CHECK_ARG((const float*)dX != H && dX != fold->S && dX != fold->dS && dX != fold->rowloss);
    HeadBwdP hp{};
    const int rc = gm_head_from_args(*head, &hp, fold);
    if (rc) return rc;
    return dx_impl(stream, H, ldh, W, dX, ldx, below, ld_below, M, K, N, epi, nullptr, 0, 0.f, &hp, fold->snap);
}

extern "C" int gm_linear_bwd_dx(void* stream, const float* dA, int64_t lda, const float* W,
                                float* dX, int64_t ldx, const float* below, int64_t ld_below,
                                int M, int K, int N, int epi) {
    return dx_impl(stream, dA, lda, W, dX, ldx, below, ld_below, M, K, N, epi, nullptr, 0, 0.f);
}

extern "C" int gm_linear_bwd_dx_add(void* stream, const float* dA, int64_t lda, const float* W,
                                    float* dX, int64_t ldx, const float* below, int64_t ld_below,
                                    int M, int K, int N, int epi, const float* add, int64_t ldadd,
                                    float add_scale) {
    GM_CHECK_ARG(add && ldadd >= K);
    return dx_impl(stream, dA, lda, W, dX, ldx, below, ld_below, M, K, N, epi, add, ldadd, add_scale);
}

static int dx_impl(void* stream, const float* dA, int64_t lda, const float* W, float* dX, int64_t ldx,
                   const float* below, int64_t ld_below, int M, int K, int N, int epi,
                   const float* add, int64_t ldadd, float add_scale, const HeadBwdP* head,
                   const float* fold_w2) {
    GM_CHECK_ARG(dA && W && dX && M > 0 && K > 0 && N > 0 && lda >= N && ldx >= K);
    GM_CHECK_ARG(epi == GM_ACT_ID || (below && ld_below >= K));
    GemmP p{};
    // C[M, K_layer] = sum_{n} dA[m,n] * W[n,k]  => GEMM dims (M, N=K_layer, K=N_layer)
    p.A = dA; p.B = W; p.C = dX; p.M = M; p.N = K; p.K = N;
    p.lda = lda; p.ldb = K; p.ldc = ldx; p.aux = below; p.ldaux = ld_below; p.epi = epi;
    p.add = add; p.ldadd = ldadd; p.add_scale = add_scale;
    p.fold_w2 = fold_w2;
    p.a_slot = no_slot(); p.b_slot = no_slot();
    const bool vec = aligned16(dA) && (lda % 4 == 0) && (N % 4 == 0) && (!fold_w2 || aligned16(fold_w2));
    const bool xvec = aligned16(W) && (K % 4 == 0);
    Rider r;
    r.head = head;
    return launch<MODE_DX>((hipStream_t)stream, p, vec, xvec, r);
}

static int dw_impl(void* stream, const float* dA, int64_t lda, const float* X, int64_t ldx,
                   gm_slot x_slot, float* dW, float* db, int M, int K, int N, int accumulate,
                   const gm_adam_epi* adam, const HeadBwdP* head = nullptr, int ones_from = 0,
                   const float* fold_w2 = nullptr);
static int dw_fill(const float* dA, int64_t lda, const float* X, int64_t ldx, gm_slot x_slot,
                   float* dW, float* db, int M, int K, int N, int accumulate,
                   const gm_adam_epi* adam, GemmP* out, bool* xvec);

extern "C" int gm_linear_bwd_dw(void* stream, const float* dA, int64_t lda, const float* X,
                                int64_t ldx, gm_slot x_slot, float* dW, float* db, int M, int K,
                                int N, int accumulate) {
    return dw_impl(stream, dA, lda, X, ldx, x_slot, dW, db, M, K, N, accumulate, nullptr);
}

extern "C" int gm_linear_bwd_dw_adam(void* stream, const float* dA, int64_t lda, const float* X,
                                     int64_t ldx, gm_slot x_slot, float* dW, float* db, int M,
                                     int K, int N, float* pW, float* mW, float* vW, float* pb,
                                     float* mb, float* vb, const float* sched, gm_slot sched_slot,
                                     double beta1, double beta2, double eps, double weight_decay,
                                     float clamp) {
    GM_CHECK_ARG(db && pW && mW && vW && pb && mb && vb && sched);
    gm_adam_epi a{};
    a.pW = pW; a.mW = mW; a.vW = vW; a.pb = pb; a.mb = mb; a.vb = vb; a.sched = sched;
    a.sched_slot = sched_slot; a.omb1 = (float)(1.0 - beta1); a.b2 = (float)beta2;
    a.omb2 = (float)(1.0 - beta2); a.eps = (float)eps; a.wd = (float)weight_decay; a.clamp = clamp;
    a.enabled = 1;
    return dw_impl(stream, dA, lda, X, ldx, x_slot, dW, db, M, K, N, 0, &a);
}

extern "C" int gm_linear_bwd_dw_adam_head(void* stream, const float* dA, int64_t lda,
                                          const float* X, int64_t ldx, gm_slot x_slot, float* dW,
                                          float* db, int M, int K, int N, float* pW, float* mW,
                                          float* vW, float* pb, float* mb, float* vb,
                                          const float* sched, gm_slot sched_slot, double beta1,
                                          double beta2, double eps, double weight_decay, float clamp,
                                          const gm_head_bwd_args* head) {
    return gm_linear_bwd_dw_adam_head_ex(stream, dA, lda, X, ldx, x_slot, dW, db, M, K, N, pW, mW, vW, pb, mb,
                                         vb, sched, sched_slot, beta1, beta2, eps, weight_decay, clamp, head, 0);
}

extern "C" int gm_linear_bwd_dw_adam_head_ex(void* stream, const float* dA, int64_t lda,
                                             const float* X, int64_t ldx, gm_slot x_slot, float* dW,
                                             float* db, int M, int K, int N, float* pW, float* mW,
                                             float* vW, float* pb, float* mb, float* vb,
                                             const float* sched, gm_slot sched_slot, double beta1,
                                             double beta2, double eps, double weight_decay, float clamp,
                                             const gm_head_bwd_args* head, int ones_from) {
    GM_CHECK_ARG(head);
    // sched == NULL: plain gradients (data-parallel runs all-reduce before the optimizer step)
    GM_CHECK_ARG(!sched || (db && pW && mW && vW && pb && mb && vb));
    // the head may update (w2, b2) and writes gw2/gb2/loss: none of it may alias the GEMM's operands
    GM_CHECK_ARG((const float*)head->w2 != pW || !pW);
    GM_CHECK_ARG(head->gw2 != dW);
    HeadBwdP hp{};
    const int rc = gm_head_from_args(*head, &hp);
    if (rc) return rc;
    gm_adam_epi a{};
    if (sched) {
        a.pW = pW; a.mW = mW; a.vW = vW; a.pb = pb; a.mb = mb; a.vb = vb; a.sched = sched;
        a.sched_slot = sched_slot; a.omb1 = (float)(1.0 - beta1); a.b2 = (float)beta2;
        a.omb2 = (float)(1.0 - beta2); a.eps = (float)eps; a.wd = (float)weight_decay; a.clamp = clamp;
        a.enabled = 1;
    }
    return dw_impl(stream, dA, lda, X, ldx, x_slot, dW, db, M, K, N, 0, sched ? &a : nullptr, &hp, ones_from);
}

extern "C" int gm_linear_bwd_dw_adam_head_fold(void* stream, const float* H, int64_t ldh,
                                               const float* X, int64_t ldx, gm_slot x_slot, float* dW,
                                               float* db, int M, int K, int N, float* pW, float* mW,
                                               float* vW, float* pb, float* mb, float* vb,
                                               const float* sched, gm_slot sched_slot, double beta1,
                                               double beta2, double eps, double weight_decay, float clamp,
                                               const gm_head_bwd_args* head, const gm_head_fold_args* fold) {
    GM_CHECK_ARG(head && fold && !head->gen_mode && head->H == H && 2 * head->B == M && head->Hd == N);
    GM_CHECK_ARG(!sched || (db && pW && mW && vW && pb && mb && vb));
    GM_CHECK_ARG((const float*)head->w2 != pW || !pW);
    GM_CHECK_ARG(head->gw2 != dW && (const float*)dW != H);
    // the GEMM workgroups read the SNAPSHOT of (w2, b2): the head workgroups may step the parameters
    GM_CHECK_ARG(fold->snap != (const float*)head->w2 && fold->snap != (const float*)head->b2);
    HeadBwdP hp{};
    const int rc = gm_head_from_args(*head, &hp, fold);
    if (rc) return rc;
    gm_adam_epi a{};
    if (sched) {
        a.pW = pW; a.mW = mW; a.vW = vW; a.pb = pb; a.mb = mb; a.vb = vb; a.sched = sched;
        a.sched_slot = sched_slot; a.omb1 = (float)(1.0 - beta1); a.b2 = (float)beta2;
        a.omb2 = (float)(1.0 - beta2); a.eps = (float)eps; a.wd = (float)weight_decay; a.clamp = clamp;
        a.enabled = 1;
    }
    return dw_impl(stream, H, ldh, X, ldx, x_slot, dW, db, M, K, N, 0, sched ? &a : nullptr, &hp, 0, fold->snap);
}

static int dw_adam_fill(const gm_dw_adam_args& a, GemmP* p, bool* xvec) {
    if (!a.sched)            // plain gradient (no optimizer step in the epilogue)
        return dw_fill(a.dA, a.lda, a.X, a.ldx, a.x_slot, a.dW, a.db, a.M, a.K, a.N, 0, nullptr, p, xvec);
    GM_CHECK_ARG(a.db && a.pW && a.mW && a.vW && a.pb && a.mb && a.vb);
    gm_adam_epi e{};
    e.pW = a.pW; e.mW = a.mW; e.vW = a.vW; e.pb = a.pb; e.mb = a.mb; e.vb = a.vb; e.sched = a.sched;
    e.sched_slot = a.sched_slot; e.omb1 = (float)(1.0 - a.beta1); e.b2 = (float)a.beta2;
    e.omb2 = (float)(1.0 - a.beta2); e.eps = (float)a.eps; e.wd = (float)a.weight_decay;
    e.clamp = a.clamp; e.enabled = 1;
    return dw_fill(a.dA, a.lda, a.X, a.ldx, a.x_slot, a.dW, a.db, a.M, a.K, a.N, 0, &e, p, xvec);
}

extern "C" int gm_linear_bwd_dw_adam_pair(void* stream, const gm_dw_adam_args* first,
                                          const gm_dw_adam_args* second) {
    GM_CHECK_ARG(first && second);
    // neither may consume what the other produces or updates
    GM_CHECK_ARG(first->dW != second->dW && (first->pW != second->pW || !first->pW));
    GM_CHECK_ARG(!first->pW || ((const float*)first->pW != second->dA && (const float*)first->pW != second->X));
    GM_CHECK_ARG(!second->pW || ((const float*)second->pW != first->dA && (const float*)second->pW != first->X));
    GM_CHECK_ARG(first->dW != second->dA && first->dW != second->X && second->dW != first->dA && second->dW != first->X);
    GemmP pa{}, pb{};
    bool xa = false, xb = false;
    int rc = dw_adam_fill(*first, &pa, &xa);
    if (rc) return rc;
    rc = dw_adam_fill(*second, &pb, &xb);
    if (rc) return rc;
    Rider r;
    r.pair = &pb;
    r.pair_xvec = xb;
    return launch<MODE_DW>((hipStream_t)stream, pa, false, xa, r);
}

static int dw_impl(void* stream, const float* dA, int64_t lda, const float* X, int64_t ldx,
                   gm_slot x_slot, float* dW, float* db, int M, int K, int N, int accumulate,
                   const gm_adam_epi* adam, const HeadBwdP* head, int ones_from, const float* fold_w2) {
    GemmP p{};
    bool xvec = false;
    const int rc = dw_fill(dA, lda, X, ldx, x_slot, dW, db, M, K, N, accumulate, adam, &p, &xvec);
    if (rc) return rc;
    GM_CHECK_ARG(ones_from >= 0 && ones_from <= M && (ones_from == 0 || head));
    p.ones_from = ones_from;
    p.fold_w2 = fold_w2;
    if (fold_w2 && !aligned16(fold_w2)) xvec = false;
    Rider r;
    r.head = head;
    return launch<MODE_DW>((hipStream_t)stream, p, false, xvec, r);
}

static int dw_fill(const float* dA, int64_t lda, const float* X, int64_t ldx, gm_slot x_slot,
                   float* dW, float* db, int M, int K, int N, int accumulate,
                   const gm_adam_epi* adam, GemmP* out, bool* xvec_out) {
    GM_CHECK_ARG(dA && X && dW && M > 0 && K > 0 && N > 0 && lda >= N && ldx >= K);
    GemmP p{};
    if (adam) p.adam = *adam;
    // C[N_layer, K_layer(+1)] = sum_{m} dA[m,n] * X[m,k]  => GEMM dims (M=N_layer, N=K_layer(+1), K=batch)
    p.A = dA; p.B = X; p.C = dW; p.M = N; p.N = K + (db ? 1 : 0); p.K = M;
    p.lda = lda; p.ldb = ldx; p.ldc = K; p.db = db; p.n_real = K; p.accumulate = accumulate;
    p.a_slot = no_slot(); p.b_slot = x_slot;
    // both operands are x-contiguous; the 16-byte + quad-transpose path needs every row start and
    // the tile edges on 4-element boundaries (the virtual ones-column sits at x == K, K % 4 == 0)
    const bool xvec = aligned16(dA) && aligned16(X) && (lda % 4 == 0) && (ldx % 4 == 0) &&
                      (N % 4 == 0) && (K % 4 == 0) && (x_slot.stride % 4 == 0) && N >= 4 && K >= 4;
    *out = p;
    *xvec_out = xvec;
    return 0;
}
