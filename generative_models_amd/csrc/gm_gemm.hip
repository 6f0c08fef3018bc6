// gm_gemm.hip -- fp32 MFMA GEMMs for the 784<->400<->20 MLP layers (forward, dX, dW) on gfx950.
//
// One kernel body, three operand layouts:
//   fwd  (NT): Y[M,N]  = act(X[M,K] * W[N,K]^T + b)         ns_gan.py:44-45,58-59
//   dx   (NN): dX[M,K] = dA[M,N] * W[N,K] (* act'(below))   autograd of the above, ns_gan.py:138,155
//   dw   (TN): dW[N,K] = dA[M,N]^T * X[M,K], db = colsum(dA)
//
// Design for this problem (B=256: every GEMM is ~0.1-0.3 GFLOP, L2/MALL-resident, and the step is a chain of 8
// dependent launches -- latency and the per-CU memory pipeline, not FLOPs, are the enemy):
//   * split-reduction kernels (gemm16_*): one 32x32 output tile per 1024-thread workgroup (16x32, 32x48 / 48x32,
//     32x64 / 64x32 chosen by tile count: pick_tile) so that even B=256 launches 200-225 workgroups; the reduction
//     dimension is split ACROSS THE 16 WAVES of the workgroup in round-robin 16-deep chunks, no two waves touch the
//     same k, so operands are NOT shared through LDS: each wave loads its chunks straight into
//     v_mfma_f32_16x16x4_f32 fragment registers, with no barrier inside the reduction loop;
//   * every operand load is 16 bytes per lane (what an instruction costs the vector cache depends on which lanes read
//     adjacent bytes: tools/fill_probe, profiles/r05_fill_law.md -- round 4's "40 cycles whatever it carries" was
//     the uncoalesced fragment pattern).  k-contiguous operands: lane (i = lane & 15, g = lane >> 4) loads the 4
//     consecutive k = 16c + 4g .. of row i and MFMA j consumes element j (an identical k-permutation for A and B:
//     legal because a sum does not care); x-contiguous operands: one 16-byte load of 4 consecutive x at one k, a quad
//     reading 64 contiguous bytes of one k-row (round 5: coalesced), plus a 4x4 transpose over lanes l, l^4, l^8, l^12
//     (DPP, no LDS) and a fixed output permutation undone in the epilogue; weight gradients over >= 768 rows:
//     whole chunks by LDS-DMA into wave-private buffers and interleaved fragments (gemm16_dw_dma);
//   * loads are branch-free (clamped addresses, zeroing selects deferred to the consume stage);
//   * the partial tiles are combined through LDS in wave order, then bias / activation / activation-gradient /
//     accumulate / Adam epilogues are applied (weight-gradient tiles of several blocks in one pass: dw_reduce_onepass);
//   * many-row forward / input-gradient launches (M >= 1024): LDS-staged 64x64 / 32x64 macro tiles (gemm_lds_kernel);
//   * exact fp32 (MFMA f32 == fmaf chain); deterministic (no atomics: graph replay == eager launch bitwise);
//   * db falls out of the dW GEMM for free: X gets a virtual ones-column at index K;
//   * optimizer-in-epilogue: the thread that produces a gradient element also applies Adam to the parameter;
//   * independent work rides in the same grid (critic-head workgroups in the layer-1 weight gradient / input
//     gradient, the batch gather in the generator's first forward, a second weight gradient as a pair).
// What was tried and measured slower is recorded in profiles/r01 .. r04_experiments.md, not kept as switches here.
#include "gm_common.h"
#include "gm_head.h"
#include "gm_gather.h"
#include "gm_ldsdma.h"

#include <cstdlib>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------
// Per-wave timeline probe (tools/wave_timeline.py -> profiles/r06_*_wave_timeline.md).  ONLY in a build of the library
// with -DGM_STAMPS (libgm_hip_stamps.so, never the shipped one: GM_STAMP() compiles to nothing otherwise).  Every GEMM
// launch gets a slot of a device buffer (the host numbers launches in issue order, so the nodes of a captured graph
// keep theirs across replays); thread 0 of EVERY workgroup folds its entry / exit time into the slot header (min / max
// of the 100 MHz wall clock: launch-to-launch gaps), and lane 0 of every wave of ONE workgroup (the probe tile) records
// (shader cycle counter, wall clock) at the marked points of the kernel body.
// ------------------------------------------------------------------------------------------
#ifdef GM_STAMPS
namespace gm_stamps {
constexpr int WAVES = 16, IDS = 40, HDR = 8, MAXB = 512, SLOT_WORDS = HDR + 2 * MAXB + WAVES * IDS * 2, SLOTS = 64;
// host side: where the slots live and which tile stamps; they travel to the kernels INSIDE GemmP (scalar registers: a
// first version read two device globals per stamp -- a memory round trip in every wave of every workgroup -- and folded
// the launch's entry / exit times with atomics on one address: the step under the probe took 127 us instead of 68)
static unsigned long long* host_buf = nullptr;
static int host_tile = 0, host_next = 0;
struct Ctx { unsigned long long* slot; int tile; };
__device__ __forceinline__ void rec(const Ctx& c, int tile, int id) {
    if (!c.slot || tile != c.tile || (threadIdx.x & 63)) return;
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long* w = c.slot + HDR + 2 * MAXB + ((threadIdx.x >> 6) * IDS + id) * 2;
    w[0] = clock64(); w[1] = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
}
// every workgroup: its own entry / exit word (no atomics, nothing waits for the store)
__device__ __forceinline__ void edge(const Ctx& c, bool exit_, int tile, int M, int N, int K, int mode) {
    if (!c.slot || threadIdx.x) return;
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    if (blk < MAXB) c.slot[HDR + 2 * blk + (exit_ ? 1 : 0)] = wall_clock64();
    if (!exit_ && tile == 0) {                                // (a rider launch's workgroup 0 is not a GEMM tile)
        c.slot[0] = 1; c.slot[2] = (unsigned long long)gridDim.x * gridDim.y; c.slot[3] = M; c.slot[4] = N; c.slot[5] = K;
        c.slot[6] = mode; c.slot[7] = blockDim.x;
    }
}
}  // namespace gm_stamps
#define GM_STAMP(ctx, tile, id) gm_stamps::rec((ctx), (tile), (id))
#define GM_STAMP_EDGE(p, exit_, tile, mode) gm_stamps::edge((p).stamp, (exit_), (tile), (p).M, (p).N, (p).K, (mode))
// keeps the stamp behind the instructions that produce `v` (the wait for an operand, the last MFMA of a chunk)
#define GM_STAMP_AFTER(v) asm volatile("" ::"v"(v))
extern "C" int gm_stamps_set(unsigned long long* dev_buf, int probe_tile) {
    gm_stamps::host_buf = dev_buf; gm_stamps::host_tile = probe_tile; gm_stamps::host_next = 0;
    return 0;
}
extern "C" int gm_stamps_layout(int* out6) {
    out6[0] = gm_stamps::WAVES; out6[1] = gm_stamps::IDS; out6[2] = gm_stamps::HDR; out6[3] = gm_stamps::SLOT_WORDS;
    out6[4] = gm_stamps::SLOTS; out6[5] = gm_stamps::MAXB;
    return 0;
}
#else
#define GM_STAMP(slot, tile, id) do {} while (0)
#define GM_STAMP_EDGE(p, exit_, tile, mode) do {} while (0)
#define GM_STAMP_AFTER(v) do {} while (0)
#endif

namespace {

constexpr int TM = 32, TN = 32;

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
    int tn;                   // LDS macro-tile kernel: n-tiles
    int lds_tm, lds_mpx;      // LDS macro-tile kernel: m-tiles in total / per XCD (XCD x owns m-tiles [x * mpx, (x+1) * mpx))
    int dma;                  // dw: operand chunks by LDS-DMA (gemm16_dw_dma) instead of 16-byte loads + quad transposes
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
    // bit-packed operand rows (PK kernels): rows [0, pk_rows) of the batch-row operand -- A in the
    // forward, B = X in the weight gradient -- are read from pk_bits (GatherP's layout: element e of row r = bit e & 31 of
    // word r * pk_wpr + (e >> 5)) and expanded to 0.0f / 1.0f in registers; the fp32 rows behind them are never touched.
    // pk_rows % 32 == 0: a forward tile / a 16-row reduction chunk is packed or fp32 as a whole, and each kind gets its own
    // branch-free loop (selecting per fragment inside one loop made hipcc serialise the loads behind s_waitcnt vmcnt(0)).
    const uint32_t* pk_bits; int pk_wpr; int pk_rows;
#ifdef GM_STAMPS
    gm_stamps::Ctx stamp;     // timeline probe build only
#endif
};

// Operand base behind an optional ring slot.  Round 6: almost every launch passes NO_SLOT for both operands (only the
// generator's first layer reads its input -- the noise ring -- through one), and the general resolution -- the
// counter's load behind the argument loads (a third serial scalar-memory trip), a 64-bit multiply-add, the modulo's
// fast-path test, a 64-bit multiply -- sat in front of every kernel's FIRST operand loads: ~40 scalar instructions that
// each of a SIMD's four waves issues in turn (16 cycles apiece by the time the fourth wave is through).  Compiled out,
// the forward launches run 0.33 - 0.40 us shorter (8.07 -> 7.70 us; a run-time test for "no slot" did not keep the gain:
// hipcc scheduled one shape 0.7 us slower around the branch -- profiles/r06_experiments.md section 5).  SL is a template
// argument of the kernels: the host picks the slot-free instantiation whenever both operands come without one.
template <bool SL>
__device__ __forceinline__ const float* slot_base(const float* P, const gm_slot& s) {
    if constexpr (SL) return P + gm_slot_offset(s);
    return P;
}
inline bool has_slot(const gm_slot& s) { return s.ctr != nullptr || s.add != 0; }

// elements e .. e+3 (e % 4 == 0) of a packed row from the word that holds them; expanded where the fragment is consumed
__device__ __forceinline__ float4 pk_expand(uint32_t word, int e) {
    const uint32_t v = word >> (e & 31);
    return make_float4((float)(v & 1u), (float)((v >> 1) & 1u), (float)((v >> 2) & 1u), (float)((v >> 3) & 1u));
}

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

// Index of element (row, col) of wave w's 32 x 32 partial tile in the cross-wave reduction buffer.  Round 6: the column
// is swizzled by one bit of the row -- col ^ 16 where ((row >> 2) ^ row) is odd.  The MFMA's C layout puts lanes 0..15
// and 16..31 of a half-wave on rows that are 4 apart (1 apart with the x-contiguous operands' output permutation): 128
// (32) floats = the same LDS bank, so every ds_write of a partial tile was a 2-way bank conflict with half of the banks
// idle (SQ_LDS_BANK_CONFLICT = 512 of 554 LDS-active cycles per workgroup, profiles/r05_nsgan_b256_sq_pmc.txt; the stamped
// timeline shows 0.75 us between the slowest wave's last MFMA and the barrier behind these writes).  Readers take whole
// rows (32 lanes = 32 columns): any permutation of a row's columns is conflict-free for them.
__device__ __forceinline__ int red_idx(int w, int row, int col) {
    return (w * 32 + row) * 32 + (col ^ ((((row >> 2) ^ row) & 1) << 4));
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
                for (int ww = 0; ww < WAVES; ++ww) v += red[red_idx(ww, row, col)];
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
                for (int ww = 0; ww < WAVES; ++ww) v += red[red_idx(ww, row, col)];
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
        for (int ww = 0; ww < WAVES; ++ww) v += red[red_idx(ww, row, col)];
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
// (Round 6, kernel entry: the rider launches take 0.9 - 1.9 KB of arguments, loaded in four or five groups each behind
// an s_waitcnt lgkmcnt(0); the stamped timelines show 1.5 - 2.2 us between a wave's entry and its first operand load
// there.  One burst of scalar loads over the whole argument block at the top of every kernel -- so that the later groups
// hit the scalar cache -- measured SLOWER: step 68.3 -> 69.4 us, every launch +0.1 .. 0.3 us; the groups already hit.
// profiles/r06_experiments.md section 4.)
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
    static constexpr bool B_KC = (MODE == MODE_FWD);       // A(m,k) = X[m][k]; B(k,n) = W[n][k] (fwd) / W[k][n] (dx)
    static constexpr int LDK = BK + 4, LDXB = BN + 4;
    static constexpr int A_SZ = BM * LDK;
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
    static_assert(MODE != MODE_DW, "forward (A, B k-contiguous) and input gradient (B = W[k][n]) only");
    static_assert(PD >= 2, "two stages are stored before the first barrier");
    __shared__ __attribute__((aligned(16))) float lds[C::FLOATS];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wk = w / (C::WM * C::WN), wmn = w % (C::WM * C::WN), wm = wmn / C::WN, wn = wmn % C::WN;
    const int b = blockIdx.x, xcd = b & 7, l = b >> 3;
    const int tile_m = xcd * p.lds_mpx + l / p.tn, tile_n = l % p.tn;
    if (tile_m >= p.lds_tm) return;                               // workgroup-uniform
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const float* A = p.A;                                    // (launches with a ring slot take the 16-wave kernels)
    const float* B = p.B;
#ifdef GM_STAMPS
    const gm_stamps::Ctx st_slot = p.stamp; const int st_tile = (int)blockIdx.x;
#endif
    GM_STAMP_EDGE(p, false, st_tile, MODE + 10);
    GM_STAMP(st_slot, st_tile, 0);                            // entry

    constexpr int KQ = BK / 4;                                    // 16-byte units per k-contiguous row
    constexpr int XQ = BN / 4;                                    // 16-byte units per x-contiguous row
    constexpr int UA = BM * KQ, CA = (UA + NT - 1) / NT;
    constexpr int UB = C::B_KC ? BN * KQ : BK * XQ, CB = (UB + NT - 1) / NT;
    // per-thread base address of each 16-byte unit of the stage tiles (row clamps applied once)
    const float* baseA[CA];
    const float* baseB[CB];
#pragma unroll
    for (int i = 0; i < CA; ++i) {
        const int u = min(t + i * NT, UA - 1);
        baseA[i] = A + (int64_t)min(m0 + u / KQ, p.M - 1) * p.lda + 4 * (u % KQ);
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
        const int u = min(t + i * NT, UB - 1);
        if (C::B_KC) baseB[i] = B + (int64_t)min(n0 + u / KQ, p.N - 1) * p.ldb + 4 * (u % KQ);
        else baseB[i] = B + min(n0 + 4 * (u % XQ), p.N - 4);      // + k * ldb per stage
    }
    // One 16-byte unit of stage q.  The lambdas RETURN the value (a lambda that writes a captured
    // register array makes hipcc keep the array in scratch memory, with a vmcnt(0) behind every
    // load -- measured: 1.4 us per stage).  Branch-free: `inside` (the whole stage lies below K) is
    // a compile-time constant in the unrolled kernels, otherwise the k index is clamped.
    auto load_a = [&](const float* base, int i, int q, bool inside) -> float4 {
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
            *reinterpret_cast<float4*>(&Bs[kr * LDXB + 4 * xq]) = inside ? v : keep4(q * BK + kr < p.K, v);
        }
    };
    // MFMA fragments of one 8-deep k group: lane (r, h) holds k = kb + 4h + j, j = 0..3
    auto frag_a = [&](int buf, int ku, int ti) -> float4 {
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

    constexpr int NDSR = KU * (TI + (C::B_KC ? TJ : 4 * TJ));   // LDS read instructions per stage
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
        if ((s_) < 30) GM_STAMP(st_slot, st_tile, 3 + (s_));                       \
    }

    const int S = (p.K + BK - 1) / BK;
    GM_LDS_GLOAD(0, 0)
    GM_LDS_GLOAD(1 % PD, 1)
    GM_LDS_LSTORE(0, 0, 0)
    GM_LDS_LSTORE(1 % PD, 1, 1)
#pragma unroll
    for (int q = 2; q < PD + 2; ++q) { GM_LDS_GLOAD(q % PD, q) }
    GM_STAMP(st_slot, st_tile, 1);                            // two stages stored (their loads had landed), PD more requested
    __syncthreads();
    GM_LDS_FRAGS(0, 0)
    GM_STAMP_AFTER(fa[0][0][0].x);
    GM_STAMP(st_slot, st_tile, 2);                            // first fragments in registers: the pipeline starts
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
    GM_STAMP(st_slot, st_tile, 36);                           // partial tiles of the WK groups in LDS
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
        GM_STAMP(st_slot, st_tile, 37);                       // epilogue stores issued
        GM_STAMP_EDGE(p, true, st_tile, MODE + 10);
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
// (profiles/r02_experiments.md): 128x64 tiles lose to two rounds of 64x64 on the 2048x784 output; round 5 (call M):
// forcing 32x64 tiles so that TWO workgroups share a CU is no faster anywhere (fwd 2048x784->400 19.7 -> 20.0 us,
// 2048x400->784 20.6 -> 23.0, dX 1024 rows 12.1 -> 13.2): this rule stays.
inline int lds_pick_cfg(int M, int N) {
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
    if (S == NS784) hipLaunchKernelGGL((gemm_lds_kernel<MODE, BM, BN, WTM, WTN, WK, KU, PD, NS784>), grid, block, 0, s, p);
    else if (S == NS400) hipLaunchKernelGGL((gemm_lds_kernel<MODE, BM, BN, WTM, WTN, WK, KU, PD, NS400>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_lds_kernel<MODE, BM, BN, WTM, WTN, WK, KU, PD, 0>), grid, block, 0, s, p);
    GM_LAUNCH_RET();
}

// Tile configuration of the LDS kernel for this forward / input-gradient launch, or 0 when the launch is not for it
// (the caller continues with the split-reduction kernels).  1: 64x64 tile = 2 x 1 waves of 32x64 (two accumulators
// each) x 4 reduction groups, BK = 32; 2: 32x64 tile = 1 x 2 waves of 32x32 x 4 reduction groups.
// (Weight gradients through this kernel were measured slower than the split-reduction form in rounds 2 and 3 -- 2048
// rows: 35.9 / 43.0 vs 31.5 us -- and are not offered any more.)
template <int MODE>
int lds_cfg_for(const GemmP& p, bool vec, bool xv) {
    static_assert(MODE != MODE_DW, "forward and input gradient only");
    constexpr int min_m = 1024;                              // (ops.lds_min_m() mirrors it on the host side)
    if (p.M < min_m || p.K < 64 || !vec || (MODE == MODE_DX && !xv) || p.N < 32) return 0;
    return lds_pick_cfg(p.M, p.N);
}

template <int MODE>
int launch_lds(hipStream_t s, const GemmP& p, int cfg) {
    // (round 6, profiles/r06_experiments.md section 3: BK = 64 (13 barriers instead of 25), 2 x 2 waves x 2 k-groups (half the
    // partial tiles in the epilogue) and 2 register stages instead of 4 all measured level or slower than this --
    // 20.0 / 19.9 / 20.0 against 19.7 us on 2048 x 784 -> 400: the stages run at 1 150 - 1 210 cycles against 1 024 of
    // MFMA issue, what the launch loses it loses to tile quantisation, profiles/r06_ns_b1024_wave_timeline.md)
    if (cfg == 1) return launch_lds_cfg<MODE, 64, 64, 32, 64, 4, 1, 4>(s, p);
    return launch_lds_cfg<MODE, 32, 64, 32, 32, 4, 1, 4>(s, p);
}

// ------------------------------------------------------------------------------------------
// The split-reduction body on v_mfma_f32_16x16x4_f32: a wave's
// operand fragments are 16 rows x 16 k per instruction -- lane (i = lane&15, g = lane>>4) loads the
// 4 consecutive k = 16c+4g..+3 of row i, so ONE load instruction touches 16 cache lines with 64
// useful bytes each (the 32x32x2 form touches 32 lines with 32 bytes each): half the L1 tag
// lookups per byte.  The wave keeps four independent 16x16 accumulators (2x2 sub-tiles of the
// 32x32 tile), so the 40-cycle dependent latency of the 16x16x4 MFMA is always covered.
// ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- coalesced x-contiguous fragment loads (round 5) ------------------------------------------------------------
// What a wave-wide 16-byte load costs the CU's vector cache depends on WHICH LANES read adjacent bytes
// (tools/fill_probe, profiles/r05_fill_law.md; 16 waves x 4 loads in flight, L2-resident operands):
//     1 KB contiguous                                   21 cycles per instruction, 16 cache accesses
//     16 rows x 64 B, the four lanes of a QUAD adjacent   30 cycles, 16 accesses
//     16 rows x 64 B, adjacent bytes 16 LANES APART       44 cycles, 64 accesses + tag-conflict stalls
// (round 4 read the 44 cycles of its own loads as a law of the machine).  For the x-contiguous operands -- both operands
// of the weight gradient, W in the input gradient -- the coalesced form is free: lane -> k-row 4 (lane >> 4) +
// ((lane >> 2) & 3), x-quad lane & 3 (a quad reads 64 contiguous bytes of one k-row); the 4x4 transpose that turns
// "four x at one k" into "four k at one x" then runs over lane bits 3:2 (lanes l, l^4, l^8, l^12) instead of the quad,
// same number of VALU operations; and the lane ends up holding output index SIGMA(lane & 15) = 4 (lane & 3) +
// ((lane >> 2) & 3) of its 16-wide sub-tile instead of lane & 15 -- a fixed permutation of the tile's rows / columns
// that the epilogue undoes where it writes the accumulators out (XMAP).  Bit-identical results (same k per MFMA step
// and lane group).  Measured (profiles/r05_experiments.md section 2, same-call alternation): layer-1 weight gradient +
// head 12.4 -> 12.2 us, NSGAN bs=256 step 69.0 -> 68.7 us.
// The k-contiguous operands (X / W in the forward, dA in the input gradient) stay in fragment layout: their coalesced
// form needs a lane exchange (four ds_bpermute_b32 per fragment) and measured SLOWER inside the step (69.1 -> 70.9 us);
// so did requesting all of a wave's chunks before the first MFMA (69.0 -> 76.5 us) -- these launches are chains of
// dependent round trips, not vector-cache-throughput bound (same log).
__device__ __forceinline__ int sigma16(int i) { return ((i & 3) << 2) | ((i >> 2) & 3); }

__device__ __forceinline__ float4 raw_xc4_16(const float* __restrict__ P, int64_t ld, int x0, int X,
                                             int c, int K, int lane) {
    // e: which of the 4 k-rows of lane-group g; q: which x-quad of the 16-wide sub-tile
    const int e = (lane >> 2) & 3, q = lane & 3, g = lane >> 4;
    const int k = min(16 * c + 4 * g + e, K - 1);
    const int x = min(x0 + 4 * q, X - 4);
    return *reinterpret_cast<const float4*>(P + (int64_t)k * ld + x);
}

// lane ^ 8 and lane ^ 4 inside a row of 16 lanes, from direction-free DPP controls only (a rotation by half a row;
// a mirror of 8 lanes followed by a mirror of 4)
__device__ __forceinline__ float dpp_xor8(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128, 0xF, 0xF, true));   // row_ror:8
}
__device__ __forceinline__ float dpp_xor4(float v) {
    const int t = __builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true);            // row_half_mirror: l ^ 7
    return __int_as_float(__builtin_amdgcn_mov_dpp(t, 0x1B, 0xF, 0xF, true));                    // quad_perm [3,2,1,0]: ^ 3
}
// new[lane c][reg j] = old[lane j][reg c] over the 4 lanes {l, l^4, l^8, l^12} (c = (lane >> 2) & 3)
__device__ __forceinline__ float4 lane48_transpose(float4 v, int lane) {
    const bool b3 = lane & 8, b2 = lane & 4;
    const float s0 = dpp_xor8(b3 ? v.x : v.z), s1 = dpp_xor8(b3 ? v.y : v.w);
    if (b3) { v.x = s0; v.y = s1; } else { v.z = s0; v.w = s1; }
    const float t0 = dpp_xor4(b2 ? v.x : v.y), t1 = dpp_xor4(b2 ? v.z : v.w);
    if (b2) { v.x = t0; v.z = t1; } else { v.y = t0; v.w = t1; }
    return v;
}

// W consecutive floats as one load (interleaved fragments of the LDS-DMA weight gradient below)
template <int W> struct __attribute__((aligned(4))) ILV { float v[W]; };

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
// layout of gemm16_dw_dma.
template <int MI, int NI, bool ILO, bool XMAP = false>
__device__ __forceinline__ void dw_reduce_onepass(const GemmP& p, float* red, f32x4 (&acc)[MI][NI], int m0, int n0,
                                                  bool sync_first) {
    constexpr int RT = 16 * MI, CT = 16 * NI, IMG = RT * CT, G = IMG / 2;
    constexpr bool SWZ = !ILO && XMAP && (CT % 32 == 0);
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
                // XMAP: the coalesced x-contiguous loads leave output index sigma16(.) of each 16-wide sub-tile in a lane
                const int row = ILO ? MI * (4 * g4 + r) + e : 16 * e + (XMAP ? sigma16(4 * g4 + r) : 4 * g4 + r);
                const int col = ILO ? NI * i16 + f : 16 * f + (XMAP ? sigma16(i16) : i16);
                // SWZ (images whose rows are a whole number of 32-bank turns, e.g. the generator's 48 x 32 tiles): the two
                // lane groups of a half-wave sit on neighbouring rows = the same banks; flip column bit 4 on odd rows
                img[row * CT + (SWZ ? col ^ ((row & 1) << 4) : col)] = acc[e][f][r];
            }
    __syncthreads();
#ifdef GM_STAMPS
    const int st_tile = (m0 / RT) * (int)gridDim.x + n0 / CT;
#endif
    GM_STAMP(p.stamp, st_tile, 10);                      // partial tiles of all waves in LDS
    if (t >= G) return;
    const int row = t / (CT / 2), c2 = t % (CT / 2);
    float2 v = make_float2(0.f, 0.f);
#pragma unroll
    for (int ww = 0; ww < 16; ++ww) {
        const float2 x = *reinterpret_cast<const float2*>(&red[ww * IMG + row * CT + (SWZ ? (2 * c2) ^ ((row & 1) << 4) : 2 * c2)]);
        v.x += x.x; v.y += x.y;
    }
    GM_STAMP_AFTER(v.x);
    GM_STAMP(p.stamp, st_tile, 11);                      // sixteen images summed; the epilogue's round trips follow
    const int m = m0 + row, n = n0 + 2 * c2;
    if (m >= p.M) return;
    if (n + 1 < p.n_real) { store2_dw(p, v, m, n); return; }
    if (n < p.N) store_element<MODE_DW>(p, v.x, m, n);       // the ones column (bias gradient) / a ragged edge
    if (n + 1 < p.N) store_element<MODE_DW>(p, v.y, m, n + 1);
}

// Cross-wave reduction + epilogue of the interleaved-fragment weight gradient (gemm16_dw_dma): accumulator (e, f) register r of lane (i16, g4) is output (row MI*(4*g4 + r) + e, column
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
                            red[red_idx(w, row & 31, col & 31)] = acc[e][f][r];
                    }
            __syncthreads();
            reduce_and_store<MODE_DW, WAVES, (MI > 1 ? 32 : 16)>(p, red, t, m0 + 32 * bm, n0 + 32 * bnk,
                                                                 (2 * bnk + 1 < NI) ? 0x7fffffff : n0 + 32 * bnk + 16,
                                                                 (2 * bm + 1 < MI) ? 32 : 16);
        }
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
template <int MI, int NI, bool OF, int FOLD, bool SL>
__device__ __forceinline__ void gemm16_dw_dma(const GemmP& p, float* red, int bx, int by, float* sds,
                                              const FoldP* fold, int gx) {
    constexpr int WAVES = 16, NP = MI + NI, CH = NP * 256;   // pieces / floats of a wave's chunk buffer
    // XCD-aware tile map (round 6).  Workgroups go to the eight XCDs round-robin by their linear id, and each XCD has
    // its own L2: with tile = linear id every L2 pulls the WHOLE of both operands over the fabric -- at 2048 rows 8 x
    // (3.3 + 6.4) MB = 78 MB per launch, 3.9 TB/s of fabric reads for the launch's 20 us (profiles/r05_ns_b1024_pmc_*).
    // Here the workgroups that share an XCD (same id mod 8) take CONSECUTIVE tiles in m-fastest order: ~28 tiles = two or
    // three 48-column strips of X, so an L2 fetches all of dH but only its strips of X.  (gx: n-tiles of this GEMM; a
    // rotation of the XCD numbering -- the head workgroups in front of the tiles -- does not matter.)
    if (gx > 0) {
        const int gy = (p.M + 16 * MI - 1) / (16 * MI), total = gx * gy;
        const int L = by * gx + bx, x = L & 7, k = L >> 3;
        const int base = total >> 3, rem = total & 7;
        const int T = x * base + min(x, rem) + k;            // XCD x owns tiles [x * base + min(x, rem), + base + (x < rem))
        by = T % gy; bx = T / gy;                            // m fastest: neighbours share their X columns
    }
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i16 = lane & 15, g4 = lane >> 4;
    const int m0 = by * (16 * MI), n0 = bx * (16 * NI);
#ifdef GM_STAMPS
    const gm_stamps::Ctx st_slot = p.stamp; const int st_tile = by * max(gx, 1) + bx;
#endif
    GM_STAMP_EDGE(p, false, st_tile, MODE_DW + 20);
    GM_STAMP(st_slot, st_tile, 0);
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
        src[j] = (isA ? slot_base<SL>(p.A, p.a_slot) : slot_base<SL>(p.B, p.b_slot)) + col;
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
        GM_STAMP(st_slot, st_tile, 17);
        if constexpr (FOLD == 1) fold_fill_lds(*fold, sds, fold->R);   // behind the first chunk's loads; ends with a barrier
        for (int q = 0; q < nq; ++q) {
            const int c = w + q * WAVES;
            slab::wait_vm<0>();                              // this wave's pieces have landed
            if (q < 8) GM_STAMP(st_slot, st_tile, 20 + 2 * q);   // chunk q: pieces landed in LDS
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
            GM_STAMP_AFTER(acc[MI - 1][NI - 1][0]);
            if (q < 8) GM_STAMP(st_slot, st_tile, 21 + 2 * q);   // chunk q: MFMAs retired
        }
    };
    if (special) run(std::true_type{}); else run(std::false_type{});
    GM_STAMP(st_slot, st_tile, 9);
    dw_il_reduce<MI, NI>(p, red, acc, m0, n0, true);         // the ring shares `red`: everybody out of the loop first
    GM_STAMP(st_slot, st_tile, 12);
    GM_STAMP_EDGE(p, true, st_tile, MODE_DW + 20);
}

// LDS floats of the 16-wave kernels: the 64 KB block-by-block reduction buffer; for weight gradients of multi-block
// tiles the sixteen whole-tile images of dw_reduce_onepass; the DMA form's sixteen chunk buffers
template <int MODE, bool DMA, int MI, int NI> struct RedSize {
    static constexpr int base = 16 * 32 * 32;
    static constexpr int onepass = (MODE == MODE_DW && MI * NI > 4) ? 16 * 256 * MI * NI : 0;
    static constexpr int dma = (MODE == MODE_DW && DMA) ? 16 * (MI + NI) * 256 : 0;
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
// TP (with FOLD == 1): the rows' dS come from the two-phase prologue (RaGAN / Fisher critic steps, gm_head.h).
template <int MODE, bool VEC, int WAVES, int G, bool XV, int MI, int NI, bool OF = false, int FOLD = 0, bool DMA = false,
          bool TP = false, bool PK = false, bool SL = true>
__device__ __forceinline__ void gemm16_body(const GemmP& p, float* red, int bx, int by,
                                            float* sds = nullptr, const FoldP* fold = nullptr, int gx = 0) {
    static_assert(!TP || (FOLD == 1 && !DMA), "two-phase losses: folded weight gradient, operands through registers");
    static_assert(FOLD == 0 || (FOLD == 1 && MODE == MODE_DW && XV) || (FOLD == 2 && MODE == MODE_DX && VEC),
                  "folded head: 16-byte operand paths only");
    static_assert(G == 1, "per-chunk schedule (G stays in the kernel names so that they keep their shape across rounds)");
    static_assert(!PK || (WAVES == 16 && !DMA && ((MODE == MODE_FWD && VEC) || (MODE == MODE_DW && XV))),
                  "bit-packed operand rows: 16-byte operand paths through registers only");
    // DMA: the launch chose the LDS-DMA weight gradient (its own kernel instantiations: as a run-time branch inside the
    // shared kernels a second body cost the bs=256 step 1.5 us in registers and code it never runs)
    if constexpr (DMA && MODE == MODE_DW && XV && WAVES == 16 && FOLD != 2) {
        gemm16_dw_dma<MI, NI, OF, FOLD, SL>(p, red, bx, by, sds, fold, gx);
        return;
    }
    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g4 = lane >> 4;
    const int m0 = by * (16 * MI), n0 = bx * (16 * NI);
#ifdef GM_STAMPS
    const gm_stamps::Ctx st_slot = p.stamp; const int st_tile = by * (int)gridDim.x + bx;
#endif
    GM_STAMP_EDGE(p, false, st_tile, MODE);
    GM_STAMP(st_slot, st_tile, 0);                            // entry

    const float* A = slot_base<SL>(p.A, p.a_slot);
    const float* B = slot_base<SL>(p.B, p.b_slot);
    const int b_cols = (MODE == MODE_DW) ? p.n_real : p.N;
    const int ones_col = (MODE == MODE_DW && p.db) ? p.n_real : -1;
    const int nchunks = (p.K + 15) >> 4;
    GM_STAMP_AFTER(A); GM_STAMP_AFTER(B); GM_STAMP_AFTER(nchunks);
    GM_STAMP(st_slot, st_tile, 16);                           // operand bases resolved (kernel arguments + slot counters read)

    // folded head: what stays fixed per lane across the reduction
    float4 fw[FOLD == 1 ? MI : 1];
    float fds[FOLD == 2 ? MI : 1];
    // x-contiguous 16-byte operands: which x-quad / which of its lane-group's four k-rows this lane LOADS
    const int xq_ld = lane & 3, xe_ld = (lane >> 2) & 3;
    // ... and which output index of the 16-wide sub-tile it then HOLDS (XMAP: sigma16)
    constexpr bool XMAP_A = MODE == MODE_DW && XV, XMAP_B = MODE != MODE_FWD && XV;
    const int ia = XMAP_A ? sigma16(i16) : i16, ib = XMAP_B ? sigma16(i16) : i16;
    if constexpr (FOLD == 1) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)                       // w2 of this lane's four A columns
            fw[mi] = *reinterpret_cast<const float4*>(
                p.fold_w2 + min(m0 + 16 * mi + 4 * xq_ld, p.M - 4));
    }
    auto load_a = [&](int c, int mi) -> float4 {
        const int kb = 16 * c + 4 * g4, x0 = m0 + 16 * mi;
        if (MODE == MODE_DW) {
            if (XV) return raw_xc4_16(A, p.lda, x0, p.M, c, p.K, lane);
            return raw_xc(A, p.lda, x0 + i16, p.M, kb, p.K);
        }
        return raw_kc<VEC>(A, p.lda, x0 + i16, p.M, kb, p.K);
    };
    auto load_b = [&](int c, int ni) -> float4 {
        const int kb = 16 * c + 4 * g4, x0 = n0 + 16 * ni;
        if (MODE == MODE_FWD) return raw_kc<VEC>(B, p.ldb, x0 + i16, p.N, kb, p.K);
        if (XV) return raw_xc4_16(B, p.ldb, x0, b_cols, c, p.K, lane);
        return raw_xc(B, p.ldb, x0 + i16, b_cols, kb, p.K);
    };
    auto fix_a = [&](float4 v, int c, int mi, float4 wk) -> float4 {
        const int kb = 16 * c + 4 * g4, x = m0 + 16 * mi + ia;
        if constexpr (FOLD == 1)                              // the loaded row is k = 16c + 4g + e (clamped)
            v = fold_dh4(v, sds[min(16 * c + 4 * g4 + xe_ld, p.K - 1)], fw[mi]);
        if constexpr (FOLD == 2) v = fold_dh4(v, fds[mi], wk);
        if (MODE == MODE_DW)
            return fix_xc(XV ? lane48_transpose(v, lane) : v, x, p.M, kb, p.K, -1);
        return fix_kc(v, x, p.M, kb, p.K);
    };
    auto fix_b = [&](float4 v, int c, int ni) -> float4 {
        const int kb = 16 * c + 4 * g4, x = n0 + 16 * ni + ib;
        if (MODE == MODE_FWD) return fix_kc(v, x, p.N, kb, p.K);
        return fix_xc(XV ? lane48_transpose(v, lane) : v, x, b_cols, kb, p.K, ones_col, OF ? p.ones_from : 0);
    };

    // (Round 6: NOT zeroing the accumulators up front -- the first chunk's MFMAs taking 0 as their C operand, that chunk
    // peeled in front of the loop so that its loads are the kernel's first work -- removed 16 - 32 v_mov from in front of
    // the first operand loads and measured mixed: 256-row launches -0.1 us, 512-row forwards +0.2 us, step 65.85 ->
    // 66.1 us.  Not kept.  profiles/r06_experiments.md section 5.)
    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nq = (nchunks - w + WAVES - 1) / WAVES;        // chunks w, w+WAVES, ... of this wave
    // One chunk at a time: load its fragments, consume them, next chunk.  The four waves of a SIMD drift apart and
    // overlap each other's loads and MFMAs.  Everything that ADDED code to this loop lost, whatever it removed
    // (profiles/r01 .. r03_experiments.md): batches of G chunks of loads ahead of their MFMAs (waves in lockstep:
    // fwd 512x784x400 G=4 8.8 us, G=2 7.7, G=1 7.4-7.6), a rolling prefetch of the next chunk (step 71.7 -> 76.3 us),
    // k-steps outermost in the MFMA block, a fast path for interior tiles, spreading the reduction's tail over waves.
    auto consume = [&](const float4 (&ra)[MI], const float4 (&rb)[NI], float4 wk, int q) {
        const int cq = w + q * WAVES;
        float4 fa[MI], fb[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[mi] = fix_a(ra[mi], cq, mi, wk);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fb[ni] = fix_b(rb[ni], cq, ni);
        GM_STAMP_AFTER(fa[MI - 1].x); GM_STAMP_AFTER(fb[NI - 1].x);
        if (q < 4) GM_STAMP(st_slot, st_tile, 1 + 2 * q);     // chunk q: operands landed and fixed up, MFMAs start
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
        GM_STAMP_AFTER(acc[MI - 1][NI - 1][0]);
        if (q < 4) GM_STAMP(st_slot, st_tile, 2 + 2 * q);     // chunk q: last MFMA retired
    };
    auto load_wk = [&](int cc) -> float4 {                    // FOLD == 2: w2 of the chunk's four reduction columns
        if constexpr (FOLD == 2) return *reinterpret_cast<const float4*>(p.fold_w2 + min(16 * cc + 4 * g4, p.K - 4));
        return make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // The folded head's prologue (LDS fill of dS, one barrier) runs BEHIND the first operand loads: both trips to the
    // fabric are in flight together (in front of the loop it cost a second, serial round trip per launch).  Waves
    // without a chunk still take the barrier.
    auto fold_prologue = [&]() {
        if constexpr (FOLD == 1) {
            if constexpr (TP) {                               // (the uniform results are head workgroup 0's business)
                FoldTP tp_unused;
                fold_fill_lds_tp(*fold, sds, fold->R, tp_unused);
            } else {
#ifdef GM_STAMPS
                {                                             // fold_fill_lds, with the probe's stamps inside
                    for (int r = threadIdx.x; r < fold->R; r += blockDim.x) {
                        float s_ = fold_score(*fold, r);
                        GM_STAMP_AFTER(s_);
                        if (r == (int)threadIdx.x) GM_STAMP(st_slot, st_tile, 14);   // this wave's partial dots have landed
                        float ds_, l_;
                        fold_row(*fold, r, s_, ds_, l_);
                        sds[r] = ds_;
                    }
                    GM_STAMP(st_slot, st_tile, 15);           // rows done (waves without rows: at once), barrier next
                    __syncthreads();
                }
#else
                fold_fill_lds(*fold, sds, fold->R);           // every reduction row (ends with the barrier)
#endif
            }
        } else if constexpr (FOLD == 2) {
            if (t < 16 * MI) {                                // the tile's own rows
                float s_, ds_, l_;
                fold_row(*fold, min(m0 + t, fold->R - 1), s_, ds_, l_);
                sds[t] = ds_;
            }
            GM_STAMP(st_slot, st_tile, 15);
            __syncthreads();
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) fds[mi] = sds[16 * mi + i16];     // dS of this lane's A rows
        }
    };
    if constexpr (PK) {
        // Bit-packed batch rows (GemmP::pk_bits): the forward's A rows of a tile below pk_rows, the weight gradient's
        // B (= X) rows of a reduction chunk below pk_rows.  One 4-byte load per fragment instead of a 16-byte one --
        // the word that holds this lane's four elements -- expanded to 0.0f / 1.0f where the fp32 fragment would be
        // fixed up; from there on the same instructions on the same values.  Packed and fp32 chunks run in SEPARATE
        // loops (a wave's chunks w, w + 16, ... are packed up to q_pk).
        constexpr int PN = MODE == MODE_FWD ? MI : NI;
        auto pk_elem = [&](int c, int j) -> int {             // first of the four elements this lane takes from its word
            return MODE == MODE_FWD ? min(16 * c + 4 * g4, p.K - 4) : min(n0 + 16 * j + 4 * xq_ld, b_cols - 4);
        };
        auto load_pk = [&](int c, int j) -> uint32_t {
            const int row = MODE == MODE_FWD ? m0 + 16 * j + i16 : 16 * c + 4 * g4 + xe_ld;
            return p.pk_bits[(int64_t)row * p.pk_wpr + (pk_elem(c, j) >> 5)];
        };
        const int nq_s = __builtin_amdgcn_readfirstlane(nq);
        int q_pk;
        if (MODE == MODE_FWD) q_pk = (m0 < p.pk_rows) ? nq_s : 0;
        else q_pk = min(nq_s, max(0, ((p.pk_rows >> 4) - __builtin_amdgcn_readfirstlane(w) + WAVES - 1) / WAVES));
        auto do_chunk = [&](auto is_pk, auto with_prologue, int q) {
            constexpr bool P = decltype(is_pk)::value;
            float4 ra[MI], rb[NI];
            uint32_t pw[PN];
            const int cc = w + q * WAVES;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                if constexpr (P && MODE == MODE_FWD) pw[mi] = load_pk(cc, mi);
                else ra[mi] = load_a(cc, mi);
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                if constexpr (P && MODE == MODE_DW) pw[ni] = load_pk(cc, ni);
                else rb[ni] = load_b(cc, ni);
            }
            const float4 wk = load_wk(cc);
            if constexpr (decltype(with_prologue)::value) fold_prologue();
            if constexpr (P) {
#pragma unroll
                for (int j = 0; j < PN; ++j) {
                    const float4 x = pk_expand(pw[j], pk_elem(cc, j));
                    if constexpr (MODE == MODE_FWD) ra[j] = x; else rb[j] = x;
                }
            }
            consume(ra, rb, wk, q);
        };
        constexpr std::true_type yes{};
        constexpr std::false_type no{};
        int q0 = 0;
        if constexpr (FOLD != 0) {
            if (nq_s > 0) { if (q_pk > 0) do_chunk(yes, yes, 0); else do_chunk(no, yes, 0); }
            else fold_prologue();
            q0 = 1;
        }
        for (int q = q0; q < q_pk; ++q) do_chunk(yes, no, q);
        for (int q = max(q0, q_pk); q < nq_s; ++q) do_chunk(no, no, q);
    } else {
    int q_first = 0;
    if constexpr (FOLD != 0) {
        float4 ra[MI], rb[NI];
        float4 wk = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool have = nq > 0;
        // folded weight gradient: this thread's row of partial dots is requested FIRST (gm_head.h fold_part_load)
        float4 pv[4];
        if constexpr (FOLD == 1 && !TP) {
            if (t < fold->R) fold_part_load(*fold, t, pv);
        }
        // ... and the folded input gradient's: the tile's own rows (vmcnt retires in order: behind the operand loads
        // the partial dots came back after them, and the barrier that every wave's first MFMA waits for with them;
        // round 6: dX + head 6.5 - 6.7 -> 6.4 us, step -0.3 us, same bits)
        if constexpr (FOLD == 2) {
            if (t < 16 * MI) fold_part_load(*fold, min(m0 + t, fold->R - 1), pv);
        }
        if (have) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) ra[mi] = load_a(w, mi);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) rb[ni] = load_b(w, ni);
            wk = load_wk(w);
        }
        GM_STAMP(st_slot, st_tile, 17);                       // first chunk's operand loads issued
        if constexpr (FOLD == 1 && !TP) fold_fill_lds_pre(*fold, sds, fold->R, pv);
        else if constexpr (FOLD == 2) {
            if (t < 16 * MI) {                                // the tile's own rows, partial dots already requested
                float s_, ds_, l_;
                fold_row_of(*fold, min(m0 + t, fold->R - 1), fold_score_of(*fold, pv), s_, ds_, l_);
                sds[t] = ds_;
            }
            GM_STAMP(st_slot, st_tile, 15);
            __syncthreads();
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) fds[mi] = sds[16 * mi + i16];
        }
        else fold_prologue();
        GM_STAMP(st_slot, st_tile, 13);                       // folded head: dS of every reduction row rebuilt in LDS
        if (have) consume(ra, rb, wk, 0);
        q_first = 1;
    }
    for (int q = q_first; q < nq; ++q) {
        float4 ra[MI], rb[NI];
        const int cc = w + q * WAVES;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) ra[mi] = load_a(cc, mi);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) rb[ni] = load_b(cc, ni);
        const float4 wk = load_wk(cc);
        consume(ra, rb, wk, q);
    }
    }
    GM_STAMP(st_slot, st_tile, 9);                            // reduction loop done (this wave)
    if constexpr (MODE == MODE_DW && WAVES == 16 && MI * NI > 4) {
        if (p.vec_epi) {                                     // kernel-argument uniform
            dw_reduce_onepass<MI, NI, false, XMAP_A>(p, red, acc, m0, n0, false);
            GM_STAMP(st_slot, st_tile, 12);                   // epilogue stores issued
            GM_STAMP_EDGE(p, true, st_tile, MODE);
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
                // (XMAP: rows follow the A operand's lane -> output map, columns the B operand's)
                const int row = XMAP_A ? sigma16(g4 * 4 + rgi) : g4 * 4 + rgi;
                red[red_idx(w, row, ib)] = acc[2 * bm][2 * bn][rgi];
                if (2 * bn + 1 < NI) red[red_idx(w, row, 16 + ib)] = acc[2 * bm][(2 * bn + 1 < NI) ? 2 * bn + 1 : 0][rgi];
                if (2 * bm + 1 < MI) {
                    constexpr int MIX = MI > 1 ? MI - 1 : 0;             // (keeps the index in range for MI == 1)
                    const int mu = (2 * bm + 1 < MI) ? 2 * bm + 1 : MIX;
                    red[red_idx(w, 16 + row, ib)] = acc[mu][2 * bn][rgi];
                    if (2 * bn + 1 < NI)
                        red[red_idx(w, 16 + row, 16 + ib)] = acc[mu][(2 * bn + 1 < NI) ? 2 * bn + 1 : 0][rgi];
                }
            }
            __syncthreads();
            if (bm + bn == 0) GM_STAMP(st_slot, st_tile, 10); // partial tiles of all waves in LDS (first block)
            reduce_and_store<MODE, WAVES, (MI > 1 ? 32 : 16)>(p, red, t, m0 + 32 * bm, n0 + 32 * bn,
                                                              (2 * bn + 1 < NI) ? 0x7fffffff : n0 + 32 * bn + 16,
                                                              (2 * bm + 1 < MI) ? 32 : 16);
        }
    GM_STAMP(st_slot, st_tile, 12);                           // epilogue stores issued
    GM_STAMP_EDGE(p, true, st_tile, MODE);
}

template <int MODE, bool VEC, int WAVES, int G, bool XV, int MI, int NI, bool DMA = false, bool SL = false>
__global__ __launch_bounds__(WAVES * 64) void gemm16_kernel(GemmP p) {
    __shared__ __attribute__((aligned(16))) float red[(WAVES == 16) ? RedSize<MODE, DMA, MI, NI>::value : WAVES * 32 * 32];
    gemm16_body<MODE, VEC, WAVES, G, XV, MI, NI, false, 0, DMA, false, false, SL>(p, red, blockIdx.x, blockIdx.y, nullptr, nullptr,
                                                                                  (int)gridDim.x);
}

// The weight-gradient GEMM with the critic head's backward workgroups riding in the same grid:
// rows [0, hrows) of the grid are head workgroups (dispatched first), the rest are GEMM tiles.  The
// two touch disjoint outputs and neither reads what the other writes (gm_hip.h), so the launch
// boundary -- and its ~2 us of idle machine inside a graph -- between them disappears.
template <int MODE, bool VEC, int G, bool XV, int MI, int NI, bool OF = false, bool FOLDED = false, bool DMA = false,
          bool TP = false, bool PK = false>
__device__ __forceinline__ void gemm16_with_head(const GemmP& p, const HeadBwdP& hp, int hrows,
                                                 int hblocks) {
    __shared__ __attribute__((aligned(16))) float red[RedSize<MODE, DMA, MI, NI>::value];
    __shared__ float sds[FOLDED ? FOLD_MAX_ROWS : 1];
    constexpr int FOLD = FOLDED ? (MODE == MODE_DW ? 1 : 2) : 0;
    // (folded head: the dS prologue runs inside the bodies, behind their first operand loads)
    if ((int)blockIdx.y < hrows) {                           // workgroup-uniform
        const int bid = blockIdx.y * gridDim.x + blockIdx.x;
        if (bid < hblocks) head_bwd_body<TP>(hp, bid, sds);
        return;
    }
    gemm16_body<MODE, VEC, 16, G, XV, MI, NI, OF, FOLD, DMA, TP, PK, false>(p, red, blockIdx.x, blockIdx.y - hrows, sds, &hp.fold,
                                                                            (int)gridDim.x);
}

template <bool VEC, int G, bool XV, int MI, int NI, bool OF = false, bool FOLDED = false, bool DMA = false>
__global__ __launch_bounds__(1024) void gemm16_dw_head_kernel(GemmP p, HeadBwdP hp, int hrows,
                                                              int hblocks) {
    gemm16_with_head<MODE_DW, VEC, G, XV, MI, NI, OF, FOLDED, DMA>(p, hp, hrows, hblocks);
}

// The folded critic step's two launches reading the real rows of [x ; G(z)] as BITS (GemmP::pk_bits): one tile shape
// each -- the shapes pick_tile gives the MNIST critic (32x32 forward tiles, 32x48 weight-gradient tiles).
__global__ __launch_bounds__(1024) void gemm16_fwd_bits_kernel(GemmP p) {
    __shared__ __attribute__((aligned(16))) float red[RedSize<MODE_FWD, false, 2, 2>::value];
    gemm16_body<MODE_FWD, true, 16, 1, false, 2, 2, false, 0, false, false, true, false>(p, red, blockIdx.x, blockIdx.y);
}
__global__ __launch_bounds__(1024) void gemm16_dw_head_bits_kernel(GemmP p, HeadBwdP hp, int hrows, int hblocks) {
    gemm16_with_head<MODE_DW, false, 1, true, 2, 3, false, true, false, false, true>(p, hp, hrows, hblocks);
}

// The same launch for the critic steps whose loss is not a mean of per-row terms (RaGAN, Fisher): folded head with
// the two-phase prologue in every workgroup (gm_head.h fold_fill_lds_tp).  Its own kernels: the separable variants'
// instantiations do not carry the block reductions.
template <int MI, int NI>
__global__ __launch_bounds__(1024) void gemm16_dw_head_tp_kernel(GemmP p, HeadBwdP hp, int hrows, int hblocks) {
    gemm16_with_head<MODE_DW, false, 1, true, MI, NI, false, true, false, true>(p, hp, hrows, hblocks);
}

// The generator step's dX GEMM carrying the one scalar workgroup of the head (loss + tick): the
// generator-mode head_bwd has nothing else to do once head_fwd_loss wrote dH.
template <int G, int MI, int NI, bool FOLDED = false>
__global__ __launch_bounds__(1024) void gemm16_dx_head_kernel(GemmP p, HeadBwdP hp, int hrows,
                                                              int hblocks) {
    gemm16_with_head<MODE_DX, true, G, true, MI, NI, false, FOLDED>(p, hp, hrows, hblocks);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

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

// Forward GEMM whose reduction is at most 32 long (the generator's / the VAE decoder's first layer: K = z_dim = 20), round 6.
// The 16-wave kernels give such a launch 16 waves per 32 x 32 tile of which two have a chunk; the other fourteen resolve
// their arguments, zero accumulators, write zero partial tiles and sit in the barrier in front of a 16-image sum --
// 2.1 us of workgroup residence for 0.4 us of loads and MFMAs (stamped; profiles/r06_experiments.md section 15).  Here
// a WAVE owns a 16 x 32 piece outright: both chunks' fragments requested up front (6 loads), 16 MFMAs, no LDS, no
// barrier, the epilogue straight from the accumulators; 4 waves = a 32 x 64 workgroup tile.
// Bit-identical to the 16-wave form: chunk c's four MFMAs start from a zero accumulator exactly as wave c's did, and
// the two partial tiles are added in wave order onto 0.f (reduce_and_store: v = 0; v += image[w], the other fourteen
// images being zeros); fragments and fix-ups are the same functions.  The batch gather's workgroups ride in rows
// [0, grows) of the grid as in gemm16_fwd_gather_kernel (4 image rows per workgroup here).
template <bool SL, bool GATHER>
__global__ __launch_bounds__(256) void gemm16_k32_fwd_kernel(GemmP p, GatherP gp, int grows, int gblocks) {
    if constexpr (GATHER) {
        if ((int)blockIdx.y < grows) {                       // workgroup-uniform
            const int bid = blockIdx.y * gridDim.x + blockIdx.x;
            if (bid < gblocks) gather_body(gp, bid);
            return;
        }
    }
    const int by = (int)blockIdx.y - (GATHER ? grows : 0), bx = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i16 = lane & 15, g4 = lane >> 4;
    const int m0 = 32 * by + 16 * (w >> 1), n0 = 64 * bx + 32 * (w & 1);
#ifdef GM_STAMPS
    const gm_stamps::Ctx st_slot = p.stamp; const int st_tile = by * (int)gridDim.x + bx;
#endif
    GM_STAMP_EDGE(p, false, st_tile, MODE_FWD);
    GM_STAMP(st_slot, st_tile, 0);                            // entry
    if (m0 >= p.M || n0 >= p.N) { GM_STAMP_EDGE(p, true, st_tile, MODE_FWD); return; }   // wave-uniform (no barrier in this kernel)
    const float* A = slot_base<SL>(p.A, p.a_slot);
    const float* B = slot_base<SL>(p.B, p.b_slot);
    float4 ra[2], rb[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {                            // (K <= 16: chunk 1 is clamped and fixed to zeros)
        const int kb = 16 * c + 4 * g4;
        ra[c] = raw_kc<true>(A, p.lda, m0 + i16, p.M, kb, p.K);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) rb[c][ni] = raw_kc<true>(B, p.ldb, n0 + 16 * ni + i16, p.N, kb, p.K);
    }
    f32x4 part[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int kb = 16 * c + 4 * g4;
        const float4 fa = fix_kc(ra[c], m0 + i16, p.M, kb, p.K);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const float4 fb = fix_kc(rb[c][ni], n0 + 16 * ni + i16, p.N, kb, p.K);
            f32x4 c4 = f32x4{0.f, 0.f, 0.f, 0.f};
            c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.x, fb.x, c4, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.y, fb.y, c4, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.z, fb.z, c4, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.w, fb.w, c4, 0, 0, 0);
            part[c][ni] = c4;
            if (c == 0 && ni == 0) { GM_STAMP_AFTER(fa.x); GM_STAMP_AFTER(fb.x); GM_STAMP(st_slot, st_tile, 1); }   // first operands in
        }
    }
    GM_STAMP_AFTER(part[1][1][0]);
    GM_STAMP(st_slot, st_tile, 2);                            // MFMAs retired
    // store_element<MODE_FWD>'s arithmetic per element (bias, activation, the interpolation rider), with the
    // kernel-argument-uniform branches outside the eight elements
    float v[2][4];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int n = min(n0 + 16 * ni + i16, p.N - 1);
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = 0.f;
            x += part[0][ni][r];
            x += part[1][ni][r];
            if (p.bias) x += bv;
            v[ni][r] = x;
        }
    }
    if (p.epi == GM_ACT_RELU) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[ni][r] = fmaxf(v[ni][r], 0.f);
    } else if (p.epi == GM_ACT_SIGMOID) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[ni][r] = gm_sigmoid(v[ni][r]);
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int n = n0 + 16 * ni + i16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 4 * g4 + r;
            if (m < p.M && n < p.N) p.C[(int64_t)m * p.ldc + n] = v[ni][r];
        }
    }
    if (p.ip_out) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int n = n0 + 16 * ni + i16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 4 * g4 + r;
                if (m < p.M && n < p.N && m < p.ip_rows) {
                    const float ev = (p.ip_eps + gm_slot_offset(p.ip_slot))[m];
                    p.ip_out[(int64_t)m * p.ip_ldo + n] = gm_interp_unfused(ev, p.ip_x[(int64_t)m * p.ip_ldx + n], v[ni][r]);
                }
            }
        }
    }
    GM_STAMP(st_slot, st_tile, 12);                           // stores issued
    GM_STAMP_EDGE(p, true, st_tile, MODE_FWD);
}

// Two weight-gradient GEMMs over the same batch rows (same reduction length, same tile shape) as
// ONE launch: workgroups [0, na) are tiles of the first, the rest tiles of the second.  The
// generator step's dW2 (784x401) and dW1 (400x21) are independent once dH is known.
// SLA / SLB: which of the two resolves ring slots on its operands (the generator's pair: only the second, whose X is the
// noise ring; the VAE's pairs: neither)
template <int G, bool XV, int MI, int NI, bool DMA = false, bool SLA = true, bool SLB = true>
__global__ __launch_bounds__(1024) void gemm16_dw_pair_kernel(GemmP pa, GemmP pb, int na, int tna,
                                                              int tnb) {
    __shared__ __attribute__((aligned(16))) float red[RedSize<MODE_DW, DMA, MI, NI>::value];
    const int id = blockIdx.x;
    if (id < na) gemm16_body<MODE_DW, false, 16, G, XV, MI, NI, false, 0, DMA, false, false, SLA>(pa, red, id % tna, id / tna);
    else gemm16_body<MODE_DW, false, 16, G, XV, MI, NI, false, 0, DMA, false, false, SLB>(pb, red, (id - na) % tnb, (id - na) / tnb);
}

// The same launch closing a VAE batch (round 4): workgroup 0 adds up the batch's reconstruction and KL partials
// (gm_fin2_sums: what gm_sum_finalize2_tick's own launch did, same order, same bits -- their producers finished
// launches ago), the others are the pair's tiles.  The device step counter may only advance once every workgroup has
// resolved its slots (Adam's schedule slot in the tiles' epilogues, the loss slot in workgroup 0): each workgroup
// arrives on f.done when it is finished and the LAST arriver ticks and re-arms the counter.
template <int G, bool XV, int MI, int NI, bool DMA = false, bool SLA = true, bool SLB = true>
__global__ __launch_bounds__(1024) void gemm16_dw_pair_fin_kernel(GemmP pa, GemmP pb, int na, int tna, int tnb,
                                                                  gm_fin2 f) {
    __shared__ __attribute__((aligned(16))) float red[RedSize<MODE_DW, DMA, MI, NI>::value];
    const int id = (int)blockIdx.x - 1;
    if (id < 0) gm_fin2_sums(f, reinterpret_cast<double*>(red));
    else if (id < na) gemm16_body<MODE_DW, false, 16, G, XV, MI, NI, false, 0, DMA, false, false, SLA>(pa, red, id % tna, id / tna);
    else gemm16_body<MODE_DW, false, 16, G, XV, MI, NI, false, 0, DMA, false, false, SLB>(pb, red, (id - na) % tnb, (id - na) / tnb);
    __syncthreads();                                         // every thread's slot reads are behind it
    if (threadIdx.x == 0) {
        const unsigned int arrived = __hip_atomic_fetch_add(f.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == gridDim.x - 1) {
            __hip_atomic_store(f.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (f.tick) *f.tick += 1;
        }
    }
}

// May the epilogue move whole float4s (store4)?  Every array it touches 16-byte aligned, leading dimensions in
// whole float4s.
template <int MODE>
int vec_epi_ok(const GemmP& p) {
    if (!aligned16(p.C) || p.ldc % 4 != 0) return 0;
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
    const gm_fin2* fin = nullptr;        // MODE_DW pair: the VAE batch's two loss sums + counter tick
};

// Tile shapes of the 16-wave kernels, as sub-tiles (16 x 16) per wave: MI x NI
enum { T22 = 0, T24, T42, T12, T23, T32 };       // 32x32, 32x64, 64x32, 16x32, 32x48, 48x32
inline int tile_mi(int t) { return t == T42 ? 4 : t == T32 ? 3 : t == T12 ? 1 : 2; }
inline int tile_ni(int t) { return t == T24 ? 4 : t == T23 ? 3 : 2; }

// Tile shape of a launch from its tile count (all measured: profiles/r01 .. r03_experiments.md):
//   32x32 by default;
//   16x32 when there are at most 128 32x32 tiles (twice the workgroups, half the MFMA chain each);
//   32x64 / 64x32 when there are more tiles than CUs and a wave still gets >= 2 chunks of reduction per tile;
//   weight gradients: 32x48 / 48x32 wherever that covers the output in one round of <= 256 workgroups (221 instead of
//   169 or 325 workgroups on the 400 x 785 / 784 x 401 outputs); forward over 3B = 768 rows: 48x32 (208 workgroups).
template <int MODE>
int pick_tile(const GemmP& p) {
    const int tm = (p.M + TM - 1) / TM, tn = (p.N + TN - 1) / TN;
    int t = T22;
    if (tm * tn > 256 && (p.K + 15) / 16 >= 32) t = (tn >= tm) ? T24 : T42;
    else if (tm * tn <= 128 && p.M > 16) t = T12;      // (16x32 for the 208-tile 2B-row forwards too: 68.8 -> 72.9 us per step, round 5)
    if (MODE == MODE_DW) {
        if (t == T24 && tm * ((p.N + 47) / 48) <= 256) t = T23;
        else if (t == T42 && tn * ((p.M + 47) / 48) <= 256) t = T32;
        else if (t == T22 && tm * tn > 256) {
            if (tn >= tm && tm * ((p.N + 47) / 48) <= 256) t = T23;
            else if (tn < tm && tn * ((p.M + 47) / 48) <= 256) t = T32;
        }
    }
    if (MODE == MODE_FWD && t == T42 && !p.hd_part && !p.sq_part && tn * ((p.M + 47) / 48) <= 256) t = T32;
    return t;
}

// LAUNCH(MI, NI, DMA) for the tile `tile`; the LDS-DMA instantiations exist for the 48-wide weight-gradient tiles
#define GM_TILE_SWITCH(tile, dma, LAUNCH)                                           \
    do {                                                                            \
        switch (tile) {                                                             \
        case T24: LAUNCH(2, 4, false); break;                                       \
        case T42: LAUNCH(4, 2, false); break;                                       \
        case T12: LAUNCH(1, 2, false); break;                                       \
        case T23: if (dma) LAUNCH(2, 3, true); else LAUNCH(2, 3, false); break;     \
        case T32: if (dma) LAUNCH(3, 2, true); else LAUNCH(3, 2, false); break;     \
        default: LAUNCH(2, 2, false); break;                                        \
        }                                                                           \
    } while (0)

template <int MODE>
int launch(hipStream_t s, const GemmP& p_in, bool vec, bool xvec = false, const Rider& rider = Rider()) {
    const HeadBwdP* head = rider.head;
    // folded critic head: the head workgroups AND the GEMM's A operand depend on the fold -- only the riding launches
    // below implement it; every other configuration is refused, never silently run without it
    const bool folded = head && head->fold.enabled;
    GemmP p = p_in;
    p.vec_epi = vec_epi_ok<MODE>(p);
#ifdef GM_STAMPS
    p.stamp.slot = gm_stamps::host_buf ? gm_stamps::host_buf + (size_t)(gm_stamps::host_next++ % gm_stamps::SLOTS) * gm_stamps::SLOT_WORDS
                                       : nullptr;
    p.stamp.tile = gm_stamps::host_tile;
#endif
    const bool xv = xvec && MODE != MODE_FWD;
    // ring slots on the operands (the generator's first layer on the noise ring): the slot-resolving instantiations of
    // the plain / gather / pair kernels; the head-riding, bit-packed and LDS macro-tile kernels are compiled without
    const bool slots = has_slot(p.a_slot) || has_slot(p.b_slot);
    if (slots && (folded || p.pk_bits)) {
        gm_set_error("ring slots on the operands of a folded-head / bit-packed launch are not supported");
        return GM_EINVAL;
    }
    const bool ride_head = head && !slots;                   // (else the head workgroups get their own launch below)
    // weight gradients over >= GM_DW_DMA_MIN_K rows on 16-byte aligned operands: chunks by LDS-DMA (gemm16_dw_dma).
    // Measured (profiles/r04_experiments.md): 2048 rows 26.0 -> 24.2 us, 1024 rows 15.3 -> 15.2, 768 rows 12.2 -> 11.9;
    // below that the extra hop through LDS costs more than the load instructions it saves (512 rows 9.0 -> 9.2, 256
    // rows 6.3 -> 7.0): the 16-byte + quad-transpose form keeps the short reductions.  (32-bit element offsets.)
    auto dma_ok = [&](const GemmP& q, bool qxv) {
        constexpr int dma_min_k = 768;
        return MODE == MODE_DW && qxv && q.K >= dma_min_k && q.M >= 4 && q.n_real >= 4 &&
               ((int64_t)q.K + 64) * (q.lda > q.ldb ? q.lda : q.ldb) < (1ll << 31);
    };
    p.dma = dma_ok(p, xv);
    // bit-packed operand rows: the two launches of the folded critic step only, one tile shape each (every other
    // configuration is refused -- there is no fp32 copy of those rows to fall back to)
    if (p.pk_bits) {
        if constexpr (MODE == MODE_FWD) {
            if (!vec || !p.hd_part || rider.gather || p.ip_out) {
                gm_set_error("bit-packed rows: the forward needs 16-byte aligned operands and the folded head's partial dots");
                return GM_EINVAL;
            }
            const dim3 bgrid((p.N + 31) / 32, (p.M + 31) / 32);
            hipLaunchKernelGGL(gemm16_fwd_bits_kernel, bgrid, dim3(1024), 0, s, p);
            GM_LAUNCH_RET();
        } else if constexpr (MODE == MODE_DW) {
            if (!head || !folded || head->fold.enabled != 1 || !xv || p.ones_from > 0 || rider.pair) {
                gm_set_error("bit-packed rows: the weight gradient needs the folded head of a separable loss and "
                             "16-byte aligned operands");
                return GM_EINVAL;
            }
            p.dma = false;                                   // operands through registers (the expansion lives there)
            const dim3 bgrid((p.N + 47) / 48, (p.M + 31) / 32);
            const int hblocks = gm_head_bwd_blocks(*head);
            const int hrows = (hblocks + (int)bgrid.x - 1) / (int)bgrid.x;
            hipLaunchKernelGGL(gemm16_dw_head_bits_kernel, dim3(bgrid.x, bgrid.y + hrows), dim3(1024), 0, s, p, *head,
                               hrows, hblocks);
            GM_LAUNCH_RET();
        } else {
            gm_set_error("bit-packed rows: forward and weight gradient only");
            return GM_EINVAL;
        }
    }
    // many-row forward / input-gradient launches: LDS-staged macro tiles.  Riders get their own launch first (a head /
    // gather workgroup set is microseconds next to a >= 1024-row GEMM).
    if constexpr (MODE != MODE_DW) {
        if (!rider.pair && !folded && !p.hd_part && !p.sq_part && !slots) {
            const int cfg = lds_cfg_for<MODE>(p, vec, xv);
            if (cfg) {
                if (head) hipLaunchKernelGGL(head_bwd_kernel, dim3(gm_head_bwd_blocks(*head)), dim3(1024), 0, s, *head);
                if (rider.gather)
                    hipLaunchKernelGGL(gather_rows_kernel, dim3(gm_gather_blocks(*rider.gather, 4)), dim3(256), 0, s, *rider.gather);
                return launch_lds<MODE>(s, p, cfg);
            }
        }
    }
    // forwards over a reduction of at most 32: one wave per 16 x 32 piece, no cross-wave reduction (gemm16_k32_fwd_kernel)
    if constexpr (MODE == MODE_FWD) {
        if (vec && p.K <= 32 && !head && !p.hd_part && !p.sq_part) {
            const dim3 kgrid((p.N + 63) / 64, (p.M + 31) / 32);
            if (rider.gather) {
                const GatherP& gp = *rider.gather;
                const int gblocks = gm_gather_blocks(gp, 4);
                const int grows = (gblocks + (int)kgrid.x - 1) / (int)kgrid.x;
                const dim3 ggrid(kgrid.x, kgrid.y + grows);
                if (slots) hipLaunchKernelGGL((gemm16_k32_fwd_kernel<true, true>), ggrid, dim3(256), 0, s, p, gp, grows, gblocks);
                else hipLaunchKernelGGL((gemm16_k32_fwd_kernel<false, true>), ggrid, dim3(256), 0, s, p, gp, grows, gblocks);
            } else {
                if (slots) hipLaunchKernelGGL((gemm16_k32_fwd_kernel<true, false>), kgrid, dim3(256), 0, s, p, GatherP{}, 0, 0);
                else hipLaunchKernelGGL((gemm16_k32_fwd_kernel<false, false>), kgrid, dim3(256), 0, s, p, GatherP{}, 0, 0);
            }
            GM_LAUNCH_RET();
        }
    }
    const int tile = pick_tile<MODE>(p);
    const int mi = tile_mi(tile), ni = tile_ni(tile);
    const dim3 grid((p.N + 16 * ni - 1) / (16 * ni), (p.M + 16 * mi - 1) / (16 * mi));
    if constexpr (MODE == MODE_DW) {
        if (ride_head) {                 // the critic head's backward workgroups ride in rows [0, hrows) of the grid
            const int hblocks = gm_head_bwd_blocks(*head);
            const int hrows = (hblocks + (int)grid.x - 1) / (int)grid.x;
            const dim3 hgrid(grid.x, grid.y + hrows);
            if (folded && (!xv || p.ones_from > 0)) {
                gm_set_error("folded head: the weight gradient needs 16-byte aligned operands and no stacked rows");
                return GM_EINVAL;
            }
            // template arguments: VEC (a k-contiguous notion: false), G, XV, MI, NI, OF (ones column with a row offset:
            // WGAN-GP's stacked gradient), FOLDED, DMA
#define GM_LH(X, OFV, FD) GM_TILE_SWITCH(tile, p.dma && (X), GM_LH1_##X##_##OFV##_##FD)
#define GM_LHK(X, OFV, FD, MI_, NI_, D_) hipLaunchKernelGGL((gemm16_dw_head_kernel<false, 1, X, MI_, NI_, OFV, FD, ((X) && (D_))>), hgrid, dim3(1024), 0, s, p, *head, hrows, hblocks)
#define GM_LH1_true_false_true(MI_, NI_, D_) GM_LHK(true, false, true, MI_, NI_, D_)
#define GM_LH1_true_true_false(MI_, NI_, D_) GM_LHK(true, true, false, MI_, NI_, D_)
#define GM_LH1_true_false_false(MI_, NI_, D_) GM_LHK(true, false, false, MI_, NI_, D_)
#define GM_LH1_false_true_false(MI_, NI_, D_) GM_LHK(false, true, false, MI_, NI_, D_)
#define GM_LH1_false_false_false(MI_, NI_, D_) GM_LHK(false, false, false, MI_, NI_, D_)
            if (folded && head->fold.enabled == 2) {
                // RaGAN / Fisher: operands through registers (the two-phase prologue's scratch and the DMA form's
                // chunk buffers would share LDS)
                p.dma = false;
#define GM_LHTP(MI_, NI_, D_) hipLaunchKernelGGL((gemm16_dw_head_tp_kernel<MI_, NI_>), hgrid, dim3(1024), 0, s, p, *head, hrows, hblocks)
                GM_TILE_SWITCH(tile, false, GM_LHTP);
#undef GM_LHTP
            } else if (folded) GM_LH(true, false, true);
            else if (xv) { if (p.ones_from > 0) GM_LH(true, true, false); else GM_LH(true, false, false); }
            else { if (p.ones_from > 0) GM_LH(false, true, false); else GM_LH(false, false, false); }
#undef GM_LH1_false_false_false
#undef GM_LH1_false_true_false
#undef GM_LH1_true_false_false
#undef GM_LH1_true_true_false
#undef GM_LH1_true_false_true
#undef GM_LHK
#undef GM_LH
            GM_LAUNCH_RET();
        }
    }
    if constexpr (MODE == MODE_DX) {
        if (ride_head && vec && xv && (tile == T22 || tile == T12)) {   // the generator-mode head's scalar workgroup rides along
            const int hblocks = gm_head_bwd_blocks(*head);
            const int hrows = (hblocks + (int)grid.x - 1) / (int)grid.x;
            const dim3 hgrid(grid.x, grid.y + hrows);
            if (tile == T12) {
                if (folded) hipLaunchKernelGGL((gemm16_dx_head_kernel<1, 1, 2, true>), hgrid, dim3(1024), 0, s, p, *head, hrows, hblocks);
                else hipLaunchKernelGGL((gemm16_dx_head_kernel<1, 1, 2, false>), hgrid, dim3(1024), 0, s, p, *head, hrows, hblocks);
            } else {
                if (folded) hipLaunchKernelGGL((gemm16_dx_head_kernel<1, 2, 2, true>), hgrid, dim3(1024), 0, s, p, *head, hrows, hblocks);
                else hipLaunchKernelGGL((gemm16_dx_head_kernel<1, 2, 2, false>), hgrid, dim3(1024), 0, s, p, *head, hrows, hblocks);
            }
            GM_LAUNCH_RET();
        }
    }
    if (folded) {
        gm_set_error("folded head: this launch configuration cannot carry it (needs the riding kernels: 16-byte "
                     "aligned operands)");
        return GM_EINVAL;
    }
    if (head)        // this configuration cannot carry the head workgroups: separate launch
        hipLaunchKernelGGL(head_bwd_kernel, dim3(gm_head_bwd_blocks(*head)), dim3(1024), 0, s, *head);
    if constexpr (MODE == MODE_FWD) {
        if (rider.gather) {
            const GatherP& gp = *rider.gather;
            if (vec && (tile == T22 || tile == T12)) {           // the batch gather rides in rows [0, grows) of the grid
                const int gblocks = gm_gather_blocks(gp, 16);
                const int grows = (gblocks + (int)grid.x - 1) / (int)grid.x;
                const dim3 ggrid(grid.x, grid.y + grows);
                if (tile == T12) hipLaunchKernelGGL((gemm16_fwd_gather_kernel<true, 1, 1, 2>), ggrid, dim3(1024), 0, s, p, gp, grows, gblocks);
                else hipLaunchKernelGGL((gemm16_fwd_gather_kernel<true, 1, 2, 2>), ggrid, dim3(1024), 0, s, p, gp, grows, gblocks);
                GM_LAUNCH_RET();
            }
            hipLaunchKernelGGL(gather_rows_kernel, dim3(gm_gather_blocks(gp, 4)), dim3(256), 0, s, gp);
        }
    }
    if constexpr (MODE == MODE_DW) {
        if (rider.pair) {
            GemmP pb = *rider.pair;
            pb.vec_epi = vec_epi_ok<MODE_DW>(pb);
            if (xv && rider.pair_xvec && tile != T12 && pb.K == p.K) {
                // both GEMMs on one tile shape; LDS-DMA only when both may take it
                const bool dma = p.dma && dma_ok(pb, true);
                p.dma = pb.dma = dma;
                const int tna = (int)grid.x, na = (int)(grid.x * grid.y);
                const int tnb = (pb.N + 16 * ni - 1) / (16 * ni), tmb = (pb.M + 16 * mi - 1) / (16 * mi);
                const dim3 pgrid(na + tnb * tmb);
                const bool sl_b = has_slot(pb.a_slot) || has_slot(pb.b_slot);
#define GM_LP2(K_, GRID_, MI_, NI_, D_, ...)                                                                              \
    do {                                                                                                                  \
        if (slots) hipLaunchKernelGGL((K_<1, true, MI_, NI_, D_, true, true>), GRID_, dim3(1024), 0, s, __VA_ARGS__);     \
        else if (sl_b) hipLaunchKernelGGL((K_<1, true, MI_, NI_, D_, false, true>), GRID_, dim3(1024), 0, s, __VA_ARGS__); \
        else hipLaunchKernelGGL((K_<1, true, MI_, NI_, D_, false, false>), GRID_, dim3(1024), 0, s, __VA_ARGS__);         \
    } while (0)
#define GM_LP(MI_, NI_, D_) GM_LP2(gemm16_dw_pair_kernel, pgrid, MI_, NI_, D_, p, pb, na, tna, tnb)
#define GM_LPF(MI_, NI_, D_) GM_LP2(gemm16_dw_pair_fin_kernel, dim3(pgrid.x + 1), MI_, NI_, D_, p, pb, na, tna, tnb, *rider.fin)
                if (rider.fin) GM_TILE_SWITCH(tile, dma, GM_LPF);
                else GM_TILE_SWITCH(tile, dma, GM_LP);
#undef GM_LP2
#undef GM_LPF
#undef GM_LP
                GM_LAUNCH_RET();
            }
            // not pairable in this configuration: the second GEMM gets its own launch afterwards (and the sums theirs)
            Rider none;
            int rc = launch<MODE_DW>(s, p_in, vec, xvec, none);
            if (rc) return rc;
            rc = launch<MODE_DW>(s, pb, false, rider.pair_xvec, none);
            if (rc || !rider.fin) return rc;
            const gm_fin2& f = *rider.fin;
            return gm_sum_finalize2_tick(s, f.pa, f.na, f.sa, f.oa, f.slot_a, f.pb, f.nb, f.sb, f.ob, f.slot_b, f.tick);
        }
    }
    // template arguments: MODE, VEC, waves, G, XV, MI, NI, DMA
    // (ring slots exist on the batch-row operand of forward and weight-gradient launches only: gm_hip.h)
    if (MODE == MODE_DX && slots) { gm_set_error("ring slots: forward and weight-gradient operands only"); return GM_EINVAL; }
#define GM_L16K(V, X, MI_, NI_, D_)                                                                                          \
    do {                                                                                                                     \
        if (MODE != MODE_DX && slots)                                                                                        \
            hipLaunchKernelGGL((gemm16_kernel<MODE, V, 16, 1, X, MI_, NI_, (MODE == MODE_DW && (X) && (D_)), MODE != MODE_DX>), grid, dim3(1024), 0, s, p); \
        else                                                                                                                 \
            hipLaunchKernelGGL((gemm16_kernel<MODE, V, 16, 1, X, MI_, NI_, (MODE == MODE_DW && (X) && (D_)), false>), grid, dim3(1024), 0, s, p); \
    } while (0)
#define GM_L16_tt(MI_, NI_, D_) GM_L16K(true, true, MI_, NI_, D_)
#define GM_L16_tf(MI_, NI_, D_) GM_L16K(true, false, MI_, NI_, D_)
#define GM_L16_ft(MI_, NI_, D_) GM_L16K(false, true, MI_, NI_, D_)
#define GM_L16_ff(MI_, NI_, D_) GM_L16K(false, false, MI_, NI_, D_)
    if (xv) { if (vec) GM_TILE_SWITCH(tile, p.dma, GM_L16_tt); else GM_TILE_SWITCH(tile, p.dma, GM_L16_ft); }
    else    { if (vec) GM_TILE_SWITCH(tile, false, GM_L16_tf); else GM_TILE_SWITCH(tile, false, GM_L16_ff); }
#undef GM_L16_ff
#undef GM_L16_ft
#undef GM_L16_tf
#undef GM_L16_tt
#undef GM_L16K
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

// gm_linear_fwd_headpart whose first `rows` rows of X are read from the packed copy (gm_gather_rows_bits_packed)
extern "C" int gm_linear_fwd_headpart_bits(void* stream, const float* X, int64_t ldx, gm_slot x_slot,
                                           const float* W, const float* bias, float* Y, int64_t ldy, int M,
                                           int K, int N, int act, const float* w2, const float* b2,
                                           float* part, int64_t ldp, float* snap, const uint32_t* xbits,
                                           int words_per_row, int rows) {
    GM_CHECK_ARG(X && W && Y && M > 0 && K > 0 && N > 0 && ldx >= K && ldy >= N);
    GM_CHECK_ARG(act >= GM_ACT_ID && act <= GM_ACT_SIGMOID);
    GM_CHECK_ARG(w2 && b2 && part && snap && ldp >= (N + 31) / 32 && ldp % 4 == 0);
    GM_CHECK_ARG(part != Y && snap != Y && (const float*)part != X && (const float*)snap != X);
    GM_CHECK_ARG(xbits && rows > 0 && rows <= M && rows % 32 == 0 && words_per_row * 32 >= K && K % 4 == 0);
    GM_CHECK_ARG((const void*)xbits != (const void*)Y && (const void*)xbits != (const void*)part);
    GemmP p{};
    p.A = X; p.B = W; p.C = Y; p.M = M; p.N = N; p.K = K;
    p.lda = ldx; p.ldb = K; p.ldc = ldy; p.bias = bias; p.epi = act;
    p.a_slot = x_slot; p.b_slot = no_slot();
    p.hd_w2 = w2; p.hd_b2 = b2; p.hd_part = part; p.hd_ldp = ldp; p.hd_snap = snap;
    p.pk_bits = xbits; p.pk_wpr = words_per_row; p.pk_rows = rows;
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

extern "C" int gm_linear_fwd_gather_bits_packed(void* stream, const float* X, int64_t ldx, gm_slot x_slot,
                                                const float* W, const float* bias, float* Y, int64_t ldy, int M,
                                                int K, int N, int act, const uint32_t* bits, int words_per_row,
                                                int64_t n_rows, const int64_t* idx, gm_slot idx_slot,
                                                uint32_t* out_bits, int B) {
    GatherP g{};
    const int rc = gm_gather_fill_bits_packed(bits, words_per_row, n_rows, idx, idx_slot, out_bits, B, &g);
    if (rc) return rc;
    return fwd_gather_impl(stream, X, ldx, x_slot, W, bias, Y, ldy, M, K, N, act, g, reinterpret_cast<float*>(out_bits));
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

// gm_linear_bwd_dw_adam_head_fold whose first `rows` rows of X are read from the packed copy
extern "C" int gm_linear_bwd_dw_adam_head_fold_bits(void* stream, const float* H, int64_t ldh,
                                                    const float* X, int64_t ldx, gm_slot x_slot, float* dW,
                                                    float* db, int M, int K, int N, float* pW, float* mW,
                                                    float* vW, float* pb, float* mb, float* vb,
                                                    const float* sched, gm_slot sched_slot, double beta1,
                                                    double beta2, double eps, double weight_decay, float clamp,
                                                    const gm_head_bwd_args* head, const gm_head_fold_args* fold,
                                                    const uint32_t* xbits, int words_per_row, int rows) {
    GM_CHECK_ARG(head && fold && !head->gen_mode && head->H == H && 2 * head->B == M && head->Hd == N);
    GM_CHECK_ARG(!sched || (db && pW && mW && vW && pb && mb && vb));
    GM_CHECK_ARG((const float*)head->w2 != pW || !pW);
    GM_CHECK_ARG(head->gw2 != dW && (const float*)dW != H);
    GM_CHECK_ARG(fold->snap != (const float*)head->w2 && fold->snap != (const float*)head->b2);
    GM_CHECK_ARG(xbits && rows > 0 && rows <= M && rows % 32 == 0 && words_per_row * 32 >= K && K % 4 == 0);
    GM_CHECK_ARG((const void*)xbits != (const void*)dW && (const void*)xbits != (const void*)pW);
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
    GemmP p{};
    bool xvec = false;
    const int rf = dw_fill(H, ldh, X, ldx, x_slot, dW, db, M, K, N, 0, sched ? &a : nullptr, &p, &xvec);
    if (rf) return rf;
    p.fold_w2 = fold->snap;
    if (!aligned16(fold->snap)) xvec = false;
    p.pk_bits = xbits; p.pk_wpr = words_per_row; p.pk_rows = rows;
    Rider r;
    r.head = &hp;
    return launch<MODE_DW>((hipStream_t)stream, p, false, xvec, r);
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

extern "C" int gm_linear_bwd_dw_adam_pair_finalize(void* stream, const gm_dw_adam_args* first,
                                                   const gm_dw_adam_args* second, const gm_finalize2_args* fin) {
    GM_CHECK_ARG(first && second && fin);
    GM_CHECK_ARG(fin->pa && fin->pb && fin->out_a && fin->out_b && fin->na > 0 && fin->nb > 0 && fin->done);
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
    const gm_fin2 f{fin->pa, fin->na, fin->scale_a, fin->out_a, fin->slot_a, fin->pb, fin->nb, fin->scale_b,
                    fin->out_b, fin->slot_b, fin->tick, fin->done};
    Rider r;
    r.pair = &pb;
    r.pair_xvec = xb;
    r.fin = &f;
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
