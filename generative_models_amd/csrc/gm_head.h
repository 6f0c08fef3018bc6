// Backward half of the fused critic head (see gm_fused.hip for the forward half and the C-ABI):
// shared between its own launch (gm_head_bwd*) and the weight-gradient GEMM that can carry the
// head workgroups in its grid (gm_linear_bwd_dw_adam_head) -- the two are independent once
// head_fwd_loss has produced dH, so one launch does both.
#pragma once
#include "gm_common.h"

// ---- folded head (round 3) ---------------------------------------------------------------------
// The N = 1 critic layer needs no launch of its own: the hidden layer's forward GEMM leaves per-
// column-tile partial dots  part[r][j] = sum_{n in tile j} h[r,n] * w2[n]  (and a snapshot of w2 / b2
// as it saw them) in its epilogue; every consumer rebuilds score, row loss and dS for the rows it
// needs in a short prologue (`fold_row`: nparts loads per row, fixed summation order => the same bits
// in every workgroup and from run to run), and the hidden-layer gradient dH[r,n] = dS_r * w2[n] *
// [h[r,n] > 0] is formed in registers where the weight-gradient / input-gradient GEMM loads its A
// operand (it loads h instead of a materialised dH).  ns_gan.py:57-60,191-192,214.
struct FoldP {
    const float* part; int64_t ldp; int nparts;    // part[r * ldp + j], j < nparts <= 16 <= ldp, ldp % 4 == 0
    const float* snap;                          // [Hd]: w2 as the forward saw it; [Hd]: b2
    int variant, gen_mode, out_act, B, R, Hd;
    float hyper[8];
    float inv_b;
    const float* pen;                           // optional penalty rows (x rows)
    float* S; float* dS; float* rowloss;        // optional [R] outputs (written by head workgroup 0)
    int enabled;
};

// score, d loss / d (pre-activation score) and loss term of row r from the forward's partial dots.
// A row's partials are CONTIGUOUS (part[r * ldp + j], ldp % 4 == 0, unused slots zero, at most 16): four
// independent 16-byte loads, one memory round trip -- a loop of dependent-in-order scalar loads cost 13
// serial trips to the fabric per row when the partials were fresh from another XCD (measured: the folded
// step was 3 us SLOWER than the unfolded one until this changed).
// (round 6: the loads and the arithmetic are separate functions -- the folded weight gradient requests its rows' partial
// dots BEFORE its first operand loads: vector-memory results return in issue order, and behind 80 KB of operand
// fragments per CU the partial dots landed ~1 us later than they had to)
static __device__ __forceinline__ void fold_part_load(const FoldP& f, int r, float4 (&v)[4]) {
    const float4* q = reinterpret_cast<const float4*>(f.part + (int64_t)r * f.ldp);
    const int n4 = (f.nparts + 3) >> 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = q[min(j, n4 - 1)];
}
static __device__ __forceinline__ float fold_score_of(const FoldP& f, const float4 (&v)[4]) {
    const int n4 = (f.nparts + 3) >> 2;
    const float bias = f.snap[f.Hd];
    float a2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < n4) { a2 += v[j].x; a2 += v[j].y; a2 += v[j].z; a2 += v[j].w; }
    a2 += bias;
    float s = a2;
    if (f.out_act == GM_ACT_SIGMOID) s = gm_sigmoid(a2);
    else if (f.out_act == GM_ACT_RELU) s = fmaxf(a2, 0.f);
    return s;
}
static __device__ __forceinline__ float fold_score(const FoldP& f, int r) {
    float4 v[4];
    fold_part_load(f, r, v);
    return fold_score_of(f, v);
}

static __device__ __forceinline__ void fold_row_of(const FoldP& f, int r, float s_in, float& s, float& ds, float& l);
static __device__ __forceinline__ void fold_row(const FoldP& f, int r, float& s, float& ds, float& l) {
    fold_row_of(f, r, fold_score(f, r), s, ds, l);
}
static __device__ __forceinline__ void fold_row_of(const FoldP& f, int r, float s_in, float& s, float& ds, float& l) {
    s = s_in;
    const bool D = !f.gen_mode;
    const bool is_x = D && r < f.B;
    // (round 6: evaluating only the row's own branch -- one log and one division instead of two -- measured 0.4 us
    // per step SLOWER than computing both on a dummy and selecting, same box, three alternations: 65.71 / 65.87 / 65.80
    // against 65.40 / 65.34 / 65.46 us; the variant switch in front of it costs more than the arithmetic it saves)
    float lx, lg, dx, dg;
    sample_terms(f.variant, D, is_x ? s : 0.5f, is_x ? 0.5f : s, f.inv_b, f.hyper, lx, lg, dx, dg);
    l = is_x ? lx : lg;
    const float dsel = is_x ? dx : dg;
    if (is_x && f.pen) l += f.hyper[7] * f.pen[r];
    ds = act_grad(dsel, s, f.out_act);
}

// dH of four consecutive columns from h, the row's dS and the columns' w2
static __device__ __forceinline__ float4 fold_dh4(float4 h, float ds, float4 w) {
    return make_float4((h.x > 0.f) ? ds * w.x : 0.f, (h.y > 0.f) ? ds * w.y : 0.f,
                       (h.z > 0.f) ? ds * w.z : 0.f, (h.w > 0.f) ? ds * w.w : 0.f);
}

constexpr int FOLD_MAX_ROWS = 2048;             // rows whose dS a workgroup keeps in LDS (8 KB)

struct HeadBwdP {
    const float* H; int64_t ldh;
    const float* dS; const float* w2; const float* rowloss;
    float* dH; int64_t lddh;
    float* gw2; float* gb2;                 // null in generator mode (D grads are not needed)
    float* loss_out; gm_slot loss_slot;
    float inv_b;
    int R, B, Hd, gen_mode;
    gm_adam_epi adam;                       // optional: Adam on (w2, b2) right here (pW=w2, pb=b2)
    int64_t* tick;                          // optional: *tick += 1 after the loss slot is written
    const float* gw2_add;                   // optional: added to gw2 before it is stored / stepped (the
                                            // gradient penalty's second-backward share, w_gp_gan.py:215)
    const float* pen_s; const float* pen_h; int64_t pen_ldh;   // optional: the gradient penalty's share of gw2,
    const float* pen_t; int64_t pen_ldt; int pen_rows;         // summed here (gm_hip.h gm_head_bwd_args)
    // (gm_head_bwd_args.gb2_add, DRAGAN: travels as pen_s with pen_rows == -1 -- the penalty-share members are
    // unused then, and one more member in this block cost the NSGAN step 0.1 - 0.3 us: round 4, same-box A/B)
    FoldP fold;                             // folded head: dS / rowloss come from fold_row, not from memory
};

// 16 columns x 64 row-groups per 1024-thread workgroup: 25 workgroups for Hd=400, 8 rows per thread
// at R=512 -- enough parallelism that the 800 KB read + 800 KB write is not a serial row walk.
constexpr int HB_COLS = 16, HB_RG = 64;

// bid: index of this workgroup among the head workgroups (0 .. gm_head_bwd_blocks()-1); 1024 threads.
// the workgroup's LDS copy of dS for rows [0, R): every thread strides over the rows
static __device__ __forceinline__ void fold_fill_lds(const FoldP& f, float* sds, int R) {
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        float s, ds, l;
        fold_row(f, r, s, ds, l);
        sds[r] = ds;
    }
    __syncthreads();
}
// the same with this thread's FIRST row's partial dots already requested (fold_part_load(f, threadIdx.x, v) if
// threadIdx.x < R): same arithmetic on the same values
static __device__ __forceinline__ void fold_fill_lds_pre(const FoldP& f, float* sds, int R, const float4 (&v)[4]) {
    int r = threadIdx.x;
    if (r < R) {
        float s, ds, l;
        fold_row_of(f, r, fold_score_of(f, v), s, ds, l);
        sds[r] = ds;
        r += blockDim.x;
    }
    for (; r < R; r += blockDim.x) {
        float s, ds, l;
        fold_row(f, r, s, ds, l);
        sds[r] = ds;
    }
    __syncthreads();
}

// ---- folded head, losses that are NOT a mean of per-row terms (round 4): RaGAN and Fisher critic steps ------------
// ra_gan.py:204-205 needs mean(D(G(z))) before any row's gradient and the sum of the x rows' du before the g rows';
// fisher_gan.py:214-223 needs four moments of the scores.  Every consumer workgroup already rebuilds EVERY row's score
// (it needs dS of all reduction rows), so the batch sums are block reductions in the same prologue: fp64, wave sums
// then the sixteen waves' partials in wave order -- the same bits in every workgroup and from run to run.  The
// expressions are gm_gan_loss's (gm_ops.hip), which stays the data-parallel form (its phases sit around scalar
// exchanges).  Fisher's lambda is read from aux[0] by every workgroup while head workgroup 0 computes its successor:
// that goes to aux[5] and is committed by a launch of its own (gm_fisher_commit) -- a workgroup that starts late must
// not see it.  f.pen carries the aux pointer (Fisher; the penalty rows it normally names do not exist here).
struct FoldTP { float loss, lam_next, m1x, m1g, m2x, m2g; };

template <int N>
static __device__ __forceinline__ void fold_block_sums(double (&v)[N], double (*sc)[16]) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = gm_wave_sum_d(v[k]);
    __syncthreads();                                          // sc: free again
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) sc[k][threadIdx.x >> 6] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double t = 0.0;
        for (int q = 0; q < 16; ++q) t += sc[k][q];
        v[k] = t;
    }
}

// 1024 threads; R <= FOLD_MAX_ROWS = 2048 rows: at most two per thread, scores kept in registers across the phases
static __device__ __forceinline__ void fold_fill_lds_tp(const FoldP& f, float* sds, int R, FoldTP& out) {
    __shared__ double sc[4][16];
    constexpr int Q = 2;
    const int B = f.B;
    const float ib = f.inv_b;
    float sv[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int r = threadIdx.x + q * 1024;
        sv[q] = fold_score(f, min(r, R - 1));
    }
    out = FoldTP{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (f.variant == GM_LOSS_RA) {
        double a[1] = {0.0};
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int r = threadIdx.x + q * 1024;
            if (r >= B && r < R) a[0] += (double)sv[q];
        }
        fold_block_sums<1>(a, sc);
        const float mg = (float)(a[0] * (double)ib);
        double b[2] = {0.0, 0.0};                             // loss terms; du of the x rows
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int r = threadIdx.x + q * 1024;
            if (r < B) {
                const float u = gm_sigmoid(sv[q] - mg);
                b[0] += (double)logf(u + EPS);
                const float gu = (-0.5f * ib) / (u + EPS);       // dL/du
                const float du = (gu * (1.f - u)) * u;           // through the inner sigmoid
                sds[r] = act_grad(du, sv[q], f.out_act);
                b[1] += (double)du;
            } else if (r < R) {
                b[0] += (double)logf(gm_sigmoid(1.f - sv[q]) + EPS);
            }
        }
        fold_block_sums<2>(b, sc);
        const float sum_du = (float)b[1];                     // d/dmg = -sum_du ; dmg/dsg_j = 1/B
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int r = threadIdx.x + q * 1024;
            if (r >= B && r < R) {
                const float v = gm_sigmoid(1.f - sv[q]);
                const float gv = (-0.5f * ib) / (v + EPS);
                const float dv = -((gv * (1.f - v)) * v);         // d(1 - sg)/dsg = -1
                sds[r] = act_grad(dv - sum_du * ib, sv[q], f.out_act);
            }
        }
        out.loss = -(float)(b[0] * (double)ib) / 2.f;
    } else {                                                  // GM_LOSS_FISHER
        double m[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int r = threadIdx.x + q * 1024;
            const float x = sv[q];
            if (r < B) { m[0] += x; m[1] += (double)(x * x); }
            else if (r < R) { m[2] += x; m[3] += (double)(x * x); }
        }
        fold_block_sums<4>(m, sc);
        const float m1x = (float)(m[0] * (double)ib), m2x = (float)(m[1] * (double)ib);
        const float m1g = (float)(m[2] * (double)ib), m2g = (float)(m[3] * (double)ib);
        const float lam = f.pen[0], rho = f.hyper[0];
        const float omega = 1.f - (0.5f * m2x + 0.5f * m2g);
        const float dO = -(lam - rho * omega);                // dL/dOmega
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int r = threadIdx.x + q * 1024;
            // dOmega/ds_i = -0.5 * 2 * s_i / B
            if (r < B) sds[r] = act_grad(-ib + dO * (-(sv[q] * ib)), sv[q], f.out_act);
            else if (r < R) sds[r] = act_grad(ib + dO * (-(sv[q] * ib)), sv[q], f.out_act);
        }
        out.loss = -((m1x - m1g) + lam * omega - (rho / 2.f) * (omega * omega));
        out.lam_next = lam + rho * (-omega);                  // lambda += rho * lambda.grad
        out.m1x = m1x; out.m1g = m1g; out.m2x = m2x; out.m2g = m2g;
    }
    __syncthreads();
}

// sds: folded head only -- LDS for the workgroup's copy of dS[0..R), filled HERE behind the first h loads
template <bool TP = false>
static __device__ __forceinline__ void head_bwd_body(const HeadBwdP& p, int bid, float* sds = nullptr) {
    FoldTP tp{};
    __shared__ float sh[HB_RG][HB_COLS + 1];
    __shared__ double shd[3][16];
    const int cl = threadIdx.x & (HB_COLS - 1), rg = threadIdx.x / HB_COLS;
    const int c = bid * HB_COLS + cl;
    const bool folded = p.fold.enabled != 0;
    // the step's schedule scalars: uniform, two dependent scalar loads (counter -> table row) -- issued
    // first so that the chain is over long before the Adam updates at the end need it
    float step_size = 0.f, bc2_sqrt = 1.f;
    if (p.adam.enabled) {
        const int64_t si = gm_slot_index(p.adam.sched_slot);
        step_size = p.adam.sched[2 * si]; bc2_sqrt = p.adam.sched[2 * si + 1];
    }
    float acc = 0.f;
    float pP = 0.f, pM = 0.f, pV = 0.f;                  // Adam state of w2[c], loaded behind the h loads
    const bool col_adam = p.gw2 && p.adam.enabled && rg == 0 && c < p.Hd;
    if (folded) {
        if (p.gw2) {                                     // kernel-argument uniform: every thread takes this path
            // Nothing to write per row (dH is formed in the GEMM's registers).  The first batch of a
            // thread's h loads goes out back to back, unconditionally (clamped addresses); BEHIND them
            // the workgroup rebuilds dS[0..R) from the forward's partial dots (one barrier, reached by
            // all 1024 threads at this one place), so both trips to the fabric overlap; then the dot
            // products in row order (same bits as the loop of the unfolded form).
            constexpr int U = 8;
            const bool active = c < p.Hd;
            const int cc = min(c, p.Hd - 1);
            float hv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) hv[u] = p.H[(int64_t)min(rg + u * HB_RG, p.R - 1) * p.ldh + cc];
            if (col_adam) { pP = p.adam.pW[c]; pM = p.adam.mW[c]; pV = p.adam.vW[c]; }
            if constexpr (TP) fold_fill_lds_tp(p.fold, sds, p.R, tp);
            else fold_fill_lds(p.fold, sds, p.R);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = rg + u * HB_RG;
                if (active && r < p.R) acc = fmaf(sds[r], hv[u], acc);
            }
            for (int r0 = rg + U * HB_RG; r0 < p.R; r0 += U * HB_RG) {       // R > 512
#pragma unroll
                for (int u = 0; u < U; ++u) hv[u] = p.H[(int64_t)min(r0 + u * HB_RG, p.R - 1) * p.ldh + cc];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int r = r0 + u * HB_RG;
                    if (active && r < p.R) acc = fmaf(sds[r], hv[u], acc);
                }
            }
        }
    } else if (c < p.Hd && (p.dH || p.gw2)) {
        const float w = p.w2[c];
        for (int r = rg; r < p.R; r += HB_RG) {
            const float h = p.H[(int64_t)r * p.ldh + c];
            const float d = p.dS[r];
            if (p.dH) p.dH[(int64_t)r * p.lddh + c] = (h > 0.f) ? d * w : 0.f;
            acc = fmaf(d, h, acc);
        }
        if (col_adam) { pP = p.adam.pW[c]; pM = p.adam.mW[c]; pV = p.adam.vW[c]; }
        if (p.pen_t) {                                       // kernel-argument uniform
            float a2 = 0.f;
            for (int r = rg; r < p.pen_rows; r += HB_RG) {
                const float sv = p.pen_s[r], hv = p.pen_h[(int64_t)r * p.pen_ldh + c];
                const float tv = p.pen_t[(int64_t)r * p.pen_ldt + c];
                if (sv > 0.f && hv > 0.f) a2 += tv;
            }
            acc += a2;
        }
    }
    if (p.gw2) {
        sh[rg][cl] = acc;
        __syncthreads();
        if (rg == 0 && c < p.Hd) {
            float v = 0.f;
            for (int q = 0; q < HB_RG; ++q) v += sh[q][cl];
            if (p.gw2_add) v += p.gw2_add[c];
            p.gw2[c] = v;
            if (p.adam.enabled) {          // every thread of this block read w2[c] before the barrier
                adam_update(pP, v, pM, pV, step_size, bc2_sqrt, p.adam.omb1, p.adam.b2, p.adam.omb2,
                            p.adam.eps, p.adam.wd, p.adam.clamp);
                p.adam.pW[c] = pP; p.adam.mW[c] = pM; p.adam.vW[c] = pV;
            }
        }
    }
    if (bid == 0) {
        // scalars: loss = inv_b * sum l_r ; gb2 = fl(sum over x rows) + fl(sum over g rows)
        float bP = 0.f, bM = 0.f, bV = 0.f;
        const bool b_adam = p.gb2 && p.adam.enabled && threadIdx.x == 0;
        if (b_adam) { bP = p.adam.pb[0]; bM = p.adam.mb[0]; bV = p.adam.vb[0]; }
        double sl = 0.0, sx = 0.0, sg = 0.0;
        for (int r = threadIdx.x; r < p.R; r += 1024) {
            double d;
            if (TP) {                                        // the workgroup's LDS copy is the final dS
                d = (double)sds[r];
                if (p.fold.S) p.fold.S[r] = fold_score(p.fold, r);
                if (p.fold.dS) p.fold.dS[r] = sds[r];
            } else if (folded) {
                float s, ds, l;
                fold_row(p.fold, r, s, ds, l);
                sl += (double)l;
                d = (double)ds;
                if (p.fold.S) p.fold.S[r] = s;             // observability: nothing on the path reads these
                if (p.fold.dS) p.fold.dS[r] = ds;
                if (p.fold.rowloss) p.fold.rowloss[r] = l;
            } else {
                sl += (double)p.rowloss[r];
                d = (double)p.dS[r];
            }
            if (!p.gen_mode && r < p.B) sx += d; else sg += d;
        }
        // the three block sums together: wave sums, one barrier, 16 partials each in wave order
        const double a0 = gm_wave_sum_d(sl), a1 = gm_wave_sum_d(sx), a2 = gm_wave_sum_d(sg);
        __syncthreads();                                   // (sh / shd reuse: everybody is past the column sums)
        if ((threadIdx.x & 63) == 0) {
            shd[0][threadIdx.x >> 6] = a0; shd[1][threadIdx.x >> 6] = a1; shd[2][threadIdx.x >> 6] = a2;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot[3] = {0.0, 0.0, 0.0};
            for (int k = 0; k < 3; ++k)
                for (int q = 0; q < 16; ++q) tot[k] += shd[k][q];
            p.loss_out[gm_slot_index(p.loss_slot)] = TP ? tp.loss : (float)(tot[0] * (double)p.inv_b);
            if (TP && p.fold.variant == GM_LOSS_FISHER) {      // aux: [0] lambda (committed by gm_fisher_commit from [5])
                float* aux = const_cast<float*>(p.fold.pen);
                aux[5] = tp.lam_next;
                aux[1] = tp.m1x; aux[2] = tp.m1g; aux[3] = tp.m2x; aux[4] = tp.m2g;
            }
            if (p.gb2) {
                float gb = (float)tot[1] + (float)tot[2];
                if (p.pen_rows < 0) gb += p.pen_s[0];      // gb2_add
                p.gb2[0] = gb;
                if (p.adam.enabled) {
                    adam_update(bP, gb, bM, bV, step_size, bc2_sqrt, p.adam.omb1, p.adam.b2, p.adam.omb2,
                                p.adam.eps, p.adam.wd, p.adam.clamp);
                    p.adam.pb[0] = bP; p.adam.mb[0] = bM; p.adam.vb[0] = bV;
                }
            }
            // the per-graph tick folded into this single-writer point: kernels later in the same
            // iteration address their slots with add - mul (engine), the next iteration sees ctr+1
            if (p.tick) *p.tick += 1;
        }
    }
}

static __global__ __launch_bounds__(1024) void head_bwd_kernel(HeadBwdP p) {
    __shared__ float sds[FOLD_MAX_ROWS];
    head_bwd_body(p, blockIdx.x, sds);
}

// number of head workgroups a launch needs (scalars only when neither dH nor gw2 is produced)
static inline int gm_head_bwd_blocks(const HeadBwdP& p) {
    return (p.dH || p.gw2) ? (p.Hd + HB_COLS - 1) / HB_COLS : 1;
}

// Host side: validate the public argument block and turn it into the kernel's parameter block.
static inline int gm_head_from_args(const gm_head_bwd_args& a, HeadBwdP* out,
                                    const gm_head_fold_args* fold = nullptr) {
    GM_CHECK_ARG(a.H && a.w2 && a.loss_out && a.B > 0 && a.Hd > 0);
    GM_CHECK_ARG(fold || (a.dS && a.rowloss));
    HeadBwdP p{};
    if (a.with_adam) {
        GM_CHECK_ARG(a.gw2 && a.gb2 && a.b2 && a.mW && a.vW && a.mb && a.vb && a.sched && !a.gen_mode);
        gm_adam_epi& e = p.adam;
        e.pW = a.w2; e.mW = a.mW; e.vW = a.vW; e.pb = a.b2; e.mb = a.mb; e.vb = a.vb;
        e.sched = a.sched; e.sched_slot = a.sched_slot; e.omb1 = (float)(1.0 - a.beta1);
        e.b2 = (float)a.beta2; e.omb2 = (float)(1.0 - a.beta2); e.eps = (float)a.eps;
        e.wd = (float)a.weight_decay; e.clamp = a.clamp; e.enabled = 1;
    }
    p.tick = a.tick;
    p.gw2_add = a.gw2_add;
    GM_CHECK_ARG(!a.pen_t || (a.pen_s && a.pen_h && a.gw2 && !fold && a.pen_rows > 0 && a.pen_ldh >= a.Hd &&
                              a.pen_ldt >= a.Hd));
    p.pen_s = a.pen_s; p.pen_h = a.pen_h; p.pen_ldh = a.pen_ldh; p.pen_t = a.pen_t; p.pen_ldt = a.pen_ldt;
    p.pen_rows = a.pen_rows;
    GM_CHECK_ARG(!a.gb2_add || (a.gb2 && !a.gen_mode && !a.pen_t));
    if (a.gb2_add) { p.pen_s = a.gb2_add; p.pen_rows = -1; }
    p.H = a.H; p.ldh = a.ldh; p.dS = a.dS; p.w2 = a.w2; p.rowloss = a.rowloss; p.dH = a.dH;
    p.lddh = a.lddh; p.gw2 = a.gw2; p.gb2 = a.gb2; p.loss_out = a.loss_out;
    p.loss_slot = a.loss_slot; p.inv_b = a.inv_b; p.gen_mode = a.gen_mode; p.B = a.B;
    p.R = a.gen_mode ? a.B : 2 * a.B; p.Hd = a.Hd;
    if (fold) {
        const gm_head_fold_args& g = *fold;
        GM_CHECK_ARG(g.part && g.snap && g.nparts > 0 && g.nparts <= 16 && g.ldp >= g.nparts && g.ldp % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(g.part) & 15) == 0 && g.n_hyper >= 0 && g.n_hyper <= 8);
        GM_CHECK_ARG(g.nparts == (a.Hd + 31) / 32);
        GM_CHECK_ARG(g.out_act >= GM_ACT_ID && g.out_act <= GM_ACT_SIGMOID);
        const bool two_phase = !a.gen_mode && (g.variant == GM_LOSS_RA || g.variant == GM_LOSS_FISHER);
        // RaGAN / Fisher critic steps: only as riders of the weight gradient (every workgroup holds all rows' scores)
        GM_CHECK_ARG(!two_phase || (a.gw2 && 2 * a.B <= FOLD_MAX_ROWS && (g.variant != GM_LOSS_FISHER || g.pen)));
        GM_CHECK_ARG(!a.dH);                       // nothing materialises dH in the folded form
        GM_CHECK_ARG(p.R <= FOLD_MAX_ROWS || !a.gw2);
        FoldP& f = p.fold;
        f.part = g.part; f.ldp = g.ldp; f.nparts = g.nparts; f.snap = g.snap;
        f.variant = g.variant; f.gen_mode = a.gen_mode; f.out_act = g.out_act; f.B = a.B; f.R = p.R;
        f.Hd = a.Hd; f.inv_b = a.inv_b; f.pen = g.pen;
        for (int i = 0; i < 8; ++i) f.hyper[i] = (i < g.n_hyper) ? g.hyper[i] : 0.f;
        f.S = g.S; f.dS = g.dS; f.rowloss = g.rowloss;
        f.enabled = two_phase ? 2 : 1;                  // 2: fold_fill_lds_tp instantiations
    }
    *out = p;
    return 0;
}
