// Backward half of the fused critic head (see gm_fused.hip for the forward half and the C-ABI):
// shared between its own launch (gm_head_bwd*) and the weight-gradient GEMM that can carry the
// head workgroups in its grid (gm_linear_bwd_dw_adam_head) -- the two are independent once
// head_fwd_loss has produced dH, so one launch does both.
#pragma once
#include "gm_common.h"

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
};

// 16 columns x 64 row-groups per 1024-thread workgroup: 25 workgroups for Hd=400, 8 rows per thread
// at R=512 -- enough parallelism that the 800 KB read + 800 KB write is not a serial row walk.
constexpr int HB_COLS = 16, HB_RG = 64;

// bid: index of this workgroup among the head workgroups (0 .. gm_head_bwd_blocks()-1); 1024 threads.
static __device__ __forceinline__ void head_bwd_body(const HeadBwdP& p, int bid) {
    __shared__ float sh[HB_RG][HB_COLS + 1];
    __shared__ double shd[16];
    const int cl = threadIdx.x & (HB_COLS - 1), rg = threadIdx.x / HB_COLS;
    const int c = bid * HB_COLS + cl;
    float acc = 0.f;
    if (c < p.Hd && (p.dH || p.gw2)) {
        const float w = p.w2[c];
        for (int r = rg; r < p.R; r += HB_RG) {
            const float h = p.H[(int64_t)r * p.ldh + c];
            const float d = p.dS[r];
            if (p.dH) p.dH[(int64_t)r * p.lddh + c] = (h > 0.f) ? d * w : 0.f;
            acc = fmaf(d, h, acc);
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
                const int64_t si = gm_slot_index(p.adam.sched_slot);
                float P = p.adam.pW[c], M = p.adam.mW[c], V = p.adam.vW[c];
                adam_update(P, v, M, V, p.adam.sched[2 * si], p.adam.sched[2 * si + 1], p.adam.omb1,
                            p.adam.b2, p.adam.omb2, p.adam.eps, p.adam.wd, p.adam.clamp);
                p.adam.pW[c] = P; p.adam.mW[c] = M; p.adam.vW[c] = V;
            }
        }
    }
    if (bid == 0) {
        // scalars: loss = inv_b * sum l_r ; gb2 = fl(sum over x rows) + fl(sum over g rows)
        double sl = 0.0, sx = 0.0, sg = 0.0;
        for (int r = threadIdx.x; r < p.R; r += 1024) {
            sl += (double)p.rowloss[r];
            const double d = (double)p.dS[r];
            if (!p.gen_mode && r < p.B) sx += d; else sg += d;
        }
        double v[3] = {sl, sx, sg};
        float outv[3];
        for (int k = 0; k < 3; ++k) {
            double a = gm_wave_sum_d(v[k]);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = a;
            __syncthreads();
            double tot = 0.0;
            for (int q = 0; q < 16; ++q) tot += shd[q];
            outv[k] = (k == 0) ? (float)(tot * (double)p.inv_b) : (float)tot;
        }
        if (threadIdx.x == 0) {
            p.loss_out[gm_slot_index(p.loss_slot)] = outv[0];
            if (p.gb2) {
                const float gb = outv[1] + outv[2];
                p.gb2[0] = gb;
                if (p.adam.enabled) {
                    const int64_t si = gm_slot_index(p.adam.sched_slot);
                    float P = p.adam.pb[0], M = p.adam.mb[0], V = p.adam.vb[0];
                    adam_update(P, gb, M, V, p.adam.sched[2 * si], p.adam.sched[2 * si + 1],
                                p.adam.omb1, p.adam.b2, p.adam.omb2, p.adam.eps, p.adam.wd,
                                p.adam.clamp);
                    p.adam.pb[0] = P; p.adam.mb[0] = M; p.adam.vb[0] = V;
                }
            }
            // the per-graph tick folded into this single-writer point: kernels later in the same
            // iteration address their slots with add - mul (engine), the next iteration sees ctr+1
            if (p.tick) *p.tick += 1;
        }
    }
}

static __global__ __launch_bounds__(1024) void head_bwd_kernel(HeadBwdP p) {
    head_bwd_body(p, blockIdx.x);
}

// number of head workgroups a launch needs (scalars only when neither dH nor gw2 is produced)
static inline int gm_head_bwd_blocks(const HeadBwdP& p) {
    return (p.dH || p.gw2) ? (p.Hd + HB_COLS - 1) / HB_COLS : 1;
}

// Host side: validate the public argument block and turn it into the kernel's parameter block.
static inline int gm_head_from_args(const gm_head_bwd_args& a, HeadBwdP* out) {
    GM_CHECK_ARG(a.H && a.dS && a.w2 && a.rowloss && a.loss_out && a.B > 0 && a.Hd > 0);
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
    p.H = a.H; p.ldh = a.ldh; p.dS = a.dS; p.w2 = a.w2; p.rowloss = a.rowloss; p.dH = a.dH;
    p.lddh = a.lddh; p.gw2 = a.gw2; p.gb2 = a.gb2; p.loss_out = a.loss_out;
    p.loss_slot = a.loss_slot; p.inv_b = a.inv_b; p.gen_mode = a.gen_mode; p.B = a.B;
    p.R = a.gen_mode ? a.B : 2 * a.B; p.Hd = a.Hd;
    *out = p;
    return 0;
}
