// Shared helpers for the gfx950 kernels (internal; the public C-ABI is include/gm_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gm_hip.h"

extern "C" void gm_set_error(const char* msg);

#define GM_CHECK_ARG(cond)                                        \
    do {                                                          \
        if (!(cond)) {                                            \
            gm_set_error("bad argument: " #cond);                 \
            return GM_EINVAL;                                     \
        }                                                         \
    } while (0)

#define GM_LAUNCH_RET()                                           \
    do {                                                          \
        hipError_t e__ = hipGetLastError();                       \
        if (e__ != hipSuccess) {                                  \
            gm_set_error(hipGetErrorString(e__));                 \
            return -(int)e__;                                     \
        }                                                         \
        return 0;                                                 \
    } while (0)

// Device-side slot resolution: ((ctr ? *ctr : 0) * mul + add) % ring * stride.
__device__ __forceinline__ int64_t gm_slot_index(const gm_slot& s) {
    int64_t t = s.ctr ? *s.ctr : 0;
    int64_t i = t * (int64_t)s.mul + (int64_t)s.add;
    if (s.ring > 0) i %= (int64_t)s.ring;
    return i;
}
__device__ __forceinline__ int64_t gm_slot_offset(const gm_slot& s) {
    return gm_slot_index(s) * s.stride;
}

// wave64 reductions
__device__ __forceinline__ float gm_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double gm_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float gm_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// Sum of n floats by one 1024-thread workgroup, this thread's share: fp64, fixed order, loads in batches of 8
// INDEPENDENT ones (a row-tile partial array of the fused reconstruction loss has 14 336 entries at B = 512).
__device__ __forceinline__ double gm_strided_sum_1024(const float* __restrict__ p, int n) {
    double acc = 0.0;
    for (int i0 = threadIdx.x; i0 < n; i0 += 1024 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[min(i0 + u * 1024, n - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * 1024 < n) acc += (double)v[u];
    }
    return acc;
}

// The two sums that close a VAE batch (vae.py:203, :212), by one 1024-thread workgroup; sh: 32 doubles of LDS.
struct gm_fin2 {
    const float* pa; int na; float sa; float* oa; gm_slot slot_a;
    const float* pb; int nb; float sb; float* ob; gm_slot slot_b;
    int64_t* tick;                 // device step counter to advance (or null)
    unsigned int* done;            // arrival counter of the launch the sums ride in (gemm16_dw_pair_fin_kernel)
};
__device__ __forceinline__ void gm_fin2_sums(const gm_fin2& f, double* sh) {
    double a = gm_strided_sum_1024(f.pa, f.na), b = gm_strided_sum_1024(f.pb, f.nb);
    a = gm_wave_sum_d(a); b = gm_wave_sum_d(b);
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = a; sh[16 + (threadIdx.x >> 6)] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, tb = 0.0;
        for (int w = 0; w < 16; ++w) { ta += sh[w]; tb += sh[16 + w]; }
        f.oa[gm_slot_index(f.slot_a)] = (float)(ta * (double)f.sa);
        f.ob[gm_slot_index(f.slot_b)] = (float)(tb * (double)f.sb);
    }
}

// z = mu + eps * exp(log_var / 2) (vae.py:100-106) with the product rounded before the add, as torch's
// `mu + eps * std` does; pinned (fp contract off) so that the two places that compute it -- the reparameterisation
// workgroups, which store z for the backward pass, and the decoder's first-layer GEMM, which forms its A operand from
// (mu, log_var, eps) on the fly (gm_vae_reparam_fwd) -- agree bit for bit.
__device__ __forceinline__ float gm_reparam_z(float mu, float e, float lv) {
#pragma clang fp contract(off)
    const float s = expf(lv / 2.f);
    const float t = e * s;
    return mu + t;
}

// eps * x + (1 - eps) * g as torch computes it (w_gp_gan.py:197-201): two rounded products and an add.  The pragma is
// what keeps hipcc (-ffp-contract=fast) from fusing one product into the add wherever this gets inlined; HIP's
// __fmul_rn / __fadd_rn are plain operators and do not.
__device__ __forceinline__ float gm_interp_unfused(float ev, float x, float g) {
#pragma clang fp contract(off)
    const float a = ev * x;
    const float om = 1.f - ev;
    const float b = om * g;
    return a + b;
}

// ---- shared by the loss kernels (gm_ops.hip) and the fused critic-head kernels (gm_fused.hip) ----
constexpr float EPS = 1e-8f;

static __device__ __forceinline__ float act_grad(float g, float s, int out_act) {
    if (out_act == GM_ACT_SIGMOID) return (g * (1.f - s)) * s;   // SigmoidBackward: grad*(1-y)*y
    if (out_act == GM_ACT_RELU) return s > 0.f ? g : 0.f;
    return g;
}

// Per-sample loss terms and d loss / d score for the separable variants (appendix A.2):
// lx/dx for a real-sample score x, lg/dg for a generated-sample score g; `D` = critic mode.
static __device__ __forceinline__ void sample_terms(int variant, bool D, float x, float g, float ib,
                                             const float* hyper, float& lx, float& lg, float& dx,
                                             float& dg) {
    lx = lg = dx = dg = 0.f;
    switch (variant) {
    case GM_LOSS_NS:
        if (D) {   // ns_gan.py:191-192
            const float ux = x + EPS, ug = (1.f - g) + EPS;
            lx = -logf(ux); lg = -logf(ug);
            dx = (-ib) / ux; dg = -((-ib) / ug);
        } else {   // ns_gan.py:214
            const float ug = g + EPS;
            lg = -logf(ug); dg = (-ib) / ug;
        }
        break;
    case GM_LOSS_MM:
        if (D) {
            const float ux = x + EPS, ug = (1.f - g) + EPS;
            lx = -logf(ux); lg = -logf(ug);
            dx = (-ib) / ux; dg = -((-ib) / ug);
        } else {   // mm_gan.py:235
            const float ug = (1.f - g) + EPS;
            lg = logf(ug); dg = -(ib / ug);
        }
        break;
    case GM_LOSS_W:
    case GM_LOSS_FISHER:   // generator mode only reaches here: -mean(sg)
        if (D) { lx = -x; lg = g; dx = -ib; dg = ib;
 }
        else   { lg = -g; dg = -ib; }
        break;
    case GM_LOSS_LS: {
        const float a = hyper[0], b = hyper[1], c = hyper[2];
        if (D) {   // ls_gan.py:192-193
            lx = 0.5f * ((x - b) * (x - b)); lg = 0.5f * ((g - a) * (g - a));
            dx = (0.5f * ib) * (2.f * (x - b)); dg = (0.5f * ib) * (2.f * (g - a));
        } else {   // ls_gan.py:213
            lg = 0.5f * ((g - c) * (g - c)); dg = (0.5f * ib) * (2.f * (g - c));
        }
        break;
    }
    case GM_LOSS_RA:       // generator mode: plain NS (ra_gan.py:227)
    {
        const float ug = g + EPS;
        lg = -logf(ug); dg = (-ib) / ug;
        break;
    }
    case GM_LOSS_F_TV: {
        const float tg = tanhf(g);
        if (D) { const float tx = tanhf(x);
                 lx = -(0.5f * tx); lg = 0.5f * tg;
                 dx = -(0.5f * ib) * (1.f - tx * tx); dg = (0.5f * ib) * (1.f - tg * tg); }
        else   { lg = -(0.5f * tg); dg = -(0.5f * ib) * (1.f - tg * tg); }
        break;
    }
    case GM_LOSS_F_FKL: {
        const float e = expf(g - 1.f);
        if (D) { lx = -x; lg = e; dx = -ib; dg = ib * e; }
        else   { lg = -e; dg = -(ib * e); }
        break;
    }
    case GM_LOSS_F_RKL:
        if (D) { const float e = expf(x); lx = e; lg = -1.f - g; dx = ib * e; dg = -ib; }
        else   { lg = -(-1.f - g); dg = ib; }
        break;
    case GM_LOSS_F_PEARSON: {
        const float q = 0.25f * (g * g) + g;
        if (D) { lx = -x; lg = q; dx = -ib; dg = ib * (0.5f * g + 1.f); }
        else   { lg = -q; dg = -(ib * (0.5f * g + 1.f)); }
        break;
    }
    case GM_LOSS_F_HELLINGER: {
        const float eg = expf(g);
        const float h = (1.f - eg) / eg;                    // = exp(-g) - 1
        if (D) { const float ex = expf(x);
                 lx = -(1.f - ex); lg = h; dx = ib * ex; dg = -(ib / eg); }
        else   { lg = -h; dg = ib / eg; }
        break;
    }
    case GM_LOSS_F_JS: {
        const float eg = expf(g);
        if (D) { const float enx = expf(-x);
                 lx = -(2.f - (1.f + enx)); lg = -(2.f - eg);
                 dx = -(ib * enx); dg = ib * eg; }
        else   { lg = 2.f - eg; dg = -(ib * eg); }
        break;
    }
    default: break;
    }
}


// ---- Adam arithmetic shared by adam_kernel and the fused dW / head epilogues ----
struct gm_adam_epi {          // optional "optimizer in the gradient epilogue" descriptor
    float* pW; float* mW; float* vW;      // parameter / moments of the weight the GEMM produces dW for
    float* pb; float* mb; float* vb;      // same for the bias (db)
    const float* sched; gm_slot sched_slot;
    float omb1, b2, omb2, eps, wd, clamp;
    int enabled;
};

static __device__ __forceinline__ void adam_update(float& pp, float gg, float& mm, float& vv,
                                                    float step_size, float bc2_sqrt, float omb1,
                                                    float b2, float omb2, float eps, float wd,
                                                    float clamp) {
    // Every rounding is pinned (round 4): left to -ffp-contract=fast, the same source fused differently depending on
    // what it was inlined into (the float2 epilogue and the element-wise one of gm_gemm.hip disagreed in the last bit
    // of exp_avg / exp_avg_sq; HIP's __fmul_rn / __fadd_rn are plain operators and do NOT stop the fusion -- the pragma
    // does).  The sequence is torch's _single_tensor_adam on CPU: lerp_ is ONE fused multiply-add (at::vec::fmadd,
    // weight < 0.5), mul_ / addcmul_ / sqrt / div / add_ / addcdiv_ round after every operation.
#pragma clang fp contract(off)
    if (wd != 0.f) gg = __builtin_fmaf(wd, pp, gg);
    mm = __builtin_fmaf(omb1, gg - mm, mm);
    vv = vv * b2;
    const float g2 = (omb2 * gg) * gg;
    vv = vv + g2;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    const float upd = ((-step_size) * mm) / denom;
    pp = pp + upd;
    if (clamp > 0.f) pp = fminf(fmaxf(pp, -clamp), clamp);
}
