// gm_fused.hip -- variant-specific fused kernels around the GEMMs:
//   WGAN-GP: interpolate, gradient-path mask product, row-norm penalty + its gradient, second-
//            backward column reduction (w_gp_gan.py:195-218; hand-derived in SURVEY.md A.3)
//   VAE    : reparameterise + KL (+ its gradients), squared-error reconstruction loss + gradient,
//            reparameterisation backward (vae.py:100-106, 203, 212)
//   generic: deterministic finalisation of per-block partial sums.
// All are HBM/L2-bound elementwise or row/column reductions: coalesced float4 rows, wave64
// shuffle reductions, LDS only to combine the 4 waves of a workgroup.
#include "gm_common.h"

// ------------------------------------------------------------------------------------------
// K9  x_hat = eps*x + (1-eps)*G(z)            (w_gp_gan.py:197-201)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void interp_kernel(const float* __restrict__ eps, gm_slot eps_slot,
                                                    const float* __restrict__ x, int64_t ldx,
                                                    const float* __restrict__ g, int64_t ldg,
                                                    float* __restrict__ out, int64_t ldo, int B,
                                                    int I) {
    const float* e = eps + gm_slot_offset(eps_slot);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const float ev = e[b];
    const float om = 1.f - ev;
    for (int i = lane; i < I; i += 64)
        out[(int64_t)b * ldo + i] = ev * x[(int64_t)b * ldx + i] + om * g[(int64_t)b * ldg + i];
}

extern "C" int gm_interp(void* stream, const float* eps, gm_slot eps_slot, const float* x,
                         int64_t ldx, const float* g, int64_t ldg, float* out, int64_t ldo, int B,
                         int I) {
    GM_CHECK_ARG(eps && x && g && out && B > 0 && I > 0);
    hipLaunchKernelGGL(interp_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, eps,
                       eps_slot, x, ldx, g, ldg, out, ldo, B, I);
    GM_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------
// K10 prologue: u[b,n] = [s_b > 0] * [h[b,n] > 0] * w2[n]   (d relu(a2)/d a1 path, A.3)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gp_u_kernel(const float* __restrict__ s,
                                                  const float* __restrict__ h, int64_t ldh,
                                                  const float* __restrict__ w2,
                                                  float* __restrict__ u, int64_t ldu, int B, int H) {
    const int64_t n = (int64_t)B * H;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int b = (int)(i / H), c = (int)(i % H);
        u[(int64_t)b * ldu + c] = (s[b] > 0.f && h[(int64_t)b * ldh + c] > 0.f) ? w2[c] : 0.f;
    }
}

extern "C" int gm_gp_u(void* stream, const float* s, const float* h, int64_t ldh, const float* w2,
                       float* u, int64_t ldu, int B, int H) {
    GM_CHECK_ARG(s && h && w2 && u && B > 0 && H > 0);
    int blocks = (int)(((int64_t)B * H + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gp_u_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, s, h, ldh, w2,
                       u, ldu, B, H);
    GM_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------
// K11 per-row L2 norm, penalty term and gamma = d(lambda*mean((n-1)^2))/dg
//     pen[b] = (n_b - 1)^2 ;  gamma_b = lambda * inv_b * 2 (n_b - 1) * g_b / n_b  (0 when n_b = 0,
//     torch's norm sub-gradient).  One wave per row.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gp_norm_kernel(const float* __restrict__ g, int64_t ldg,
                                                     float* __restrict__ gam, int64_t ldm,
                                                     float* __restrict__ pen, float lambda,
                                                     float inv_b, float k, int B, int I) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const float* row = g + (int64_t)b * ldg;
    float ss = 0.f;
    for (int i = lane; i < I; i += 64) ss += row[i] * row[i];
    ss = gm_wave_sum(ss);
    const float n = sqrtf(ss);
    const float dn = n - k;
    if (lane == 0) pen[b] = dn * dn;
    const float coef = (n > 0.f) ? (lambda * (inv_b * (2.f * dn))) / n : 0.f;
    float* o = gam + (int64_t)b * ldm;
    for (int i = lane; i < I; i += 64) o[i] = row[i] * coef;
}

extern "C" int gm_gp_norm(void* stream, const float* g, int64_t ldg, float* gam, int64_t ldm,
                          float* pen, float lambda, float inv_b, float k, int B, int I) {
    GM_CHECK_ARG(g && gam && pen && B > 0 && I > 0);
    hipLaunchKernelGGL(gp_norm_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, g, ldg,
                       gam, ldm, pen, lambda, inv_b, k, B, I);
    GM_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------
// K12 tail: gw2[n] += sum_b [s_b>0][h[b,n]>0] * t[b,n]       (dP/dw2, A.3)
// 32 columns per workgroup, 8 row-groups, fixed-order LDS combine (deterministic).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gp_dw2_kernel(const float* __restrict__ s,
                                                    const float* __restrict__ h, int64_t ldh,
                                                    const float* __restrict__ t, int64_t ldt,
                                                    float* __restrict__ gw2, int B, int H) {
    __shared__ float sh[8][33];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
    float acc = 0.f;
    if (c < H)
        for (int b = rg; b < B; b += 8)
            if (s[b] > 0.f && h[(int64_t)b * ldh + c] > 0.f) acc += t[(int64_t)b * ldt + c];
    sh[rg][threadIdx.x & 31] = acc;
    __syncthreads();
    if (rg == 0 && c < H) {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) v += sh[r][threadIdx.x & 31];
        gw2[c] += v;
    }
}

extern "C" int gm_gp_dw2(void* stream, const float* s, const float* h, int64_t ldh, const float* t,
                         int64_t ldt, float* gw2, int B, int H) {
    GM_CHECK_ARG(s && h && t && gw2 && B > 0 && H > 0);
    hipLaunchKernelGGL(gp_dw2_kernel, dim3((H + 31) / 32), dim3(256), 0, (hipStream_t)stream, s, h,
                       ldh, t, ldt, gw2, B, H);
    GM_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------
// K14a VAE reparameterise + KL  (vae.py:100-106, 210-212).  ml = [mu | log_var] (B x 2Z).
//   z = mu + eps*exp(lv/2);  kl = sum 0.5*(mu^2 + exp(lv) - lv - 1)
//   kl gradient seeds: dml_kl = [mu | 0.5*(exp(lv) - 1)]
// Single workgroup (B*Z <= ~20k elements); kl written to kl_out[slot].
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vae_reparam_kernel(const float* __restrict__ ml, int64_t ldml,
                                                         const float* __restrict__ eps,
                                                         gm_slot eps_slot, float* __restrict__ z,
                                                         int64_t ldz, float* __restrict__ kl_out,
                                                         gm_slot kl_slot, int B, int Z) {
    __shared__ double sh[4];
    const float* e = eps + gm_slot_offset(eps_slot);
    double acc = 0.0;
    const int n = B * Z;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int b = i / Z, c = i % Z;
        const float mu = ml[(int64_t)b * ldml + c], lv = ml[(int64_t)b * ldml + Z + c];
        z[(int64_t)b * ldz + c] = mu + e[i] * expf(lv / 2.f);
        acc += (double)(0.5f * (((mu * mu) + expf(lv)) - lv - 1.f));
    }
    acc = gm_wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) kl_out[gm_slot_index(kl_slot)] = (float)((sh[0] + sh[1]) + (sh[2] + sh[3]));
}

extern "C" int gm_vae_reparam(void* stream, const float* ml, int64_t ldml, const float* eps,
                              gm_slot eps_slot, float* z, int64_t ldz, float* kl_out,
                              gm_slot kl_slot, int B, int Z) {
    GM_CHECK_ARG(ml && eps && z && kl_out && B > 0 && Z > 0 && ldml >= 2 * Z);
    hipLaunchKernelGGL(vae_reparam_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ml, ldml, eps,
                       eps_slot, z, ldz, kl_out, kl_slot, B, Z);
    GM_LAUNCH_RET();
}

// K14c reparameterisation backward: dml = [dz + mu | dz*eps*0.5*exp(lv/2) + 0.5*(exp(lv)-1)]
__global__ __launch_bounds__(256) void vae_reparam_bwd_kernel(const float* __restrict__ ml,
                                                             int64_t ldml,
                                                             const float* __restrict__ eps,
                                                             gm_slot eps_slot,
                                                             const float* __restrict__ dz, int64_t lddz,
                                                             float* __restrict__ dml, int64_t ldd,
                                                             int B, int Z) {
    const float* e = eps + gm_slot_offset(eps_slot);
    const int n = B * Z;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int b = i / Z, c = i % Z;
        const float mu = ml[(int64_t)b * ldml + c], lv = ml[(int64_t)b * ldml + Z + c];
        const float g = dz[(int64_t)b * lddz + c];
        dml[(int64_t)b * ldd + c] = g + 0.5f * (2.f * mu);
        // d/dlv [eps*exp(lv/2)] = eps*exp(lv/2)*0.5 ; d/dlv kl = 0.5*(exp(lv) - 1)
        dml[(int64_t)b * ldd + Z + c] = ((g * e[i]) * expf(lv / 2.f)) / 2.f + 0.5f * (expf(lv) - 1.f);
    }
}

extern "C" int gm_vae_reparam_bwd(void* stream, const float* ml, int64_t ldml, const float* eps,
                                  gm_slot eps_slot, const float* dz, int64_t lddz, float* dml,
                                  int64_t ldd, int B, int Z) {
    GM_CHECK_ARG(ml && eps && dz && dml && B > 0 && Z > 0);
    int blocks = (B * Z + 255) / 256;
    hipLaunchKernelGGL(vae_reparam_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ml,
                       ldml, eps, eps_slot, dz, lddz, dml, ldd, B, Z);
    GM_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------
// K14b reconstruction loss: recon = sum (x - xr)^2 (vae.py:203) and the gradient w.r.t. the
// decoder's PRE-sigmoid output: dA = 2*(xr - x) * xr*(1-xr)   [d/dxr (x-xr)^2 = -2(x-xr)]
// One wave per row; per-row partial sums in `partial[B]` (fp32), finalised in fixed order by
// gm_sum_finalize (deterministic, no atomics).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqerr_kernel(const float* __restrict__ x, int64_t ldx,
                                                   const float* __restrict__ xr, int64_t ldr,
                                                   float* __restrict__ dA, int64_t lda,
                                                   float* __restrict__ partial, int B, int I) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    float acc = 0.f;
    for (int i = lane; i < I; i += 64) {
        const float xv = x[(int64_t)b * ldx + i], r = xr[(int64_t)b * ldr + i];
        const float d = xv - r;
        acc += d * d;
        const float go = -(2.f * d);                       // PowBackward * SubBackward
        dA[(int64_t)b * lda + i] = (go * (1.f - r)) * r;   // SigmoidBackward
    }
    acc = gm_wave_sum(acc);
    if (lane == 0) partial[b] = acc;
}

extern "C" int gm_sqerr_sigmoid_bwd(void* stream, const float* x, int64_t ldx, const float* xr,
                                    int64_t ldr, float* dA, int64_t lda, float* partial, int B,
                                    int I) {
    GM_CHECK_ARG(x && xr && dA && partial && B > 0 && I > 0);
    hipLaunchKernelGGL(sqerr_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, xr,
                       ldr, dA, lda, partial, B, I);
    GM_LAUNCH_RET();
}

// out[slot] = scale * sum_{i<n} partial[i]   (single workgroup, fp64 accumulate, fixed order)
__global__ __launch_bounds__(256) void sum_finalize_kernel(const float* __restrict__ partial, int n,
                                                          float scale, float* __restrict__ out,
                                                          gm_slot out_slot) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += (double)partial[i];
    acc = gm_wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        out[gm_slot_index(out_slot)] = (float)(((sh[0] + sh[1]) + (sh[2] + sh[3])) * (double)scale);
}

extern "C" int gm_sum_finalize(void* stream, const float* partial, int n, float scale, float* out,
                               gm_slot out_slot) {
    GM_CHECK_ARG(partial && out && n > 0);
    hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n,
                       scale, out, out_slot);
    GM_LAUNCH_RET();
}
