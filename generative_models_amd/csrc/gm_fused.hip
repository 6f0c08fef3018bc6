// gm_fused.hip -- variant-specific fused kernels around the GEMMs:
//   WGAN-GP: interpolate, gradient-path mask product, row-norm penalty + its gradient, second-
//            backward column reduction (w_gp_gan.py:195-218; hand-derived in SURVEY.md A.3)
//   VAE    : reparameterise + KL (+ its gradients), squared-error reconstruction loss + gradient,
//            reparameterisation backward (vae.py:100-106, 203, 212)
//   generic: deterministic finalisation of per-block partial sums.
// All are HBM/L2-bound elementwise or row/column reductions: coalesced float4 rows, wave64
// shuffle reductions, LDS only to combine the 4 waves of a workgroup.
#include "gm_common.h"
#include "gm_head.h"

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
    for (int i = lane; i < I; i += 64)      // two roundings and an add, never contracted (as torch computes it)
        out[(int64_t)b * ldo + i] = gm_interp_unfused(ev, x[(int64_t)b * ldx + i], g[(int64_t)b * ldg + i]);
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
template <bool VEC4>
__global__ __launch_bounds__(64) void gp_norm_kernel(const float* __restrict__ g, int64_t ldg,
                                                    float* __restrict__ gam, int64_t ldm,
                                                    float* __restrict__ pen, float lambda,
                                                    float inv_b, float k, int B, int I) {
    // one wave per row AND per workgroup (B workgroups spread over the CUs: a latency chain, not
    // bandwidth); VEC4 (I % 4 == 0, I <= 1024, aligned rows): the row stays in registers between the
    // norm and the scaling -- one round of 16-byte loads instead of two passes of dword loads
    const int lane = threadIdx.x, b = blockIdx.x;
    const float* row = g + (int64_t)b * ldg;
    float* o = gam + (int64_t)b * ldm;
    float ss = 0.f;
    float4 v[4];
    if (VEC4) {
        const int n4 = I >> 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = reinterpret_cast<const float4*>(row)[min(lane + 64 * j, n4 - 1)];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (lane + 64 * j < n4)
                ss += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
    } else {
        for (int i = lane; i < I; i += 64) ss += row[i] * row[i];
    }
    ss = gm_wave_sum(ss);
    const float n = sqrtf(ss);
    const float dn = n - k;
    if (lane == 0) pen[b] = dn * dn;
    const float coef = (n > 0.f) ? (lambda * (inv_b * (2.f * dn))) / n : 0.f;
    if (VEC4) {
        const int n4 = I >> 2;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (lane + 64 * j < n4)
                reinterpret_cast<float4*>(o)[lane + 64 * j] =
                    make_float4(v[j].x * coef, v[j].y * coef, v[j].z * coef, v[j].w * coef);
    } else {
        for (int i = lane; i < I; i += 64) o[i] = row[i] * coef;
    }
}

extern "C" int gm_gp_norm(void* stream, const float* g, int64_t ldg, float* gam, int64_t ldm,
                          float* pen, float lambda, float inv_b, float k, int B, int I) {
    GM_CHECK_ARG(g && gam && pen && B > 0 && I > 0);
    const bool vec4 = (I % 4 == 0) && I <= 1024 && (ldg % 4 == 0) && (ldm % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(gam)) & 15) == 0;
    if (vec4) hipLaunchKernelGGL(gp_norm_kernel<true>, dim3(B), dim3(64), 0, (hipStream_t)stream, g, ldg,
                                 gam, ldm, pen, lambda, inv_b, k, B, I);
    else hipLaunchKernelGGL(gp_norm_kernel<false>, dim3(B), dim3(64), 0, (hipStream_t)stream, g, ldg,
                            gam, ldm, pen, lambda, inv_b, k, B, I);
    GM_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------
// K12 tail: gw2[n] += sum_b [s_b>0][h[b,n]>0] * t[b,n]       (dP/dw2, A.3)
// 32 columns per workgroup, 32 row-groups (1024 threads: 8 rows each at B = 256 -- with 8 row-groups a
// thread walked 32 dependent rows and the launch took 9.5 us), fixed-order LDS combine (deterministic).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void gp_dw2_kernel(const float* __restrict__ s,
                                                     const float* __restrict__ h, int64_t ldh,
                                                     const float* __restrict__ t, int64_t ldt,
                                                     float* __restrict__ gw2, int B, int H, int store) {
    __shared__ float sh[32][33];
    const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + col;
    float acc = 0.f;
    if (c < H) {
        int b = rg;
        for (; b + 96 < B; b += 128) {                       // four independent rows in flight
            const float s0 = s[b], s1 = s[b + 32], s2 = s[b + 64], s3 = s[b + 96];
            const float h0 = h[(int64_t)b * ldh + c], h1 = h[(int64_t)(b + 32) * ldh + c];
            const float h2 = h[(int64_t)(b + 64) * ldh + c], h3 = h[(int64_t)(b + 96) * ldh + c];
            const float t0 = t[(int64_t)b * ldt + c], t1 = t[(int64_t)(b + 32) * ldt + c];
            const float t2 = t[(int64_t)(b + 64) * ldt + c], t3 = t[(int64_t)(b + 96) * ldt + c];
            if (s0 > 0.f && h0 > 0.f) acc += t0;
            if (s1 > 0.f && h1 > 0.f) acc += t1;
            if (s2 > 0.f && h2 > 0.f) acc += t2;
            if (s3 > 0.f && h3 > 0.f) acc += t3;
        }
        for (; b < B; b += 32)
            if (s[b] > 0.f && h[(int64_t)b * ldh + c] > 0.f) acc += t[(int64_t)b * ldt + c];
    }
    sh[rg][col] = acc;
    __syncthreads();
    if (rg == 0 && c < H) {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) v += sh[r][col];
        gw2[c] = store ? v : (gw2[c] + v);
    }
}

extern "C" int gm_gp_dw2(void* stream, const float* s, const float* h, int64_t ldh, const float* t,
                         int64_t ldt, float* gw2, int B, int H) {
    GM_CHECK_ARG(s && h && t && gw2 && B > 0 && H > 0);
    hipLaunchKernelGGL(gp_dw2_kernel, dim3((H + 31) / 32), dim3(1024), 0, (hipStream_t)stream, s, h,
                       ldh, t, ldt, gw2, B, H, 0);
    GM_LAUNCH_RET();
}
extern "C" int gm_gp_dw2_store(void* stream, const float* s, const float* h, int64_t ldh, const float* t,
                               int64_t ldt, float* out, int B, int H) {
    GM_CHECK_ARG(s && h && t && out && B > 0 && H > 0);
    hipLaunchKernelGGL(gp_dw2_kernel, dim3((H + 31) / 32), dim3(1024), 0, (hipStream_t)stream, s, h,
                       ldh, t, ldt, out, B, H, 1);
    GM_LAUNCH_RET();
}

// One wave per row (and per workgroup): the N = 1 critic layer on x_hat (a dot product, not an MFMA
// launch) and u.  VEC4: h and w2 rows in registers (16-byte loads), u written from them.
template <bool VEC4>
__global__ __launch_bounds__(64) void head_gp_kernel(const float* __restrict__ h, int64_t ldh,
                                                    const float* __restrict__ w2,
                                                    const float* __restrict__ b2, float* __restrict__ s,
                                                    float* __restrict__ u, int64_t ldu, int B, int H) {
    const int lane = threadIdx.x, b = blockIdx.x;
    const float* hr = h + (int64_t)b * ldh;
    float* ur = u + (int64_t)b * ldu;
    float acc = 0.f;
    float4 hv[2], wv[2];
    if (VEC4) {
        const int n4 = H >> 2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i4 = min(lane + 64 * j, n4 - 1);
            hv[j] = reinterpret_cast<const float4*>(hr)[i4];
            wv[j] = reinterpret_cast<const float4*>(w2)[i4];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (lane + 64 * j < n4) {
                acc = fmaf(hv[j].x, wv[j].x, acc); acc = fmaf(hv[j].y, wv[j].y, acc);
                acc = fmaf(hv[j].z, wv[j].z, acc); acc = fmaf(hv[j].w, wv[j].w, acc);
            }
    } else {
        for (int i = lane; i < H; i += 64) acc = fmaf(hr[i], w2[i], acc);
    }
    acc = gm_wave_sum(acc);
    const float sv = fmaxf(acc + b2[0], 0.f);
    if (lane == 0) s[b] = sv;
    if (VEC4) {
        const int n4 = H >> 2;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (lane + 64 * j < n4) {
                float4 d;
                d.x = (sv > 0.f && hv[j].x > 0.f) ? wv[j].x : 0.f; d.y = (sv > 0.f && hv[j].y > 0.f) ? wv[j].y : 0.f;
                d.z = (sv > 0.f && hv[j].z > 0.f) ? wv[j].z : 0.f; d.w = (sv > 0.f && hv[j].w > 0.f) ? wv[j].w : 0.f;
                reinterpret_cast<float4*>(ur)[lane + 64 * j] = d;
            }
    } else {
        for (int i = lane; i < H; i += 64) ur[i] = (sv > 0.f && hr[i] > 0.f) ? w2[i] : 0.f;
    }
}
extern "C" int gm_head_gp(void* stream, const float* h, int64_t ldh, const float* w2, const float* b2,
                          float* s, float* u, int64_t ldu, int B, int H) {
    GM_CHECK_ARG(h && w2 && b2 && s && u && B > 0 && H > 0 && ldh >= H && ldu >= H);
    const bool vec4 = (H % 4 == 0) && H <= 512 && (ldh % 4 == 0) && (ldu % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(w2) |
                        reinterpret_cast<uintptr_t>(u)) & 15) == 0;
    if (vec4) hipLaunchKernelGGL(head_gp_kernel<true>, dim3(B), dim3(64), 0, (hipStream_t)stream, h, ldh, w2,
                                 b2, s, u, ldu, B, H);
    else hipLaunchKernelGGL(head_gp_kernel<false>, dim3(B), dim3(64), 0, (hipStream_t)stream, h, ldh, w2,
                            b2, s, u, ldu, B, H);
    GM_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------
// BIR-VAE (bir_vae.py:86-97, 180-221).
//   reparameterize: z = mu + eps, eps ~ N(0, set_var) drawn by the HOST from numpy's global RNG
//   maximum_mean_discrepancy: k(a,b) = exp(-mean_d((a-b)^2)/dim) = exp(-|a-b|^2/dim^2);
//     mmd = sum_ij k(x_i,x_j) + sum_ij k(z_i,z_j) - 2 sum_ij k(x_i,z_j),  x = torch.randn(z.shape)
//   One wave per latent row m: partial[m] = sum_j k(x_m,x_j) + sum_j k(z_m,z_j) - 2 sum_i k(x_i,z_m)
//   and d(lambda*mmd)/dz_m = lambda*(4/dim^2) * [sum_i k(x_i,z_m)(z_m-x_i) - sum_j k(z_m,z_j)(z_m-z_j)]
//   (the z-z term appears twice in the double sum: once through each argument).  O(B^2 Z) work,
//   26 MFLOP at B=256, Z=20: a latency-sized kernel, not a GEMM.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bir_reparam_kernel(const float* __restrict__ mu, int64_t ldmu,
                                                         const float* __restrict__ eps, gm_slot eps_slot,
                                                         float* __restrict__ z, int64_t ldz, int B, int Z) {
    const float* e = eps + gm_slot_offset(eps_slot);
    const int64_t n = (int64_t)B * Z, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int b = (int)(i / Z), c = (int)(i % Z);
        z[(int64_t)b * ldz + c] = mu[(int64_t)b * ldmu + c] + e[i];
    }
}
extern "C" int gm_bir_reparam(void* stream, const float* mu, int64_t ldmu, const float* eps,
                              gm_slot eps_slot, float* z, int64_t ldz, int B, int Z) {
    GM_CHECK_ARG(mu && eps && z && B > 0 && Z > 0 && ldmu >= Z && ldz >= Z);
    int blocks = (int)(((int64_t)B * Z + 255) / 256);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(bir_reparam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mu, ldmu, eps,
                       eps_slot, z, ldz, B, Z);
    GM_LAUNCH_RET();
}

constexpr int BIR_MAXZ = 64;
__global__ __launch_bounds__(256) void bir_mmd_kernel(const float* __restrict__ z, int64_t ldz,
                                                     const float* __restrict__ prior, gm_slot prior_slot,
                                                     float* __restrict__ partial, float* __restrict__ dz,
                                                     int64_t lddz, int B, int Z, float lambda) {
    const float* x = prior + gm_slot_offset(prior_slot);          // [B, Z] contiguous
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + wave;
    if (m >= B) return;
    const float inv_d = 1.0f / (float)Z;
    float zm[BIR_MAXZ], xm[BIR_MAXZ], g[BIR_MAXZ];
#pragma unroll 4
    for (int d = 0; d < Z; ++d) { zm[d] = z[(int64_t)m * ldz + d]; xm[d] = x[(int64_t)m * Z + d]; g[d] = 0.f; }
    float s_xx = 0.f, s_zz = 0.f, s_xz = 0.f;
    for (int j = lane; j < B; j += 64) {
        const float* zj = z + (int64_t)j * ldz;
        const float* xj = x + (int64_t)j * Z;
        float dxx = 0.f, dzz = 0.f, dxz = 0.f;
        for (int d = 0; d < Z; ++d) {
            const float a = xm[d] - xj[d], b = zm[d] - zj[d], c = xj[d] - zm[d];
            dxx += a * a; dzz += b * b; dxz += c * c;
        }
        // torch: mean over dim (sum * 1/dim ... as a division), then div by dim, exp(-.)
        const float kxx = expf(-((dxx / (float)Z) / (float)Z));
        const float kzz = expf(-((dzz / (float)Z) / (float)Z));
        const float kxz = expf(-((dxz / (float)Z) / (float)Z));
        s_xx += kxx; s_zz += kzz; s_xz += kxz;
        if (dz) {
            for (int d = 0; d < Z; ++d) g[d] += kxz * (zm[d] - xj[d]) - kzz * (zm[d] - zj[d]);
        }
    }
    (void)inv_d;
    // the three sums are O(B) each and nearly cancel: combine them in double
    const double t_xx = gm_wave_sum_d((double)s_xx), t_zz = gm_wave_sum_d((double)s_zz),
                 t_xz = gm_wave_sum_d((double)s_xz);
    if (lane == 0) partial[m] = (float)((t_xx + t_zz) - 2.0 * t_xz);
    if (dz) {
        const float coef = lambda * (4.0f / ((float)Z * (float)Z));
        for (int d = 0; d < Z; ++d) {
            const float t = gm_wave_sum(g[d]);
            if (lane == 0) dz[(int64_t)m * lddz + d] = coef * t;
        }
    }
}
// The same arithmetic with the latent width a compile-time constant (ZC % 4 == 0, 16-byte aligned rows): the
// row vectors live in registers (the generic kernel's runtime-indexed arrays go to scratch: 54 us at B = 512,
// Z = 20 -- 40 % of a BIR-VAE step) and every row is read as ZC/4 16-byte loads.
template <int ZC, bool GRAD>
__global__ __launch_bounds__(256) void bir_mmd_z_kernel(const float* __restrict__ z, int64_t ldz,
                                                       const float* __restrict__ prior, gm_slot prior_slot,
                                                       float* __restrict__ partial, float* __restrict__ dz,
                                                       int64_t lddz, int B, float lambda) {
    constexpr int Q = ZC / 4;
    const float* x = prior + gm_slot_offset(prior_slot);          // [B, ZC] contiguous
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + wave;
    if (m >= B) return;
    float zm[ZC], xm[ZC], g[ZC];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(z + (int64_t)m * ldz + 4 * q);
        const float4 b = *reinterpret_cast<const float4*>(x + (int64_t)m * ZC + 4 * q);
        zm[4 * q] = a.x; zm[4 * q + 1] = a.y; zm[4 * q + 2] = a.z; zm[4 * q + 3] = a.w;
        xm[4 * q] = b.x; xm[4 * q + 1] = b.y; xm[4 * q + 2] = b.z; xm[4 * q + 3] = b.w;
    }
#pragma unroll
    for (int d = 0; d < ZC; ++d) g[d] = 0.f;
    float s_xx = 0.f, s_zz = 0.f, s_xz = 0.f;
    for (int j = lane; j < B; j += 64) {
        float zj[ZC], xj[ZC];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const float4 a = *reinterpret_cast<const float4*>(z + (int64_t)j * ldz + 4 * q);
            const float4 b = *reinterpret_cast<const float4*>(x + (int64_t)j * ZC + 4 * q);
            zj[4 * q] = a.x; zj[4 * q + 1] = a.y; zj[4 * q + 2] = a.z; zj[4 * q + 3] = a.w;
            xj[4 * q] = b.x; xj[4 * q + 1] = b.y; xj[4 * q + 2] = b.z; xj[4 * q + 3] = b.w;
        }
        float dxx = 0.f, dzz = 0.f, dxz = 0.f;
#pragma unroll
        for (int d = 0; d < ZC; ++d) {
            const float a = xm[d] - xj[d], b = zm[d] - zj[d], c = xj[d] - zm[d];
            dxx += a * a; dzz += b * b; dxz += c * c;
        }
        const float kxx = expf(-((dxx / (float)ZC) / (float)ZC));
        const float kzz = expf(-((dzz / (float)ZC) / (float)ZC));
        const float kxz = expf(-((dxz / (float)ZC) / (float)ZC));
        s_xx += kxx; s_zz += kzz; s_xz += kxz;
        if (GRAD) {
#pragma unroll
            for (int d = 0; d < ZC; ++d) g[d] += kxz * (zm[d] - xj[d]) - kzz * (zm[d] - zj[d]);
        }
    }
    const double t_xx = gm_wave_sum_d((double)s_xx), t_zz = gm_wave_sum_d((double)s_zz),
                 t_xz = gm_wave_sum_d((double)s_xz);
    if (lane == 0) partial[m] = (float)((t_xx + t_zz) - 2.0 * t_xz);
    if (GRAD) {
        const float coef = lambda * (4.0f / ((float)ZC * (float)ZC));
#pragma unroll
        for (int d = 0; d < ZC; ++d) {
            const float t = gm_wave_sum(g[d]);
            if (lane == 0) dz[(int64_t)m * lddz + d] = coef * t;
        }
    }
}
template <int ZC>
static void bir_mmd_z_launch(void* stream, const float* z, int64_t ldz, const float* prior, gm_slot prior_slot,
                             float* partial, float* dz, int64_t lddz, int B, float lambda) {
    if (dz) hipLaunchKernelGGL((bir_mmd_z_kernel<ZC, true>), dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                               z, ldz, prior, prior_slot, partial, dz, lddz, B, lambda);
    else hipLaunchKernelGGL((bir_mmd_z_kernel<ZC, false>), dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                            z, ldz, prior, prior_slot, partial, dz, lddz, B, lambda);
}
extern "C" int gm_bir_mmd(void* stream, const float* z, int64_t ldz, const float* prior, gm_slot prior_slot,
                          float* partial, float* dz, int64_t lddz, int B, int Z, float lambda) {
    GM_CHECK_ARG(z && prior && partial && B > 0 && Z > 0 && Z <= BIR_MAXZ && ldz >= Z && (!dz || lddz >= Z));
    // (the slot's element offset is a multiple of B*Z, so Z % 4 == 0 keeps every prior row 16-byte aligned)
    const bool al = (ldz % 4 == 0) && ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(prior)) & 15) == 0 &&
                    (prior_slot.stride % 4 == 0);
    if (al && (Z == 20 || Z == 8 || Z == 32)) {
        if (Z == 20) bir_mmd_z_launch<20>(stream, z, ldz, prior, prior_slot, partial, dz, lddz, B, lambda);
        else if (Z == 8) bir_mmd_z_launch<8>(stream, z, ldz, prior, prior_slot, partial, dz, lddz, B, lambda);
        else bir_mmd_z_launch<32>(stream, z, ldz, prior, prior_slot, partial, dz, lddz, B, lambda);
        GM_LAUNCH_RET();
    }
    hipLaunchKernelGGL(bir_mmd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, z, ldz, prior,
                       prior_slot, partial, dz, lddz, B, Z, lambda);
    GM_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------
// K14a VAE reparameterise + KL  (vae.py:100-106, 210-212).  ml = [mu | log_var] (B x 2Z).
//   z = mu + eps*exp(lv/2);  kl = sum 0.5*(mu^2 + exp(lv) - lv - 1)
//   kl gradient seeds: dml_kl = [mu | 0.5*(exp(lv) - 1)]
// Single workgroup (B*Z <= ~20k elements); kl written to kl_out[slot].
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void vae_reparam_kernel(const float* __restrict__ ml, int64_t ldml,
                                                          const float* __restrict__ eps,
                                                          gm_slot eps_slot, float* __restrict__ z,
                                                          int64_t ldz, float* __restrict__ kl_out,
                                                          gm_slot kl_slot, int B, int Z) {
    // one 1024-thread workgroup: two expf per element make this a latency chain, so 16 waves split it
    // (256 threads: 20.6 us at B*Z = 10240, a sixth of the VAE step; profiles/r02_vae_b512_summary.md)
    __shared__ double sh[16];
    const float* e = eps + gm_slot_offset(eps_slot);
    double acc = 0.0;
    const int n = B * Z;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const int b = i / Z, c = i % Z;
        const float mu = ml[(int64_t)b * ldml + c], lv = ml[(int64_t)b * ldml + Z + c];
        z[(int64_t)b * ldz + c] = gm_reparam_z(mu, e[i], lv);
        acc += (double)(0.5f * (((mu * mu) + expf(lv)) - lv - 1.f));
    }
    acc = gm_wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += sh[w];                 // fixed order
        kl_out[gm_slot_index(kl_slot)] = (float)t;
    }
}

extern "C" int gm_vae_reparam(void* stream, const float* ml, int64_t ldml, const float* eps,
                              gm_slot eps_slot, float* z, int64_t ldz, float* kl_out,
                              gm_slot kl_slot, int B, int Z) {
    GM_CHECK_ARG(ml && eps && z && kl_out && B > 0 && Z > 0 && ldml >= 2 * Z);
    hipLaunchKernelGGL(vae_reparam_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ml, ldml, eps,
                       eps_slot, z, ldz, kl_out, kl_slot, B, Z);
    GM_LAUNCH_RET();
}

// Wide form: 256-thread workgroups over the B*Z elements, per-WORKGROUP KL partials (fp64 inside, one
// float each, fixed order) -- the step's last launch (gm_sum_finalize2_tick) adds them up.  The
// one-workgroup form above costs 8.4 us at B*Z = 10 240; this one is a ~3 us launch.
__global__ __launch_bounds__(256) void vae_reparam_wide_kernel(const float* __restrict__ ml, int64_t ldml,
                                                              const float* __restrict__ eps, gm_slot eps_slot,
                                                              float* __restrict__ z, int64_t ldz,
                                                              float* __restrict__ kl_part, int B, int Z) {
    __shared__ double sh[4];
    const float* e = eps + gm_slot_offset(eps_slot);
    const int n = B * Z, i = blockIdx.x * 256 + threadIdx.x;
    double acc = 0.0;
    if (i < n) {
        const int b = i / Z, c = i % Z;
        const float mu = ml[(int64_t)b * ldml + c], lv = ml[(int64_t)b * ldml + Z + c];
        z[(int64_t)b * ldz + c] = gm_reparam_z(mu, e[i], lv);
        acc = (double)(0.5f * (((mu * mu) + expf(lv)) - lv - 1.f));
    }
    acc = gm_wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) kl_part[blockIdx.x] = (float)((sh[0] + sh[1]) + (sh[2] + sh[3]));
}

// ------------------------------------------------------------------------------------------
// Reparameterisation + the decoder's first layer as ONE launch (round 4; vae.py:100-106 + :113):
//   workgroups [0, nrp): exactly vae_reparam_wide_kernel's work for 256 elements each (z stored for the backward
//                        pass, per-workgroup KL partials: same partition, same fp64 order, same bits);
//   the others:          one 32 x 32 tile of h = act(z W^T + b) each, z formed from (mu, log_var, eps) right where the
//                        A fragment is loaded -- nobody waits for the z the first group stores.
// The layer is narrow (K = Z <= 32: two 16-deep chunks), so a tile is 4 waves x one 16x16 accumulator pair and there
// is no cross-wave reduction: ((0 + chunk 0) + chunk 1) + bias through the same MFMA sequence as the 16-wave kernel
// of gm_gemm.hip, whose waves 0 and 1 own the two chunks (bit-identical output).
// ------------------------------------------------------------------------------------------
typedef float rp_f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void vae_reparam_fwd_kernel(const float* __restrict__ ml, int64_t ldml,
                                                             const float* __restrict__ eps, gm_slot eps_slot,
                                                             float* __restrict__ z, int64_t ldz,
                                                             float* __restrict__ kl_part, int B, int Z, int nrp,
                                                             const float* __restrict__ W, const float* __restrict__ bias,
                                                             float* __restrict__ H, int64_t ldh, int N, int act,
                                                             int tiles_n) {
    __shared__ double sh[4];
    const float* e = eps + gm_slot_offset(eps_slot);
    if ((int)blockIdx.x < nrp) {                             // workgroup-uniform
        const int n = B * Z, i = blockIdx.x * 256 + threadIdx.x;
        double acc = 0.0;
        if (i < n) {
            const int b = i / Z, c = i % Z;
            const float mu = ml[(int64_t)b * ldml + c], lv = ml[(int64_t)b * ldml + Z + c];
            z[(int64_t)b * ldz + c] = gm_reparam_z(mu, e[i], lv);
            acc = (double)(0.5f * (((mu * mu) + expf(lv)) - lv - 1.f));
        }
        acc = gm_wave_sum_d(acc);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) kl_part[blockIdx.x] = (float)((sh[0] + sh[1]) + (sh[2] + sh[3]));
        return;
    }
    const int tile = blockIdx.x - nrp;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g4 = lane >> 4;
    const int m0 = (tile / tiles_n) * 32 + 16 * (w >> 1), n0 = (tile % tiles_n) * 32 + 16 * (w & 1);
    const int row = min(m0 + i16, B - 1), col = min(n0 + i16, N - 1);
    float4 za[2], wb[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int kb = 16 * c + 4 * g4, kc = min(kb, Z - 4);    // Z % 4 == 0: a group is whole or absent
        const float4 mu = *reinterpret_cast<const float4*>(ml + (int64_t)row * ldml + kc);
        const float4 lv = *reinterpret_cast<const float4*>(ml + (int64_t)row * ldml + Z + kc);
        const float4 ee = *reinterpret_cast<const float4*>(e + (int64_t)row * Z + kc);
        const float4 ww = *reinterpret_cast<const float4*>(W + (int64_t)col * Z + kc);
        const bool ka = kb < Z, ok_a = ka && (m0 + i16 < B), ok_b = ka && (n0 + i16 < N);
        za[c] = make_float4(ok_a ? gm_reparam_z(mu.x, ee.x, lv.x) : 0.f, ok_a ? gm_reparam_z(mu.y, ee.y, lv.y) : 0.f,
                            ok_a ? gm_reparam_z(mu.z, ee.z, lv.z) : 0.f, ok_a ? gm_reparam_z(mu.w, ee.w, lv.w) : 0.f);
        wb[c] = make_float4(ok_b ? ww.x : 0.f, ok_b ? ww.y : 0.f, ok_b ? ww.z : 0.f, ok_b ? ww.w : 0.f);
    }
    rp_f32x4 a0 = rp_f32x4{0.f, 0.f, 0.f, 0.f}, a1 = rp_f32x4{0.f, 0.f, 0.f, 0.f};
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(za[0].x, wb[0].x, a0, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(za[0].y, wb[0].y, a0, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(za[0].z, wb[0].z, a0, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(za[0].w, wb[0].w, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(za[1].x, wb[1].x, a1, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(za[1].y, wb[1].y, a1, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(za[1].z, wb[1].z, a1, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(za[1].w, wb[1].w, a1, 0, 0, 0);
    const int n = n0 + i16;
    const float bv = (bias && n < N) ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 4 * g4 + r;
        float v = 0.f;                                       // the 16-wave reduction's order: 0 + chunk 0 + chunk 1
        v += a0[r];
        if (Z > 16) v += a1[r];
        v += bv;
        if (act == GM_ACT_RELU) v = fmaxf(v, 0.f);
        else if (act == GM_ACT_SIGMOID) v = gm_sigmoid(v);
        if (m < B && n < N) H[(int64_t)m * ldh + n] = v;
    }
}

// ------------------------------------------------------------------------------------------
// The two narrow GEMMs in the middle of the VAE's backward pass as ONE launch (round 4):
//   dz  = dHdec W_d1            [B, Hd] x [Hd, Z]      (decoder layer 1, vae.py:113 backwards)
//   d[mu | log_var] from dz     (gm_vae_reparam_bwd's expressions: vae.py:100-106, 210-212)
//   dHe = (dml W_ml) . [He > 0] [B, 2Z] x [2Z, Hd]     (encoder's mu / log_var layer, vae.py:93-98 backwards)
// A row block's dHe needs only that block's dml, which needs only that block's dz: a workgroup owns 16 rows and runs
// the chain for them; dml goes to memory (the encoder's weight gradient reads it) and stays in LDS for the second
// product.  Summation orders are those of the separate launches: the Hd-deep reduction as sixteen chunk owners
// (chunks w and w + 16 accumulate in one MFMA chain) added in owner order, the 2Z-deep one as up to four chunk
// accumulators added in order -- what the 16-wave kernel's waves and cross-wave reduction do.
// 256 threads: wave q plays chunk owners q, q + 4, q + 8, q + 12 of the first product and column tiles q, q + 4, ...
// of the second.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vae_bwd_mid_kernel(const float* __restrict__ dHdec, int64_t lddh,
                                                         const float* __restrict__ Wd1,      // [Hd, Z]
                                                         const float* __restrict__ ml, int64_t ldml,
                                                         const float* __restrict__ eps, gm_slot eps_slot,
                                                         float* __restrict__ dml, int64_t lddml,
                                                         const float* __restrict__ Wml,      // [2Z, Hd]
                                                         const float* __restrict__ He, int64_t ldhe,
                                                         float* __restrict__ dHe, int64_t lddhe,
                                                         int B, int Hd, int Z) {
    __shared__ float part[16][16][33];                       // chunk owner, row, column (padded)
    __shared__ float sdml[16][68];                           // the block's dml rows (2Z <= 64)
    const float* e = eps + gm_slot_offset(eps_slot);
    const int t = threadIdx.x, lane = t & 63, q = t >> 6;
    const int i16 = lane & 15, g4 = lane >> 4;
    // blockIdx.x: 16-row block; blockIdx.y: group of four 16-column tiles of dHe (one per wave).  Every column group
    // recomputes the block's dz / dml (a 16 x Z product: cheap next to one more launch or a 32-workgroup grid --
    // measured: one workgroup per row block alone made the batch 16 us SLOWER than the two launches); group 0
    // stores dml.
    const int m0 = blockIdx.x * 16;
    const int row = min(m0 + i16, B - 1);
    const bool row_ok = m0 + i16 < B;
    const int nchunks = (Hd + 15) >> 4;
    // ---- dz partials: owner w = q + 4 j accumulates chunks w, w + 16 (two column tiles: Z <= 32); all of a wave's
    // operands are requested before the first MFMA (eight chunk slots: one memory round trip, not eight)
    float4 av[4][2];
    float b0[4][2][4], b1[4][2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = q + 4 * j + 16 * u;
            const int kb = 16 * c + 4 * g4, kc = max(0, min(kb, Hd - 4));          // Hd % 4 == 0
            const bool ka = c < nchunks && kb < Hd;
            const float4 v = *reinterpret_cast<const float4*>(dHdec + (int64_t)row * lddh + kc);
            av[j][u] = (ka && row_ok) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int x = 0; x < 4; ++x) {                    // W_d1[k][col]: rows of Z floats
                const int k = max(0, min(kb + x, Hd - 1));
                const float w0 = Wd1[(int64_t)k * Z + min(i16, Z - 1)];
                const float w1 = Wd1[(int64_t)k * Z + min(16 + i16, Z - 1)];
                b0[j][u][x] = (ka && i16 < Z) ? w0 : 0.f;
                b1[j][u][x] = (ka && 16 + i16 < Z) ? w1 : 0.f;
            }
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int w = q + 4 * j;
        rp_f32x4 a0 = rp_f32x4{0.f, 0.f, 0.f, 0.f}, a1 = rp_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (w + 16 * u < nchunks) {                      // (an absent chunk adds nothing: skip its MFMAs -- +0 steps
                                                             // would not change a bit either)
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][u].x, b0[j][u][0], a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][u].y, b0[j][u][1], a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][u].z, b0[j][u][2], a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][u].w, b0[j][u][3], a0, 0, 0, 0);
                if (Z > 16) {
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][u].x, b1[j][u][0], a1, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][u].y, b1[j][u][1], a1, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][u].z, b1[j][u][2], a1, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][u].w, b1[j][u][3], a1, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {                        // C layout: column = lane & 15, row = 4 (lane >> 4) + r
            part[w][4 * g4 + r][i16] = a0[r];
            part[w][4 * g4 + r][16 + i16] = a1[r];
        }
    }
    // this wave's operands of the second product, requested before the barrier: W_ml's fragment and the mask
    const int K2 = 2 * Z, nch2 = (K2 + 15) >> 4, ntiles = (Hd + 15) >> 4;
    const int ct = blockIdx.y * 4 + q;                       // this wave's column tile (wave-uniform)
    const bool tile_ok = ct < ntiles;
    const int hcol = min(16 * ct + i16, Hd - 1);
    const bool col_ok = tile_ok && 16 * ct + i16 < Hd;
    float bv[4][4], hmask[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int k = 16 * c + 4 * g4 + x;
            const float w = Wml[(int64_t)min(k, K2 - 1) * Hd + hcol];
            bv[c][x] = (k < K2 && col_ok) ? w : 0.f;
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) hmask[r] = He[(int64_t)min(m0 + 4 * g4 + r, B - 1) * ldhe + hcol];
    __syncthreads();
    // ---- dz -> d[mu | log_var] (one (row, column) per thread and pass), to LDS and -- column group 0 -- to memory
    for (int o = t; o < 16 * Z; o += 256) {
        const int r = o / Z, n = o - r * Z, m = m0 + r;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) v += part[w][r][n];
        float dmu = 0.f, dlv = 0.f;
        if (m < B) {
            const float mu = ml[(int64_t)m * ldml + n], lv = ml[(int64_t)m * ldml + Z + n];
            const float ev = e[(int64_t)m * Z + n];
            dmu = v + 0.5f * (2.f * mu);
            dlv = ((v * ev) * expf(lv / 2.f)) / 2.f + 0.5f * (expf(lv) - 1.f);
            if (blockIdx.y == 0) {
                dml[(int64_t)m * lddml + n] = dmu;
                dml[(int64_t)m * lddml + Z + n] = dlv;
            }
        }
        sdml[r][n] = dmu;
        sdml[r][Z + n] = dlv;
    }
    for (int o = t; o < 16 * (64 - 2 * Z); o += 256) {        // the padding the last chunk reads: zeros
        const int r = o / (64 - 2 * Z), n = 2 * Z + (o - r * (64 - 2 * Z));
        sdml[r][n] = 0.f;
    }
    __syncthreads();
    // ---- dHe = (dml W_ml) . [He > 0]: reduction 2Z <= 64 deep = up to four chunk accumulators, added in order
    if (!tile_ok) return;                                    // wave-uniform; no barrier behind this point
    rp_f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        acc[c] = rp_f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < nch2) {
            const float4 af = make_float4(sdml[i16][16 * c + 4 * g4], sdml[i16][16 * c + 4 * g4 + 1],
                                          sdml[i16][16 * c + 4 * g4 + 2], sdml[i16][16 * c + 4 * g4 + 3]);
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bv[c][0], acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bv[c][1], acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bv[c][2], acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bv[c][3], acc[c], 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 4 * g4 + r;
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < nch2) v += acc[c][r];
        if (m < B && col_ok) dHe[(int64_t)m * lddhe + 16 * ct + i16] = (hmask[r] > 0.f) ? v : 0.f;
    }
}

extern "C" int gm_vae_bwd_mid(void* stream, const float* dHdec, int64_t lddh, const float* Wd1, const float* ml,
                              int64_t ldml, const float* eps, gm_slot eps_slot, float* dml, int64_t lddml,
                              const float* Wml, const float* He, int64_t ldhe, float* dHe, int64_t lddhe, int B,
                              int Hd, int Z) {
    GM_CHECK_ARG(dHdec && Wd1 && ml && eps && dml && Wml && He && dHe && B > 0 && Hd > 0 && Z > 0);
    GM_CHECK_ARG(Z <= 32 && Hd % 4 == 0 && lddh >= Hd && lddh % 4 == 0 && ldml >= 2 * Z && lddml >= 2 * Z &&
                 ldhe >= Hd && lddhe >= Hd && (reinterpret_cast<uintptr_t>(dHdec) & 15) == 0);
    GM_CHECK_ARG(dHe != dHdec && (const float*)dml != ml && (const float*)dHe != He);
    GM_CHECK_ARG(Hd <= 16 * 32);                                 // 16 owners x 2 chunks of 16
    hipLaunchKernelGGL(vae_bwd_mid_kernel, dim3((B + 15) / 16, ((Hd + 15) / 16 + 3) / 4), dim3(256), 0, (hipStream_t)stream, dHdec, lddh, Wd1,
                       ml, ldml, eps, eps_slot, dml, lddml, Wml, He, ldhe, dHe, lddhe, B, Hd, Z);
    GM_LAUNCH_RET();
}

extern "C" int gm_vae_reparam_fwd(void* stream, const float* ml, int64_t ldml, const float* eps, gm_slot eps_slot,
                                  float* z, int64_t ldz, float* kl_part, int n_part, int B, int Z, const float* W,
                                  const float* bias, float* H, int64_t ldh, int N, int act) {
    GM_CHECK_ARG(ml && eps && z && kl_part && W && H && B > 0 && Z > 0 && N > 0 && ldml >= 2 * Z && ldz >= Z && ldh >= N);
    GM_CHECK_ARG(Z <= 32 && Z % 4 == 0 && ldml % 4 == 0 && eps_slot.stride % 4 == 0 &&
                 ((reinterpret_cast<uintptr_t>(ml) | reinterpret_cast<uintptr_t>(eps) | reinterpret_cast<uintptr_t>(W)) & 15) == 0);
    GM_CHECK_ARG(act >= GM_ACT_ID && act <= GM_ACT_SIGMOID && (const float*)H != ml && H != z && (const float*)z != ml);
    const int nrp = (B * Z + 255) / 256;
    GM_CHECK_ARG(n_part >= nrp);
    const int tiles_n = (N + 31) / 32, tiles_m = (B + 31) / 32;
    hipLaunchKernelGGL(vae_reparam_fwd_kernel, dim3(nrp + tiles_m * tiles_n), dim3(256), 0, (hipStream_t)stream, ml,
                       ldml, eps, eps_slot, z, ldz, kl_part, B, Z, nrp, W, bias, H, ldh, N, act, tiles_n);
    GM_LAUNCH_RET();
}

extern "C" int gm_vae_reparam_wide(void* stream, const float* ml, int64_t ldml, const float* eps,
                                   gm_slot eps_slot, float* z, int64_t ldz, float* kl_part, int n_part,
                                   int B, int Z) {
    GM_CHECK_ARG(ml && eps && z && kl_part && B > 0 && Z > 0 && ldml >= 2 * Z);
    const int blocks = (B * Z + 255) / 256;
    GM_CHECK_ARG(n_part >= blocks);
    hipLaunchKernelGGL(vae_reparam_wide_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ml, ldml, eps,
                       eps_slot, z, ldz, kl_part, B, Z);
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

// Strided fp64 sum of partial[0..n) by one 1024-thread workgroup: thread t takes elements t, t + 1024, ...
// in batches of 8 INDEPENDENT loads (a row-tile partial array of the fused reconstruction loss has
// 14 336 entries at B = 512: 14 per thread, two memory round trips instead of 14 dependent ones).
// (gm_strided_sum_1024: gm_common.h -- the weight-gradient pair's finalize workgroup runs the same sum)

// out[slot] = scale * sum_{i<n} partial[i]   (single workgroup, fp64 accumulate, fixed order)
__global__ __launch_bounds__(1024) void sum_finalize_kernel(const float* __restrict__ partial, int n,
                                                           float scale, float* __restrict__ out,
                                                           gm_slot out_slot, int64_t* tick) {
    __shared__ double sh[16];
    double acc = gm_wave_sum_d(gm_strided_sum_1024(partial, n));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += sh[w];                 // fixed order
        out[gm_slot_index(out_slot)] = (float)(t * (double)scale);
        if (tick) *tick += 1;                       // after the slot is resolved: last launch of a step
    }
}

extern "C" int gm_sum_finalize(void* stream, const float* partial, int n, float scale, float* out,
                               gm_slot out_slot) {
    GM_CHECK_ARG(partial && out && n > 0);
    hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, partial, n,
                       scale, out, out_slot, (int64_t*)nullptr);
    GM_LAUNCH_RET();
}

extern "C" int gm_sum_finalize_tick(void* stream, const float* partial, int n, float scale, float* out,
                                    gm_slot out_slot, int64_t* tick) {
    GM_CHECK_ARG(partial && out && n > 0 && tick);
    hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, partial, n,
                       scale, out, out_slot, tick);
    GM_LAUNCH_RET();
}

// Two sums in one launch (the VAE step's reconstruction partials and KL partials), optional tick.
__global__ __launch_bounds__(1024) void sum_finalize2_kernel(gm_fin2 f) {
    __shared__ double sh[32];
    gm_fin2_sums(f, sh);
    if (threadIdx.x == 0 && f.tick) *f.tick += 1;            // after both slots are resolved
}

extern "C" int gm_sum_finalize2_tick(void* stream, const float* pa, int na, float scale_a, float* out_a,
                                     gm_slot slot_a, const float* pb, int nb, float scale_b, float* out_b,
                                     gm_slot slot_b, int64_t* tick) {
    GM_CHECK_ARG(pa && pb && out_a && out_b && na > 0 && nb > 0);
    const gm_fin2 f{pa, na, scale_a, out_a, slot_a, pb, nb, scale_b, out_b, slot_b, tick, nullptr};
    hipLaunchKernelGGL(sum_finalize2_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, f);
    GM_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------
// Fused critic head (output_dim == 1; ns_gan.py:59 + the loss lines of train_D / train_G):
//   head_fwd_loss : s_r = act(h_r . w2 + b2)  ->  per-row loss term l_r and dS_r = d loss/d a2_r
//                   (separable variants only; one wave per row, 400-wide dot by wave64 shuffle)
//   head_bwd      : dH[r,n] = dS_r * w2[n] * [h[r,n] > 0]   (ReluBackward of the hidden layer)
//                   gw2[n]  = sum_r dS_r * h[r,n],  gb2 = fl(sum_{x rows} dS) + fl(sum_{g rows} dS)
//                   loss    = inv_b * sum_r l_r  (block 0, fp64, fixed order)
// They replace four launches of the generic path (N=1 GEMV, loss, N=1 dW, K=1 dX).
// ------------------------------------------------------------------------------------------
struct HeadP {
    const float* H; int64_t ldh;
    const float* w2; const float* b2;
    int variant, gen_mode, out_act, R, B, Hd;     // R rows: D mode 2B (x rows then g rows), G mode B
    float hyper[8];
    float inv_b;
    const float* pen;                             // WGAN-GP penalty rows (x rows) or null
    float* S; float* dS; float* rowloss;
    float* dH; int64_t lddh;                      // optional: dH[r,:] = dS_r * w2 * [h > 0] right here
    // optional finalisation by the last workgroup to finish (generator mode: nothing else is left
    // for head_bwd to do): loss = inv_b * sum_r l_r in a fixed order, then the per-graph tick
    float* loss_out; gm_slot loss_slot; unsigned int* done; int64_t* tick;
};

// One wave per row.  VEC4 (Hd % 4 == 0, 16-byte aligned rows; Hd <= 512): each lane keeps its one or
// two float4 of h and w2 in registers -- one round of 16-byte loads for the dot product, and the dH
// row is written from those registers without touching h again.
template <bool VEC4>
__device__ __forceinline__ void head_row(const HeadP& p, int r, int lane) {
    const float* h = p.H + (int64_t)r * p.ldh;
    float acc = 0.f;
    float4 hv[2], wv[2];
    if (VEC4) {
        const int n4 = p.Hd >> 2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i4 = min(lane + 64 * j, n4 - 1);                   // clamped: branch-free loads
            hv[j] = reinterpret_cast<const float4*>(h)[i4];
            wv[j] = reinterpret_cast<const float4*>(p.w2)[i4];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (lane + 64 * j < n4) {
                acc = fmaf(hv[j].x, wv[j].x, acc); acc = fmaf(hv[j].y, wv[j].y, acc);
                acc = fmaf(hv[j].z, wv[j].z, acc); acc = fmaf(hv[j].w, wv[j].w, acc);
            }
        }
    } else {
        for (int i = lane; i < p.Hd; i += 64) acc = fmaf(h[i], p.w2[i], acc);
    }
    acc = gm_wave_sum(acc);
    float a2 = acc + p.b2[0];
    float s = a2;
    if (p.out_act == GM_ACT_SIGMOID) s = gm_sigmoid(a2);
    else if (p.out_act == GM_ACT_RELU) s = fmaxf(a2, 0.f);
    float ds = 0.f;
    if (lane == 0) {
        const bool D = !p.gen_mode;
        const bool is_x = D && r < p.B;
        float lx, lg, dx, dg;
        sample_terms(p.variant, D, is_x ? s : 0.5f, is_x ? 0.5f : s, p.inv_b, p.hyper, lx, lg, dx, dg);
        float l = is_x ? lx : lg;
        if (is_x && p.pen) l += p.hyper[7] * p.pen[r];     // gradient penalty rows (WGAN-GP, DRAGAN)
        ds = act_grad(is_x ? dx : dg, s, p.out_act);
        p.S[r] = s;
        p.dS[r] = ds;
        p.rowloss[r] = l;
    }
    if (p.dH) {
        // this wave still has its row of h hot in L1: write the hidden-layer gradient now, so the
        // backward head kernel only reads h once more for the column sums and writes nothing wide
        ds = __shfl(ds, 0, 64);
        float* o = p.dH + (int64_t)r * p.lddh;
        if (VEC4) {
            const int n4 = p.Hd >> 2;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int i4 = lane + 64 * j;
                if (i4 < n4) {
                    float4 d;
                    d.x = (hv[j].x > 0.f) ? ds * wv[j].x : 0.f; d.y = (hv[j].y > 0.f) ? ds * wv[j].y : 0.f;
                    d.z = (hv[j].z > 0.f) ? ds * wv[j].z : 0.f; d.w = (hv[j].w > 0.f) ? ds * wv[j].w : 0.f;
                    reinterpret_cast<float4*>(o)[i4] = d;
                }
            }
        } else {
            for (int i = lane; i < p.Hd; i += 64) o[i] = (h[i] > 0.f) ? ds * p.w2[i] : 0.f;
        }
    }
}

// "Last workgroup done" reduction: every workgroup publishes its rows (agent-scope fence), bumps a
// device counter, and the one that observes gridDim-1 sums ALL row terms in a fixed order -- the
// result does not depend on which workgroup happens to be last.  It re-arms the counter for the
// next launch and advances the iteration counter (single writer; later kernels of the same
// iteration address their slots with add - mul).
__device__ void head_finalize(const HeadP& p) {
    __shared__ int is_last;
    __shared__ double part[4];
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int prev = __hip_atomic_fetch_add(p.done, 1u, __ATOMIC_ACQ_REL,
                                                         __HIP_MEMORY_SCOPE_AGENT);
        is_last = (prev == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    double sl = 0.0;
    for (int r = threadIdx.x; r < p.R; r += 256)
        sl += (double)__hip_atomic_load(p.rowloss + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sl = gm_wave_sum_d(sl);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = ((part[0] + part[1]) + part[2]) + part[3];
        p.loss_out[gm_slot_index(p.loss_slot)] = (float)(tot * (double)p.inv_b);
        __hip_atomic_store(p.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p.tick) *p.tick += 1;
    }
}

// Launched with one wave per workgroup (R workgroups spread over all CUs: the kernel is a chain of
// dependent latencies, not bandwidth) -- or with 4 waves when the last-workgroup finalisation is on.
template <bool VEC4>
__global__ __launch_bounds__(256) void head_fwd_loss_kernel(HeadP p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + wave;
    if (r < p.R) head_row<VEC4>(p, r, lane);
    if (p.loss_out) head_finalize(p);           // kernel-argument uniform: every thread takes it
}

static int head_fwd_impl(void* stream, int variant, int gen_mode, const float* H, int64_t ldh,
                         const float* w2, const float* b2, int out_act, int B, int Hd,
                         const float* hyper, int n_hyper, float inv_b, const float* pen, float* S,
                         float* dS, float* rowloss, float* dH, int64_t lddh, float* loss_out,
                         gm_slot loss_slot, unsigned int* done, int64_t* tick);

extern "C" int gm_head_fwd_loss(void* stream, int variant, int gen_mode, const float* H,
                                int64_t ldh, const float* w2, const float* b2, int out_act, int B,
                                int Hd, const float* hyper, int n_hyper, float inv_b,
                                const float* pen, float* S, float* dS, float* rowloss, float* dH,
                                int64_t lddh) {
    gm_slot z; z.ctr = nullptr; z.mul = 0; z.add = 0; z.ring = 0; z.stride = 0;
    return head_fwd_impl(stream, variant, gen_mode, H, ldh, w2, b2, out_act, B, Hd, hyper, n_hyper,
                         inv_b, pen, S, dS, rowloss, dH, lddh, nullptr, z, nullptr, nullptr);
}

// head_fwd_loss that also finalises the loss scalar (and optionally ticks the iteration counter)
// from its last workgroup: generator mode needs no head_bwd launch at all then.
extern "C" int gm_head_fwd_loss_final(void* stream, int variant, int gen_mode, const float* H,
                                      int64_t ldh, const float* w2, const float* b2, int out_act,
                                      int B, int Hd, const float* hyper, int n_hyper, float inv_b,
                                      const float* pen, float* S, float* dS, float* rowloss,
                                      float* dH, int64_t lddh, float* loss_out, gm_slot loss_slot,
                                      unsigned int* done_ctr, int64_t* tick) {
    GM_CHECK_ARG(loss_out && done_ctr);
    return head_fwd_impl(stream, variant, gen_mode, H, ldh, w2, b2, out_act, B, Hd, hyper, n_hyper,
                         inv_b, pen, S, dS, rowloss, dH, lddh, loss_out, loss_slot, done_ctr, tick);
}

static int head_fwd_impl(void* stream, int variant, int gen_mode, const float* H, int64_t ldh,
                         const float* w2, const float* b2, int out_act, int B, int Hd,
                         const float* hyper, int n_hyper, float inv_b, const float* pen, float* S,
                         float* dS, float* rowloss, float* dH, int64_t lddh, float* loss_out,
                         gm_slot loss_slot, unsigned int* done, int64_t* tick) {
    GM_CHECK_ARG(H && w2 && b2 && S && dS && rowloss && B > 0 && Hd > 0 && n_hyper >= 0 && n_hyper <= 8);
    GM_CHECK_ARG(variant != GM_LOSS_RA || gen_mode);
    GM_CHECK_ARG(variant != GM_LOSS_FISHER || gen_mode);
    HeadP p{};
    p.H = H; p.ldh = ldh; p.w2 = w2; p.b2 = b2; p.variant = variant; p.gen_mode = gen_mode;
    p.out_act = out_act; p.B = B; p.R = gen_mode ? B : 2 * B; p.Hd = Hd; p.inv_b = inv_b;
    for (int i = 0; i < n_hyper; ++i) p.hyper[i] = hyper[i];
    p.pen = pen; p.S = S; p.dS = dS; p.rowloss = rowloss; p.dH = dH; p.lddh = lddh;
    p.loss_out = loss_out; p.loss_slot = loss_slot; p.done = done; p.tick = tick;
    const bool vec4 = (Hd % 4 == 0) && Hd <= 512 && (ldh % 4 == 0) && (!dH || lddh % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(H) | reinterpret_cast<uintptr_t>(w2) |
                        reinterpret_cast<uintptr_t>(dH)) & 15) == 0;
    const int wpb = loss_out ? 4 : 1;                      // the finalisation sums with 256 threads
    const dim3 grid((p.R + wpb - 1) / wpb), block(64 * wpb);
    if (vec4) hipLaunchKernelGGL(head_fwd_loss_kernel<true>, grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(head_fwd_loss_kernel<false>, grid, block, 0, (hipStream_t)stream, p);
    GM_LAUNCH_RET();
}

// HeadBwdP, head_bwd_body and head_bwd_kernel live in gm_head.h: the weight-gradient GEMM can
// co-schedule the head workgroups in its own launch (gm_linear_bwd_dw_adam_head, gm_gemm.hip).

static int head_bwd_launch(void* stream, const gm_head_bwd_args& a) {
    HeadBwdP p{};
    const int rc = gm_head_from_args(a, &p);
    if (rc) return rc;
    hipLaunchKernelGGL(head_bwd_kernel, dim3(gm_head_bwd_blocks(p)), dim3(1024), 0,
                       (hipStream_t)stream, p);
    GM_LAUNCH_RET();
}

extern "C" int gm_head_bwd(void* stream, const float* H, int64_t ldh, const float* dS,
                           const float* w2, const float* rowloss, float* dH, int64_t lddh,
                           float* gw2, float* gb2, float* loss_out, gm_slot loss_slot, float inv_b,
                           int gen_mode, int B, int Hd) {
    gm_head_bwd_args a{};
    a.H = H; a.ldh = ldh; a.dS = dS; a.w2 = const_cast<float*>(w2); a.rowloss = rowloss; a.dH = dH;
    a.lddh = lddh; a.gw2 = gw2; a.gb2 = gb2; a.loss_out = loss_out; a.loss_slot = loss_slot;
    a.inv_b = inv_b; a.gen_mode = gen_mode; a.B = B; a.Hd = Hd;
    return head_bwd_launch(stream, a);
}

// head_bwd with the optimizer step for (w2, b2) and/or the per-graph tick folded in.
extern "C" int gm_head_bwd_fused(void* stream, const float* H, int64_t ldh, const float* dS,
                                 float* w2, float* b2, const float* rowloss, float* dH,
                                 int64_t lddh, float* gw2, float* gb2, float* loss_out,
                                 gm_slot loss_slot, float inv_b, int gen_mode, int B, int Hd,
                                 int with_adam, float* mW, float* vW, float* mb, float* vb,
                                 const float* sched, gm_slot sched_slot, double beta1, double beta2,
                                 double eps, double weight_decay, float clamp, int64_t* tick) {
    gm_head_bwd_args a{};
    a.H = H; a.ldh = ldh; a.dS = dS; a.w2 = w2; a.b2 = b2; a.rowloss = rowloss; a.dH = dH;
    a.lddh = lddh; a.gw2 = gw2; a.gb2 = gb2; a.loss_out = loss_out; a.loss_slot = loss_slot;
    a.inv_b = inv_b; a.gen_mode = gen_mode; a.B = B; a.Hd = Hd; a.with_adam = with_adam;
    a.mW = mW; a.vW = vW; a.mb = mb; a.vb = vb; a.sched = sched; a.sched_slot = sched_slot;
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay; a.clamp = clamp;
    a.tick = tick;
    return head_bwd_launch(stream, a);
}

// ------------------------------------------------------------------------------------------
// K15 InfoGAN mutual-information loss (info_gan.py:295-302):
//   disc = F.cross_entropy(q[:, :nd], argmax(onehot))      (mean over B of logsumexp - q[target])
//   cont = F.mse_loss(q[:, nd:], c2)                       (mean over B*nc elements)
//   loss = lambda * (disc + cont);  dq = d loss / d q
// q: [B, nd+nc] (Q's identity output); noise: [B, zd+nd+nc] rows = [z | one-hot c1 | c2] (the G
// input of this step, info_gan.py:306-325), read through a ring slot.  One workgroup.
// ------------------------------------------------------------------------------------------
// One row's terms from register copies of the row (qr: nd + nc outputs of Q; cr: the one-hot block and the
// continuous code of the noise row).  Expressions and their order are the generic loop's.
template <int ND, int NC>
__device__ __forceinline__ void info_q_row(const float (&qr)[ND + NC], const float (&cr)[ND + NC], float lambda,
                                           float inv_b, float inv_bc, float* __restrict__ dr, double& acc_d,
                                           double& acc_c) {
    int tgt = 0;                                             // torch.max(..., 1)[1]: first maximal index
    float best = cr[0];
#pragma unroll
    for (int j = 1; j < ND; ++j) { if (cr[j] > best) { best = cr[j]; tgt = j; } }
    float mx = qr[0];                                        // log_softmax (max-shifted, like at::log_softmax)
#pragma unroll
    for (int j = 1; j < ND; ++j) mx = fmaxf(mx, qr[j]);
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < ND; ++j) se += expf(qr[j] - mx);
    const float lse = logf(se);
    float qt = qr[0];
#pragma unroll
    for (int j = 1; j < ND; ++j) qt = (j == tgt) ? qr[j] : qt;
    acc_d += (double)(-((qt - mx) - lse));
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        const float sm = expf((qr[j] - mx) - lse);
        dr[j] = lambda * ((sm - (j == tgt ? 1.f : 0.f)) * inv_b);
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const float d = qr[ND + j] - cr[ND + j];
        acc_c += (double)(d * d);
        dr[ND + j] = lambda * ((2.f * d) * inv_bc);
    }
}

// ND / NC > 0: compile-time widths -- a thread's row lives in registers and ALL of its loads go out before the first
// use (round 4: the generic loops made five dependent trips to memory per row; 11.8 us per launch at B = 256,
// info_gan.py's 10 + 10 code).  ND == 0: any widths.
template <int ND, int NC>
__global__ __launch_bounds__(256) void info_q_loss_kernel(const float* __restrict__ q, int64_t ldq,
                                                         const float* __restrict__ noise,
                                                         gm_slot noise_slot, int64_t ldn, int B,
                                                         int Bscale, int zd, int nd, int nc, float lambda,
                                                         float* __restrict__ dq, int64_t lddq,
                                                         float* __restrict__ loss_out,
                                                         gm_slot loss_slot) {
    __shared__ double sh[4];
    const float* nz = noise + gm_slot_offset(noise_slot);
    // Bscale: rows of the GLOBAL batch (both means' denominator; = B on one GPU)
    const float inv_b = 1.0f / (float)Bscale, inv_bc = 1.0f / (float)(Bscale * nc);
    double acc_d = 0.0, acc_c = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* qr = q + (int64_t)b * ldq;
        const float* nr = nz + (int64_t)b * ldn;
        float* dr = dq + (int64_t)b * lddq;
        if constexpr (ND > 0) {
            float qv[ND + NC], cv[ND + NC];
#pragma unroll
            for (int j = 0; j < ND + NC; ++j) { qv[j] = qr[j]; cv[j] = nr[zd + j]; }
            info_q_row<ND, NC>(qv, cv, lambda, inv_b, inv_bc, dr, acc_d, acc_c);
            continue;
        }
        // target = argmax of the one-hot block (torch.max(...,1)[1]: first maximal index)
        int tgt = 0;
        float best = nr[zd];
        for (int j = 1; j < nd; ++j) { const float v = nr[zd + j]; if (v > best) { best = v; tgt = j; } }
        // log_softmax (max-shifted, like at::log_softmax)
        float mx = qr[0];
        for (int j = 1; j < nd; ++j) mx = fmaxf(mx, qr[j]);
        float se = 0.f;
        for (int j = 0; j < nd; ++j) se += expf(qr[j] - mx);
        const float lse = logf(se);
        acc_d += (double)(-((qr[tgt] - mx) - lse));
        for (int j = 0; j < nd; ++j) {
            const float sm = expf((qr[j] - mx) - lse);
            dr[j] = lambda * ((sm - (j == tgt ? 1.f : 0.f)) * inv_b);
        }
        for (int j = 0; j < nc; ++j) {
            const float d = qr[nd + j] - nr[zd + nd + j];
            acc_c += (double)(d * d);
            dr[nd + j] = lambda * ((2.f * d) * inv_bc);
        }
    }
    double v[2] = {acc_d, acc_c};
    float outv[2];
    for (int k = 0; k < 2; ++k) {
        const double a = gm_wave_sum_d(v[k]);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
        __syncthreads();
        outv[k] = (float)(((sh[0] + sh[1]) + (sh[2] + sh[3])) * (double)(k == 0 ? inv_b : inv_bc));
    }
    if (threadIdx.x == 0) loss_out[gm_slot_index(loss_slot)] = lambda * (outv[0] + outv[1]);
}

extern "C" int gm_info_q_loss_dp(void* stream, const float* q, int64_t ldq, const float* noise,
                                 gm_slot noise_slot, int64_t ldn, int B, int B_global, int z_dim,
                                 int disc_dim, int cont_dim, float lambda, float* dq, int64_t lddq,
                                 float* loss_out, gm_slot loss_slot) {
    GM_CHECK_ARG(q && noise && dq && loss_out && B > 0 && B_global >= B && disc_dim > 0 && cont_dim > 0 && z_dim >= 0);
    if (disc_dim == 10 && cont_dim == 10)                       // info_gan.py:78-79's code widths
        hipLaunchKernelGGL((info_q_loss_kernel<10, 10>), dim3(1), dim3(256), 0, (hipStream_t)stream, q, ldq, noise,
                           noise_slot, ldn, B, B_global, z_dim, disc_dim, cont_dim, lambda, dq, lddq, loss_out,
                           loss_slot);
    else
        hipLaunchKernelGGL((info_q_loss_kernel<0, 0>), dim3(1), dim3(256), 0, (hipStream_t)stream, q, ldq, noise,
                           noise_slot, ldn, B, B_global, z_dim, disc_dim, cont_dim, lambda, dq, lddq, loss_out,
                           loss_slot);
    GM_LAUNCH_RET();
}
extern "C" int gm_info_q_loss(void* stream, const float* q, int64_t ldq, const float* noise,
                              gm_slot noise_slot, int64_t ldn, int B, int z_dim, int disc_dim,
                              int cont_dim, float lambda, float* dq, int64_t lddq, float* loss_out,
                              gm_slot loss_slot) {
    return gm_info_q_loss_dp(stream, q, ldq, noise, noise_slot, ldn, B, B, z_dim, disc_dim, cont_dim, lambda,
                             dq, lddq, loss_out, loss_slot);
}

// ------------------------------------------------------------------------------------------
// K16 BEGAN (be_gan.py:212-258, 189-195).  The critic is an autoencoder; losses are per-row L1
// reconstruction errors.  Device-resident state (float unless noted), so that the proportional
// controller and the two ReduceLROnPlateau schedulers need no host sync:
//   st[0]=K  st[1]=DX  st[2]=DG  st[3]=convergence  st[4]=scaleD  st[5]=scaleG
//   dst (double): [0]=best [1]=lrD [2]=lrG [3]=lrD0 [4]=lrG0 ; ist (int64): [0]=num_bad
// ------------------------------------------------------------------------------------------
// rowsum[r] = sum_i |Y[r,i] - X[r,i]| ;  dY[r,i] = coef_r * sign(Y - X), coef_r = 1/B for the first B
// rows (or all rows when K is null), -K/B for the rest (d(DX - K*DG)/dY, be_gan.py:225-236).
__global__ __launch_bounds__(256) void l1_rows_kernel(const float* __restrict__ Y, int64_t ldy,
                                                     const float* __restrict__ X, int64_t ldx,
                                                     int R, int I, int B, int Bscale,
                                                     const float* __restrict__ K,
                                                     float* __restrict__ dY, int64_t lddy,
                                                     float* __restrict__ rowsum) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= R) return;
    // Bscale: rows of the GLOBAL batch (the mean's denominator; = B on one GPU)
    const float coef = (K != nullptr && r >= B) ? (-K[0]) / (float)Bscale : 1.0f / (float)Bscale;
    float acc = 0.f;
    for (int i = lane; i < I; i += 64) {
        const float d = Y[(int64_t)r * ldy + i] - X[(int64_t)r * ldx + i];
        acc += fabsf(d);
        dY[(int64_t)r * lddy + i] = d > 0.f ? coef : (d < 0.f ? -coef : 0.f);
    }
    acc = gm_wave_sum(acc);
    if (lane == 0) rowsum[r] = acc;
}

extern "C" int gm_l1_rows_dp(void* stream, const float* Y, int64_t ldy, const float* X, int64_t ldx,
                             int R, int I, int B, int B_global, const float* K_dev, float* dY,
                             int64_t lddy, float* rowsum) {
    GM_CHECK_ARG(Y && X && dY && rowsum && R > 0 && I > 0 && B > 0 && B_global >= B);
    hipLaunchKernelGGL(l1_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, Y, ldy, X,
                       ldx, R, I, B, B_global, K_dev, dY, lddy, rowsum);
    GM_LAUNCH_RET();
}
extern "C" int gm_l1_rows(void* stream, const float* Y, int64_t ldy, const float* X, int64_t ldx,
                          int R, int I, int B, const float* K_dev, float* dY, int64_t lddy,
                          float* rowsum) {
    return gm_l1_rows_dp(stream, Y, ldy, X, ldx, R, I, B, B, K_dev, dY, lddy, rowsum);
}

// DX = mean(rows[0:B]), DG = mean(rows[B:2B]), D_loss = DX - K*DG  (be_gan.py:225-236)
__global__ __launch_bounds__(256) void began_dloss_kernel(const float* __restrict__ rows, int B,
                                                         int Bscale, float* __restrict__ st,
                                                         float* __restrict__ loss_out,
                                                         gm_slot loss_slot) {
    __shared__ double sh[4];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < B; i += 256) { a += (double)rows[i]; b += (double)rows[B + i]; }
    double v[2] = {a, b};
    float m[2];
    for (int k = 0; k < 2; ++k) {
        const double w = gm_wave_sum_d(v[k]);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = w;
        __syncthreads();
        m[k] = (float)(((sh[0] + sh[1]) + (sh[2] + sh[3])) / (double)Bscale);
    }
    if (threadIdx.x == 0) {
        st[1] = m[0];
        st[2] = m[1];
        loss_out[gm_slot_index(loss_slot)] = m[0] - (st[0] * m[1]);
    }
}

extern "C" int gm_began_dloss_dp(void* stream, const float* rows, int B, int B_global, float* state,
                                 float* loss_out, gm_slot loss_slot) {
    GM_CHECK_ARG(rows && state && loss_out && B > 0 && B_global >= B);
    hipLaunchKernelGGL(began_dloss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rows, B, B_global,
                       state, loss_out, loss_slot);
    GM_LAUNCH_RET();
}
extern "C" int gm_began_dloss(void* stream, const float* rows, int B, float* state, float* loss_out,
                              gm_slot loss_slot) {
    return gm_began_dloss_dp(stream, rows, B, B, state, loss_out, loss_slot);
}

// End of a BEGAN iteration (be_gan.py:189-195): convergence measure, K <- clip(K + lambda*(gamma*DX
// - DG), 0, 1), both ReduceLROnPlateau(mode=min, factor=.5, threshold=.01 rel, patience) steps
// (torch/optim/lr_scheduler.py), learning-rate scales for the Adam kernels, and the graph tick.
__global__ void began_update_kernel(float* st, double* dst, int64_t* ist, float gamma, float lambda,
                                    int64_t patience, int64_t* tick) {
    const float DX = st[1], DG = st[2], K = st[0];
    const float diff = gamma * DX - DG;
    const float conv = DX + fabsf(diff);
    st[3] = conv;
    const float ku = K + lambda * diff;
    st[0] = fminf(fmaxf(0.f, ku), 1.f);
    const double cur = (double)conv;
    if (cur < dst[0] * (1.0 - 0.01)) { dst[0] = cur; ist[0] = 0; }
    else ist[0] += 1;
    if (ist[0] > patience) {
        for (int k = 1; k <= 2; ++k) {
            const double old_lr = dst[k];
            const double new_lr = fmax(old_lr * 0.5, 0.0);
            if (old_lr - new_lr > 1e-8) dst[k] = new_lr;
        }
        ist[0] = 0;
    }
    st[4] = (float)(dst[1] / dst[3]);
    st[5] = (float)(dst[2] / dst[4]);
    if (tick) *tick += 1;
}

extern "C" int gm_began_update(void* stream, float* state, double* dstate, int64_t* istate,
                               float gamma, float lambda, int64_t patience, int64_t* tick) {
    GM_CHECK_ARG(state && dstate && istate);
    hipLaunchKernelGGL(began_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, dstate,
                       istate, gamma, lambda, patience, tick);
    GM_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------
// K13 DRAGAN (dra_gan.py:198-223; second backward hand-derived in SURVEY.md A.3).
// Critic D = sigma(a2), a2 = h.w2 + b2, h = relu(a1), a1 = x_hat W1^T + b1.
//   std   : unbiased std of ALL B*I elements of the real batch (images.data.std(), :204)
//   x_hat : delta*x + (1-delta)*(x + C*std*U)                     (:203-205, a fresh leaf)
//   rows  : g_b = s'_b v_b with v_b = (m1_b . w2) W1 (GEMM), n_b = ||g_b||, pen_b = (n_b-K)^2,
//           gamma_b = lambda*inv_b*2(n_b-K) g_b/n_b,  dv_b = s'_b gamma_b,
//           ds'_b = gamma_b . v_b,  da2_b = ds'_b * s'_b (1-2 s_b)
//   head  : gw2 += colsum(m1 . T) + da2^T h ; gb2 += sum da2 ; da1 = (da2 w2) . m1   (T = dv W1^T)
// ------------------------------------------------------------------------------------------
// Unbiased std over n = R*I elements (images.data.std(), dra_gan.py:204): two fp64 sums.
// STD_NB workgroups sum float4 slices (a single workgroup walking 200 704 elements took 45 us -- a quarter
// of DRAGAN's iteration); each publishes its pair of partial sums, bumps a counter in the workspace, and the
// workgroup that observes STD_NB-1 adds the partials in index order (same bits whichever is last), writes
// the result and re-arms the counter.  ws: 2*STD_NB doubles + one counter, zero-initialised once by the caller.
// MODE 0: out[0] = std.  MODE 1 (data parallel: the std is over the GLOBAL batch -> every rank contributes
// the sums of its rows, gm_std_from_sums finishes after the scalar all-reduce): out[0..1] = (sum x, sum x^2).
constexpr int STD_NB = 64;
template <bool VEC4, int MODE>
__global__ __launch_bounds__(256) void std_multi_kernel(const float* __restrict__ X, int64_t ldx, int R, int I,
                                                       double* __restrict__ ws, float* __restrict__ out) {
    __shared__ double sh[2][4];
    __shared__ int is_last;
    double s1 = 0.0, s2 = 0.0;
    const int tid = blockIdx.x * 256 + threadIdx.x, nth = STD_NB * 256;
    if (VEC4) {
        const int I4 = I >> 2;
        const int n4 = R * I4;
        for (int i0 = tid; i0 < n4; i0 += 4 * nth) {            // four independent 16-byte loads in flight
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + u * nth, n4 - 1);
                v[u] = *reinterpret_cast<const float4*>(X + (int64_t)(i / I4) * ldx + 4 * (i % I4));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * nth < n4) {
                    const double a = v[u].x, b = v[u].y, c = v[u].z, d = v[u].w;
                    s1 += (a + b) + (c + d);
                    s2 += (a * a + b * b) + (c * c + d * d);
                }
        }
    } else {
        const int64_t n = (int64_t)R * I;
        for (int64_t i = tid; i < n; i += nth) {
            const double v = (double)X[(i / I) * ldx + (i % I)];
            s1 += v; s2 += v * v;
        }
    }
    s1 = gm_wave_sum_d(s1); s2 = gm_wave_sum_d(s2);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s1; sh[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    unsigned int* ctr = reinterpret_cast<unsigned int*>(ws + 2 * STD_NB);
    if (threadIdx.x == 0) {
        const double a = ((sh[0][0] + sh[0][1]) + sh[0][2]) + sh[0][3];
        const double b = ((sh[1][0] + sh[1][1]) + sh[1][2]) + sh[1][3];
        __hip_atomic_store(ws + 2 * blockIdx.x, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ws + 2 * blockIdx.x + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        const unsigned int prev = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (prev == STD_NB - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!is_last || threadIdx.x >= 64) return;
    __threadfence();
    static_assert(STD_NB == 64, "one wave adds the partials");
    double a = __hip_atomic_load(ws + 2 * threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double b = __hip_atomic_load(ws + 2 * threadIdx.x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a = gm_wave_sum_d(a); b = gm_wave_sum_d(b);
    if (threadIdx.x == 0) {
        if (MODE == 0) {
            const double n = (double)R * (double)I;
            const double mean = a / n;
            const double var = (b - n * mean * mean) / (n - 1.0);
            out[0] = (float)sqrt(var > 0.0 ? var : 0.0);
        } else {
            out[0] = (float)a; out[1] = (float)b;
        }
        __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void std_from_sums_kernel(const float* __restrict__ sums2, int64_t n, float* __restrict__ out) {
    const double a = (double)sums2[0], b = (double)sums2[1];
    const double mean = a / (double)n;
    const double var = (b - (double)n * mean * mean) / (double)(n - 1);
    out[0] = (float)sqrt(var > 0.0 ? var : 0.0);
}
template <int MODE>
static int std_launch(void* stream, const float* X, int64_t ldx, int R, int I, void* ws, float* out) {
    GM_CHECK_ARG(X && out && ws && R > 0 && I > 0 && ldx >= I && (int64_t)R * I < (1ll << 31));
    GM_CHECK_ARG((reinterpret_cast<uintptr_t>(ws) & 7) == 0);
    const bool vec4 = (I % 4 == 0) && (ldx % 4 == 0) && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
    if (vec4) hipLaunchKernelGGL((std_multi_kernel<true, MODE>), dim3(STD_NB), dim3(256), 0, (hipStream_t)stream,
                                 X, ldx, R, I, (double*)ws, out);
    else hipLaunchKernelGGL((std_multi_kernel<false, MODE>), dim3(STD_NB), dim3(256), 0, (hipStream_t)stream,
                            X, ldx, R, I, (double*)ws, out);
    GM_LAUNCH_RET();
}
extern "C" int gm_std_sums(void* stream, const float* X, int64_t ldx, int R, int I, float* out2, void* ws) {
    return std_launch<1>(stream, X, ldx, R, I, ws, out2);
}
extern "C" int gm_std_from_sums(void* stream, const float* sums2, int64_t n_total, float* out) {
    GM_CHECK_ARG(sums2 && out && n_total > 1);
    hipLaunchKernelGGL(std_from_sums_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sums2, n_total, out);
    GM_LAUNCH_RET();
}
extern "C" int gm_std_all(void* stream, const float* X, int64_t ldx, int R, int I, float* out, void* ws) {
    GM_CHECK_ARG((int64_t)R * I > 1);
    return std_launch<0>(stream, X, ldx, R, I, ws, out);
}

__global__ __launch_bounds__(256) void dragan_xhat_kernel(const float* __restrict__ x, int64_t ldx,
                                                         const float* __restrict__ delta,
                                                         gm_slot delta_slot,
                                                         const float* __restrict__ U, gm_slot u_slot,
                                                         const float* __restrict__ stdv, float C,
                                                         float* __restrict__ out, int64_t ldo, int B,
                                                         int I) {
    const float* dl = delta + gm_slot_offset(delta_slot);
    const float* u = U + gm_slot_offset(u_slot);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const float d = dl[b], om = 1.f - d, cs = C * stdv[0];
    for (int i = lane; i < I; i += 64) {
        const float xv = x[(int64_t)b * ldx + i];
        out[(int64_t)b * ldo + i] = d * xv + om * (xv + cs * u[(int64_t)b * I + i]);
    }
}

extern "C" int gm_dragan_xhat(void* stream, const float* x, int64_t ldx, const float* delta,
                              gm_slot delta_slot, const float* U, gm_slot u_slot, const float* std_dev,
                              float C, float* out, int64_t ldo, int B, int I) {
    GM_CHECK_ARG(x && delta && U && std_dev && out && B > 0 && I > 0);
    hipLaunchKernelGGL(dragan_xhat_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       delta, delta_slot, U, u_slot, std_dev, C, out, ldo, B, I);
    GM_LAUNCH_RET();
}

__global__ __launch_bounds__(256) void dragan_rows_kernel(const float* __restrict__ s,
                                                         const float* __restrict__ V, int64_t ldv,
                                                         float* __restrict__ dv, int64_t lddv,
                                                         float* __restrict__ da2,
                                                         float* __restrict__ pen, float lambda,
                                                         float inv_b, float Kn, int B, int I) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const float sb = s[b], sp = (1.f - sb) * sb;             // s' = sigma'(a2)
    const float* v = V + (int64_t)b * ldv;
    float ss = 0.f;
    for (int i = lane; i < I; i += 64) { const float g = sp * v[i]; ss += g * g; }
    ss = gm_wave_sum(ss);
    const float n = sqrtf(ss);                                // ||g_b||
    const float dn = n - Kn;
    const float coef = (n > 0.f) ? (lambda * (inv_b * (2.f * dn))) / n : 0.f;   // gamma = coef * g
    // ds' = sum_j gamma_j v_j = coef * s' * ||v||^2 = coef * ||g||^2 / s'
    float dot = 0.f;
    float* o = dv + (int64_t)b * lddv;
    for (int i = lane; i < I; i += 64) {
        const float g = sp * v[i];
        const float gam = coef * g;
        dot += gam * v[i];
        o[i] = sp * gam;                                      // dv = s' * gamma
    }
    dot = gm_wave_sum(dot);
    if (lane == 0) {
        pen[b] = dn * dn;
        da2[b] = dot * (sp * (1.f - 2.f * sb));               // ds' * d s'/d a2
    }
}

extern "C" int gm_dragan_rows(void* stream, const float* s, const float* V, int64_t ldv, float* dv,
                              int64_t lddv, float* da2, float* pen, float lambda, float inv_b,
                              float K_norm, int B, int I) {
    GM_CHECK_ARG(s && V && dv && da2 && pen && B > 0 && I > 0);
    hipLaunchKernelGGL(dragan_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, s, V,
                       ldv, dv, lddv, da2, pen, lambda, inv_b, K_norm, B, I);
    GM_LAUNCH_RET();
}

__global__ __launch_bounds__(1024) void dragan_head_bwd_kernel(const float* __restrict__ H, int64_t ldh,
                                                              const float* __restrict__ T, int64_t ldt,
                                                              const float* __restrict__ da2,
                                                              const float* __restrict__ w2,
                                                              float* __restrict__ gw2,
                                                              float* __restrict__ gb2,
                                                              float* __restrict__ dA1, int64_t ldd,
                                                              int B, int Hd, int accumulate) {
    __shared__ float sh[HB_RG][HB_COLS + 1];
    __shared__ double shd[16];
    const int cl = threadIdx.x & (HB_COLS - 1), rg = threadIdx.x / HB_COLS;
    const int c = blockIdx.x * HB_COLS + cl;
    float acc = 0.f;
    if (c < Hd) {
        const float w = w2[c];
        for (int r = rg; r < B; r += HB_RG) {
            const float h = H[(int64_t)r * ldh + c];
            const float d = da2[r];
            const bool on = h > 0.f;
            acc += (on ? T[(int64_t)r * ldt + c] : 0.f) + d * h;
            dA1[(int64_t)r * ldd + c] = on ? d * w : 0.f;
        }
    }
    sh[rg][cl] = acc;
    __syncthreads();
    if (rg == 0 && c < Hd) {
        float v = 0.f;
        for (int q = 0; q < HB_RG; ++q) v += sh[q][cl];
        gw2[c] = accumulate ? gw2[c] + v : v;
    }
    if (blockIdx.x == 0) {
        double sd = 0.0;
        for (int r = threadIdx.x; r < B; r += 1024) sd += (double)da2[r];
        sd = gm_wave_sum_d(sd);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = sd;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0.0;
            for (int q = 0; q < 16; ++q) tot += shd[q];
            gb2[0] = accumulate ? gb2[0] + (float)tot : (float)tot;
        }
    }
}

extern "C" int gm_dragan_head_bwd(void* stream, const float* H, int64_t ldh, const float* T,
                                  int64_t ldt, const float* da2, const float* w2, float* gw2,
                                  float* gb2, float* dA1, int64_t ldd, int B, int Hd) {
    GM_CHECK_ARG(H && T && da2 && w2 && gw2 && gb2 && dA1 && B > 0 && Hd > 0);
    hipLaunchKernelGGL(dragan_head_bwd_kernel, dim3((Hd + HB_COLS - 1) / HB_COLS), dim3(1024), 0,
                       (hipStream_t)stream, H, ldh, T, ldt, da2, w2, gw2, gb2, dA1, ldd, B, Hd, 1);
    GM_LAUNCH_RET();
}

// Fisher GAN, folded critic step: lambda's successor (left in aux[5] by head workgroup 0 of the launch in which every
// workgroup read aux[0]) becomes lambda.  A launch of its own: the only ordering a late-starting workgroup respects.
__global__ void fisher_commit_kernel(float* aux) { aux[0] = aux[5]; }

extern "C" int gm_fisher_commit(void* stream, float* aux) {
    GM_CHECK_ARG(aux);
    hipLaunchKernelGGL(fisher_commit_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, aux);
    GM_LAUNCH_RET();
}

extern "C" int gm_dragan_head_bwd_store(void* stream, const float* H, int64_t ldh, const float* T,
                                        int64_t ldt, const float* da2, const float* w2, float* gw2,
                                        float* gb2, float* dA1, int64_t ldd, int B, int Hd) {
    GM_CHECK_ARG(H && T && da2 && w2 && gw2 && gb2 && dA1 && B > 0 && Hd > 0);
    hipLaunchKernelGGL(dragan_head_bwd_kernel, dim3((Hd + HB_COLS - 1) / HB_COLS), dim3(1024), 0,
                       (hipStream_t)stream, H, ldh, T, ldt, da2, w2, gw2, gb2, dA1, ldd, B, Hd, 0);
    GM_LAUNCH_RET();
}
