// gm_ops.hip -- non-GEMM kernels of the GAN/VAE step: batch gather, adversarial losses with their
// score gradients, flat Adam (+WGAN clamp), the per-graph tick, and HIP-graph / event helpers.
#include "gm_common.h"
#include "gm_gather.h"
#include "gm_stage.h"

#include <string>

// ------------------------------------------------------------------------------------------
// library bookkeeping
// ------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
extern "C" void gm_set_error(const char* msg) { g_last_error = msg ? msg : ""; }
extern "C" const char* gm_last_error(void) { return g_last_error.c_str(); }
extern "C" int gm_version(void) { return 100; }
extern "C" const char* gm_arch(void) { return "gfx950"; }

// ------------------------------------------------------------------------------------------
// tick
// ------------------------------------------------------------------------------------------
__global__ void tick_kernel(int64_t* ctr, int64_t inc) { *ctr += inc; }

extern "C" int gm_tick(void* stream, int64_t* ctr, int64_t inc) {
    GM_CHECK_ARG(ctr);
    hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, ctr, inc);
    GM_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------
// Stage-in: the host replays the reference's RNG protocol (csrc/gm_hostrng.cpp) straight into PINNED
// host rings that mirror the device rings slot for slot; the first kernel of every captured graph
// pulls the slots of its own iterations over PCIe (each byte read exactly once, 16 B per lane) into
// the device rings that all other kernels read.  This replaces hipMemcpyAsync uploads (three API calls
// + DMA start-up per sub-chunk, 60-100 us at the head of a short run) by one ~2 us launch inside
// the graph; the slot is resolved from the device counter like every other per-step address.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stage_in_kernel(StageP p) {
    if (p.publish && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *p.publish = gm_slot_index(p.it_slot);
    stage_copy(p);
}

// The fill gate of a stage-in (gm_stage_in_gated): ONE wave, its own launch in front of the copy on the same stream,
// polls the host's fill counter -- one word of pinned host memory.  Polled from every workgroup of the copy it cost a
// serialized PCIe read per workgroup: ~1 us per iteration staged (tools/piece_cost_probe.py, round 5: a graph launch of
// k iterations 25 + 68.7 k us with the poll inside the copy, 22 + 67.5 k us with this kernel); and workgroups that
// wait inside the copy sit on CUs for as long as the host takes to draw (round 4: bs=1024 step 145 -> 163 us).
__global__ __launch_bounds__(64) void stage_gate_wait_kernel(const int64_t* gate, gm_slot it_slot, int n_iters,
                                                             uint64_t timeout) {
    if (threadIdx.x != 0) return;
    stage_gate_wait(gate, gm_slot_index(it_slot) + n_iters, timeout);
}

static int stage_in_impl(void* stream, const gm_stage_seg* segs, int n_segs, gm_slot slot, int n_iters,
                         const int64_t* gate, gm_slot it_slot, double timeout_s, int64_t* publish = nullptr,
                         int max_blocks = 256) {
    GM_CHECK_ARG(segs && n_segs > 0 && n_segs <= GM_STAGE_MAX_SEGS && n_iters > 0);
    StageP p{};
    int64_t most = 0;
    for (int i = 0; i < n_segs; ++i) {
        GM_CHECK_ARG(segs[i].src && segs[i].dst && segs[i].bytes_per_iter > 0 && segs[i].bytes_per_iter % 4 == 0);
        const int m = segs[i].blocks > 1 ? segs[i].blocks : 1;
        GM_CHECK_ARG(segs[i].blocks >= 0 && segs[i].bytes_per_iter % (4 * m) == 0);
        GM_CHECK_ARG(segs[i].src_block_stride >= 0 && segs[i].dst_block_stride >= 0 &&
                     segs[i].src_block_stride % 4 == 0 && segs[i].dst_block_stride % 4 == 0);
        GM_CHECK_ARG(!segs[i].src_block_stride || segs[i].src_block_stride >= segs[i].bytes_per_iter / m);
        GM_CHECK_ARG(!segs[i].dst_block_stride || segs[i].dst_block_stride >= segs[i].bytes_per_iter / m);
        p.seg[i] = segs[i];
        if (segs[i].bytes_per_iter > most) most = segs[i].bytes_per_iter;
    }
    p.n_segs = n_segs; p.slot = slot; p.n_iters = n_iters;
    p.it_slot = it_slot; p.publish = publish;
    int64_t blocks = (most * n_iters / 16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > max_blocks) blocks = max_blocks;
    if (gate)                                             // wall_clock64(): 100 MHz
        hipLaunchKernelGGL(stage_gate_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, gate, it_slot, n_iters,
                           (uint64_t)(timeout_s * 1e8));
    hipLaunchKernelGGL(stage_in_kernel, dim3((unsigned)blocks, (unsigned)n_segs), dim3(256), 0,
                       (hipStream_t)stream, p);
    GM_LAUNCH_RET();
}

extern "C" int gm_stage_in(void* stream, const gm_stage_seg* segs, int n_segs, gm_slot slot, int n_iters) {
    return stage_in_impl(stream, segs, n_segs, slot, n_iters, nullptr, gm_slot{}, 0.0);
}

extern "C" int gm_stage_in_gated(void* stream, const gm_stage_seg* segs, int n_segs, gm_slot slot,
                                 int n_iters, const int64_t* gate, gm_slot it_slot, double timeout_s,
                                 int64_t* publish, int max_blocks) {
    GM_CHECK_ARG(gate && timeout_s > 0.0 && timeout_s < 3600.0 && max_blocks >= 1);
    return stage_in_impl(stream, segs, n_segs, slot, n_iters, gate, it_slot, timeout_s, publish,
                         max_blocks > 256 ? 256 : max_blocks);
}

// Device-side address of pinned host memory (hipHostMalloc / torch pin_memory()).
extern "C" int gm_host_device_ptr(void* host_ptr, void** dev_ptr_out) {
    GM_CHECK_ARG(host_ptr && dev_ptr_out);
    hipError_t e = hipHostGetDevicePointer(dev_ptr_out, host_ptr, 0);
    if (e != hipSuccess) { gm_set_error(hipGetErrorString(e)); return -(int)e; }
    return 0;
}

// ------------------------------------------------------------------------------------------
// K1 gather: out[b,:] = data[idx[b],:]   (process_batch, ns_gan.py:222-226)
// One 3136-byte image row per wave: 196 float4 -> lanes issue coalesced 16-B loads.
// ------------------------------------------------------------------------------------------
// The body lives in gm_gather.h: the generator's first forward GEMM can carry the gather workgroups
// in its own grid (gm_linear_fwd_gather, gm_gemm.hip).
extern "C" int gm_gather_rows(void* stream, const float* data, int64_t n_rows, const int64_t* idx,
                              gm_slot idx_slot, float* out, int64_t ld_out, int B, int row_elems) {
    GatherP g{};
    const int rc = gm_gather_fill(data, n_rows, idx, idx_slot, out, ld_out, B, row_elems, &g);
    if (rc) return rc;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(gm_gather_blocks(g, 4)), dim3(256), 0,
                       (hipStream_t)stream, g);
    GM_LAUNCH_RET();
}

extern "C" int gm_gather_rows_bits(void* stream, const uint32_t* bits, int words_per_row, int64_t n_rows,
                                   const int64_t* idx, gm_slot idx_slot, float* out, int64_t ld_out,
                                   int B, int row_elems) {
    GatherP g{};
    const int rc = gm_gather_fill_bits(bits, words_per_row, n_rows, idx, idx_slot, out, ld_out, B, row_elems, &g);
    if (rc) return rc;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(gm_gather_blocks(g, 4)), dim3(256), 0,
                       (hipStream_t)stream, g);
    GM_LAUNCH_RET();
}

extern "C" int gm_gather_rows_bits_packed(void* stream, const uint32_t* bits, int words_per_row, int64_t n_rows,
                                          const int64_t* idx, gm_slot idx_slot, uint32_t* out_bits, int B) {
    GatherP g{};
    const int rc = gm_gather_fill_bits_packed(bits, words_per_row, n_rows, idx, idx_slot, out_bits, B, &g);
    if (rc) return rc;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(gm_gather_blocks(g, 4)), dim3(256), 0,
                       (hipStream_t)stream, g);
    GM_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------
// K4 adversarial losses.  One 256-thread workgroup: the score vectors are [B] (B <= a few
// thousand), so a single workgroup with wave64 shuffle + LDS reductions is both the fastest
// shape and bit-deterministic.  Sums are carried in fp64 and rounded once.
// ------------------------------------------------------------------------------------------
namespace {

struct LossP {
    int variant, gen_mode, B, out_act;
    const float* sx;
    const float* sg;
    float hyper[8];
    float inv_b;
    float* loss_out;
    gm_slot loss_slot;
    float* dax;
    float* dag;
    float* aux;
    float* db;      // optional: d loss / d (head bias) = sum(dax) + sum(dag), fp64-accumulated
    // data parallel: the two losses that are not a mean of per-sample terms are
    // evaluated in phases around scalar all-reduces (gm_allreduce_scalars) of `pre`:
    //   RaGAN  1: pre[0] = sum_local sg            | 2: mean(sg) known: dax, pre[1] = sum_local du,
    //          pre[4] = sum_local dax, loss        | 3: sum(du) known: dag, db
    //   Fisher 1: pre[0..3] = local moment sums    | 2: everything else from the global moments
    // phase 0 = the whole loss in one launch (one GPU).  loss_scale: 1 on the rank that reports a
    // loss computed identically everywhere (Fisher), 0 on the others (losses are summed over ranks).
    int phase;
    float* pre;
    float loss_scale;
};

__device__ double block_sum(double v, double* sh) {
    v = gm_wave_sum_d(v);
    __syncthreads();                       // protect sh from the previous use
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}



__global__ __launch_bounds__(256) void gan_loss_kernel(LossP p) {
    __shared__ double sh[4];
    const int t = threadIdx.x, B = p.B;
    const float ib = p.inv_b;
    const bool D = !p.gen_mode;
    double acc = 0.0;           // sum of per-sample loss terms (already signed)
    float extra = 0.f;          // variant-specific scalar folded into the loss at the end

    if (p.variant == GM_LOSS_RA && D) {
        // ra_gan.py:204-205  L = -mean(log(sig(sx - mean(sg)) + eps) + log(sig(1 - sg) + eps)) / 2
        float mg, sum_du, bx = 0.f;
        if (p.phase <= 1) {
            double s1 = 0.0;
            for (int i = t; i < B; i += 256) s1 += (double)p.sg[i];
            const double tot1 = block_sum(s1, sh);
            if (p.phase == 1) { if (t == 0) p.pre[0] = (float)tot1; return; }
            mg = (float)(tot1 * (double)ib);
        } else {
            mg = p.pre[0] * ib;                               // pre[0]: sum of sg over ALL ranks
        }
        if (p.phase <= 2) {
            double sq = 0.0, sbx = 0.0;
            for (int i = t; i < B; i += 256) {
                const float u = gm_sigmoid(p.sx[i] - mg);
                const float v = gm_sigmoid(1.f - p.sg[i]);
                acc += (double)(logf(u + EPS) + logf(v + EPS));
                const float gu = (-0.5f * ib) / (u + EPS);       // dL/du
                const float du = (gu * (1.f - u)) * u;           // through the inner sigmoid
                const float ax = act_grad(du, p.sx[i], p.out_act);
                p.dax[i] = ax;
                sbx += (double)ax;
                sq += (double)du;
            }
            sum_du = (float)block_sum(sq, sh);                // d/dmg = -sum_du ; dmg/dsg_j = 1/B
            const double tot = block_sum(acc, sh);
            bx = (float)block_sum(sbx, sh);
            if (t == 0) p.loss_out[gm_slot_index(p.loss_slot)] = -(float)(tot * (double)ib) / 2.f;
            if (p.phase == 2) { if (t == 0) { p.pre[1] = sum_du; p.pre[4] = bx; } return; }
        } else {
            sum_du = p.pre[1];                                // sum of du over ALL ranks
            bx = p.pre[4];
        }
        double sbg = 0.0;
        for (int i = t; i < B; i += 256) {
            const float v = gm_sigmoid(1.f - p.sg[i]);
            const float gv = (-0.5f * ib) / (v + EPS);
            const float dv = -((gv * (1.f - v)) * v);         // d(1 - sg)/dsg = -1
            const float ag = act_grad(dv - sum_du * ib, p.sg[i], p.out_act);
            p.dag[i] = ag;
            sbg += (double)ag;
        }
        const float bg = (float)block_sum(sbg, sh);
        if (t == 0 && p.db) p.db[0] = bx + bg;
        return;
    }

    if (p.variant == GM_LOSS_FISHER && D) {
        // fisher_gan.py:214-223; aux[0] = lambda (updated in place: :155-156), aux[1..4] = moments
        float m1x, m2x, m1g, m2g;
        if (p.phase <= 1) {
            double a1 = 0, a2 = 0, b1 = 0, b2 = 0;
            for (int i = t; i < B; i += 256) {
                const float x = p.sx[i], g = p.sg[i];
                a1 += x; a2 += (double)(x * x); b1 += g; b2 += (double)(g * g);
            }
            const double t1 = block_sum(a1, sh), t2 = block_sum(a2, sh), t3 = block_sum(b1, sh), t4 = block_sum(b2, sh);
            if (p.phase == 1) {
                if (t == 0) { p.pre[0] = (float)t1; p.pre[1] = (float)t2; p.pre[2] = (float)t3; p.pre[3] = (float)t4; }
                return;
            }
            m1x = (float)(t1 * (double)ib); m2x = (float)(t2 * (double)ib);
            m1g = (float)(t3 * (double)ib); m2g = (float)(t4 * (double)ib);
        } else {                                              // sums over ALL ranks
            m1x = p.pre[0] * ib; m2x = p.pre[1] * ib; m1g = p.pre[2] * ib; m2g = p.pre[3] * ib;
        }
        const float lam = p.aux[0], rho = p.hyper[0];
        const float omega = 1.f - (0.5f * m2x + 0.5f * m2g);
        const float loss = -((m1x - m1g) + lam * omega - (rho / 2.f) * (omega * omega));
        const float dO = -(lam - rho * omega);               // dL/dOmega
        double sbx = 0.0, sbg = 0.0;
        for (int i = t; i < B; i += 256) {
            // dOmega/dsx_i = -0.5 * 2 * sx_i / B
            const float ax = act_grad(-ib + dO * (-(p.sx[i] * ib)), p.sx[i], p.out_act);
            const float ag = act_grad(ib + dO * (-(p.sg[i] * ib)), p.sg[i], p.out_act);
            p.dax[i] = ax; p.dag[i] = ag;
            sbx += (double)ax; sbg += (double)ag;
        }
        const float bx = (float)block_sum(sbx, sh), bg = (float)block_sum(sbg, sh);
        __syncthreads();
        if (t == 0) {
            if (p.db) p.db[0] = bx + bg;
            p.loss_out[gm_slot_index(p.loss_slot)] = loss * p.loss_scale;
            p.aux[0] = lam + rho * (-omega);                  // lambda += rho * lambda.grad
            p.aux[1] = m1x; p.aux[2] = m1g; p.aux[3] = m2x; p.aux[4] = m2g;
        }
        return;
    }

    // separable variants: per-sample term + per-sample gradient
    double sbx = 0.0, sbg = 0.0;
    for (int i = t; i < B; i += 256) {
        const float g = p.sg[i];
        const float x = D ? p.sx[i] : 0.5f;
        float lx, lg, dx, dg;
        sample_terms(p.variant, D, x, g, ib, p.hyper, lx, lg, dx, dg);
        // WGAN-GP: + lambda * mean((||grad|| - 1)^2), per-row terms in aux
        // (w_gp_gan.py:215-218; rows produced by gm_gp_norm)
        // (DRAGAN carries the same kind of penalty rows on the NS loss, dra_gan.py:220-223: any
        // separable variant with aux rows gets the term, like the fused head kernel)
        if (D && p.aux) lx += p.hyper[7] * p.aux[i];
        acc += (double)lx + (double)lg;
        if (D && p.dax) { const float ax = act_grad(dx, x, p.out_act); p.dax[i] = ax; sbx += (double)ax; }
        if (p.dag) { const float ag = act_grad(dg, g, p.out_act); p.dag[i] = ag; sbg += (double)ag; }
    }
    (void)extra;
    const double tot = block_sum(acc, sh);
    // head-bias gradient: the two halves are summed separately (as autograd accumulates the
    // D(x) and D(G(z)) paths) in fp64, so an exactly-cancelling gradient stays exactly zero
    // (Adam would amplify a 1e-9 residue into an O(0.1*lr) step).
    const float bx = (float)block_sum(sbx, sh), bg = (float)block_sum(sbg, sh);
    if (t == 0) {
        p.loss_out[gm_slot_index(p.loss_slot)] = (float)(tot * (double)ib);
        if (p.db) p.db[0] = bx + bg;
    }
}

}  // namespace

static int gan_loss_impl(void* stream, int variant, int gen_mode, const float* sx,
                         const float* sg, int B, int out_act, const float* hyper, int n_hyper,
                         float inv_b, float* loss_out, gm_slot loss_slot, float* dax, float* dag,
                         float* aux_io, float* db_out, int phase, float* pre, float loss_scale);

extern "C" int gm_gan_loss(void* stream, int variant, int gen_mode, const float* sx,
                           const float* sg, int B, int out_act, const float* hyper, int n_hyper,
                           float inv_b, float* loss_out, gm_slot loss_slot, float* dax, float* dag,
                           float* aux_io, float* db_out) {
    return gan_loss_impl(stream, variant, gen_mode, sx, sg, B, out_act, hyper, n_hyper, inv_b, loss_out,
                         loss_slot, dax, dag, aux_io, db_out, 0, nullptr, 1.0f);
}

extern "C" int gm_gan_loss_phase(void* stream, int variant, int gen_mode, const float* sx,
                                 const float* sg, int B, int out_act, const float* hyper, int n_hyper,
                                 float inv_b, float* loss_out, gm_slot loss_slot, float* dax, float* dag,
                                 float* aux_io, float* db_out, int phase, float* pre, float loss_scale) {
    GM_CHECK_ARG(phase >= 1 && phase <= 3 && pre);
    GM_CHECK_ARG(!gen_mode && (variant == GM_LOSS_RA || variant == GM_LOSS_FISHER));
    return gan_loss_impl(stream, variant, gen_mode, sx, sg, B, out_act, hyper, n_hyper, inv_b, loss_out,
                         loss_slot, dax, dag, aux_io, db_out, phase, pre, loss_scale);
}

static int gan_loss_impl(void* stream, int variant, int gen_mode, const float* sx,
                         const float* sg, int B, int out_act, const float* hyper, int n_hyper,
                         float inv_b, float* loss_out, gm_slot loss_slot, float* dax, float* dag,
                         float* aux_io, float* db_out, int phase, float* pre, float loss_scale) {
    GM_CHECK_ARG(sg && loss_out && B > 0 && n_hyper >= 0 && n_hyper <= 8);
    GM_CHECK_ARG(variant >= GM_LOSS_NS && variant <= GM_LOSS_F_JS);
    GM_CHECK_ARG(gen_mode || (sx && dax && dag));
    GM_CHECK_ARG(!(variant == GM_LOSS_FISHER && !gen_mode) || aux_io);
    LossP p{};
    p.variant = variant; p.gen_mode = gen_mode; p.B = B; p.out_act = out_act;
    p.sx = sx; p.sg = sg; p.inv_b = inv_b;
    for (int i = 0; i < n_hyper; ++i) p.hyper[i] = hyper[i];
    p.loss_out = loss_out; p.loss_slot = loss_slot; p.dax = dax; p.dag = dag; p.aux = aux_io;
    p.db = db_out; p.phase = phase; p.pre = pre; p.loss_scale = loss_scale;
    hipLaunchKernelGGL(gan_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, p);
    GM_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------
// K7 Adam on a flat buffer, arithmetic order of torch _single_tensor_adam:
//   g = grad + wd*p ; m += (1-b1)*(g-m) ; v = v*b2 + (1-b2)*g*g ;
//   denom = sqrt(v)/bc2_sqrt + eps ; p += (-step_size) * m / denom ; optional clamp (w_gan.py:241)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p,
                                                  const float* __restrict__ g,
                                                  float* __restrict__ m, float* __restrict__ v,
                                                  int64_t n, const float* __restrict__ sched,
                                                  gm_slot sched_slot, float one_minus_b1, float b2,
                                                  float one_minus_b2, float eps, float wd,
                                                  float clamp, const float* __restrict__ lr_scale) {
    const int64_t si = gm_slot_index(sched_slot);
    // lr_scale (BEGAN's ReduceLROnPlateau): a power of two, so scaling lr/bc1 is exact
    const float step_size = sched[2 * si] * (lr_scale ? lr_scale[0] : 1.0f), bc2_sqrt = sched[2 * si + 1];
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        adam_update(pp, gg, mm, vv, step_size, bc2_sqrt, one_minus_b1, b2, one_minus_b2, eps, wd,
                    clamp);
    };
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 P = reinterpret_cast<float4*>(p)[i];
        const float4 G = reinterpret_cast<const float4*>(g)[i];
        float4 M = reinterpret_cast<float4*>(m)[i];
        float4 V = reinterpret_cast<float4*>(v)[i];
        upd(P.x, G.x, M.x, V.x); upd(P.y, G.y, M.y, V.y);
        upd(P.z, G.z, M.z, V.z); upd(P.w, G.w, M.w, V.w);
        reinterpret_cast<float4*>(p)[i] = P;
        reinterpret_cast<float4*>(m)[i] = M;
        reinterpret_cast<float4*>(v)[i] = V;
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        upd(p[i], g[i], m[i], v[i]);
}

static int adam_impl(void* stream, float* p, const float* g, float* m, float* v, int64_t n,
                     const float* sched, gm_slot sched_slot, double beta1, double beta2, double eps,
                     double weight_decay, float clamp, const float* lr_scale);

extern "C" int gm_adam(void* stream, float* p, const float* g, float* m, float* v, int64_t n,
                       const float* sched, gm_slot sched_slot, double beta1, double beta2,
                       double eps, double weight_decay, float clamp) {
    return adam_impl(stream, p, g, m, v, n, sched, sched_slot, beta1, beta2, eps, weight_decay, clamp,
                     nullptr);
}

extern "C" int gm_adam_scaled(void* stream, float* p, const float* g, float* m, float* v, int64_t n,
                              const float* sched, gm_slot sched_slot, double beta1, double beta2,
                              double eps, double weight_decay, float clamp, const float* lr_scale) {
    GM_CHECK_ARG(lr_scale);
    return adam_impl(stream, p, g, m, v, n, sched, sched_slot, beta1, beta2, eps, weight_decay, clamp,
                     lr_scale);
}

static int adam_impl(void* stream, float* p, const float* g, float* m, float* v, int64_t n,
                     const float* sched, gm_slot sched_slot, double beta1, double beta2, double eps,
                     double weight_decay, float clamp, const float* lr_scale) {
    GM_CHECK_ARG(p && g && m && v && sched && n > 0);
    GM_CHECK_ARG(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) |
                   reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0);
    // python: 1 - beta1 evaluated in double, then cast to the op's fp32 scalar
    const float omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                       sched, sched_slot, omb1, (float)beta2, omb2, (float)eps, (float)weight_decay,
                       clamp, lr_scale);
    GM_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------
// ring slot <-> plain tensor copy (captured general path: a user's train_D / train_G read ordinary tensors,
// the graph that replays them walks prefetched rings through the device counter)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_slot_kernel(const float* __restrict__ src, gm_slot src_slot,
                                                       float* __restrict__ dst, gm_slot dst_slot, int64_t n) {
    const float* s = src + gm_slot_offset(src_slot);
    float* d = dst + gm_slot_offset(dst_slot);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] = s[i];
}

extern "C" int gm_copy_slot_f32(void* stream, const float* src, gm_slot src_slot, float* dst, gm_slot dst_slot,
                                int64_t n) {
    GM_CHECK_ARG(src && dst && n > 0);
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(copy_slot_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, src_slot, dst,
                       dst_slot, n);
    GM_LAUNCH_RET();
}

// ------------------------------------------------------------------------------------------
// activation backward (general autograd path)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dY,
                                                     const float* __restrict__ Y,
                                                     float* __restrict__ dA, int64_t n, int act) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dA[i] = act_grad(dY[i], Y[i], act);
}

extern "C" int gm_act_bwd(void* stream, const float* dY, const float* Y, float* dA, int64_t n,
                          int act) {
    GM_CHECK_ARG(dY && Y && dA && n > 0);
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dY, Y, dA,
                       n, act);
    GM_LAUNCH_RET();
}

// (gm_randperm_prefix, the O(B) prefix of torch.randperm(n), lives in gm_hostrng.cpp)

// ------------------------------------------------------------------------------------------
// HOST: advance a serialized torch CPU generator (mt19937) by n 32-bit outputs WITHOUT producing
// them.  Data-parallel ranks replay the global draw protocol but only materialise their own rows of
// each noise tensor; the other ranks' rows are skipped here (no tempering, no float math: one
// twist per 624 outputs).  Layout = at::CPUGeneratorImplState (CPUGeneratorImpl.cpp): the legacy
// POD {u64 seed; i32 left; i32 seeded; u64 next; u64 state[624]; 3 x f64; i32 valid} + {f32; bool}.
// ------------------------------------------------------------------------------------------
namespace {
struct TorchCpuGenState {
    uint64_t the_initial_seed;
    int32_t left;
    int32_t seeded;
    uint64_t next;
    uint64_t state[624];
    double normal_x, normal_y, normal_rho;
    int32_t normal_is_valid;
    float next_float_normal_sample;
    bool is_next_float_normal_sample_valid;
};

inline uint32_t mt_tw(uint64_t u, uint64_t v) {
    const uint32_t y = ((uint32_t)u & 0x80000000u) | ((uint32_t)v & 0x7fffffffu);
    return (y >> 1) ^ (((uint32_t)v & 1u) ? 0x9908b0dfu : 0u);
}

// at::mt19937::next_state(): the three-segment form (no modulo).  Dependence distances are 397 and
// 227 elements and s[k+1] is read before it is written, so any SIMD width is safe; an AVX2 clone
// is picked at run time (0.33 vs 0.55 us per twist).
#define GM_MT_NEXT_STATE_BODY                                                                      \
    _Pragma("clang loop vectorize(assume_safety)")                                                 \
    for (int k = 0; k < 624 - 397; ++k) s[k] = (uint32_t)s[k + 397] ^ mt_tw(s[k], s[k + 1]);       \
    _Pragma("clang loop vectorize(assume_safety)")                                                 \
    for (int k = 624 - 397; k < 623; ++k) s[k] = (uint32_t)s[k + 397 - 624] ^ mt_tw(s[k], s[k + 1]); \
    s[623] = (uint32_t)s[396] ^ mt_tw(s[623], s[0]);
__attribute__((target("avx2"))) void mt_next_state_avx2(uint64_t* s) { GM_MT_NEXT_STATE_BODY }
inline void mt_next_state_base(uint64_t* s) { GM_MT_NEXT_STATE_BODY }
#undef GM_MT_NEXT_STATE_BODY
inline void mt_next_state(uint64_t* s) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) mt_next_state_avx2(s); else mt_next_state_base(s);
}
}  // namespace

extern "C" int gm_mt19937_skip(void* torch_cpu_rng_state, int64_t state_bytes, uint64_t n) {
    GM_CHECK_ARG(torch_cpu_rng_state && state_bytes == (int64_t)sizeof(TorchCpuGenState));
    TorchCpuGenState* g = reinterpret_cast<TorchCpuGenState*>(torch_cpu_rng_state);
    GM_CHECK_ARG(g->seeded == 1 && g->left >= 1 && g->left <= 624 && g->next <= 624);
    // at::mt19937::operator(): if (--left == 0) next_state();  y = state[next++];
    while (n > 0) {
        const uint64_t avail = (uint64_t)(g->left - 1);       // outputs before the next twist
        if (avail == 0) {
            mt_next_state(g->state);
            g->left = 624;
            g->next = 1;                                      // this call's output was state[0]
            n -= 1;
            continue;
        }
        const uint64_t take = avail < n ? avail : n;
        g->left -= (int32_t)take;
        g->next += take;
        n -= take;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// HIP graph + event helpers
// ------------------------------------------------------------------------------------------
#define GM_HIP(call)                                              \
    do {                                                          \
        hipError_t e__ = (call);                                  \
        if (e__ != hipSuccess) {                                  \
            gm_set_error(hipGetErrorString(e__));                 \
            return -(int)e__;                                     \
        }                                                         \
    } while (0)

extern "C" int gm_graph_begin(void* stream) {
    GM_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return 0;
}
extern "C" int gm_graph_end(void* stream, void** graph_exec_out) {
    GM_CHECK_ARG(graph_exec_out);
    hipGraph_t graph = nullptr;
    GM_HIP(hipStreamEndCapture((hipStream_t)stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) { gm_set_error(hipGetErrorString(e)); return -(int)e; }
    *graph_exec_out = (void*)exec;
    return 0;
}
extern "C" int gm_graph_launch(void* graph_exec, void* stream) {
    GM_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return 0;
}
extern "C" int gm_graph_destroy(void* graph_exec) {
    if (graph_exec) GM_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return 0;
}

// Shader-clock probe: runs a dependent MFMA chain for `iters` instructions in every wave of a
// full grid and reports shader cycles (s_memtime) and the 100 MHz wall clock spanned by block 0,
// so bench.py can state the effective engine clock the roofline is quoted against.
typedef float f32x16_probe __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void clock_probe_kernel(int iters, unsigned long long* out, float* sink) {
    f32x16_probe acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float a = (float)threadIdx.x, b = 1.0f;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (acc[0] == 123.456f) sink[0] = acc[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}

extern "C" int gm_clock_probe(void* stream, int iters, unsigned long long* out2, float* sink) {
    GM_CHECK_ARG(out2 && sink && iters > 0);
    hipLaunchKernelGGL(clock_probe_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, iters, out2,
                       sink);
    GM_LAUNCH_RET();
}

extern "C" int gm_stream_create(void** stream_out) {
    GM_CHECK_ARG(stream_out);
    hipStream_t st;
    GM_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *stream_out = (void*)st;
    return 0;
}
extern "C" int gm_stream_destroy(void* stream) {
    if (stream) GM_HIP(hipStreamDestroy((hipStream_t)stream));
    return 0;
}
extern "C" int gm_stream_wait_event(void* stream, void* ev) {
    GM_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0));
    return 0;
}

extern "C" int gm_event_create(void** ev_out) {
    GM_CHECK_ARG(ev_out);
    hipEvent_t ev;
    GM_HIP(hipEventCreate(&ev));
    *ev_out = (void*)ev;
    return 0;
}
extern "C" int gm_event_record(void* ev, void* stream) {
    GM_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return 0;
}
extern "C" int gm_event_sync(void* ev) {
    GM_HIP(hipEventSynchronize((hipEvent_t)ev));
    return 0;
}
extern "C" int gm_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out) {
    GM_CHECK_ARG(ms_out);
    GM_HIP(hipEventElapsedTime(ms_out, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
    return 0;
}
extern "C" int gm_event_destroy(void* ev) {
    if (ev) GM_HIP(hipEventDestroy((hipEvent_t)ev));
    return 0;
}
