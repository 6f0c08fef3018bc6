/*
 * gm_hip.h -- C-ABI of libgm_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the hot
 * path of shayneobrien/generative-models (GAN Trainer.train/train_D/train_G, VAE compute_batch).
 *
 * The reference has no FFI of its own: its hot path bottoms out in
 * PyTorch ops.  Each entry point below therefore cites the reference call site whose torch op(s)
 * it replaces.  Conventions (all entry points):
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call is
 *     asynchronous on it and allocation-free, so it can be captured into a hipGraph;
 *   - all pointers are raw DEVICE pointers owned by the caller (PyTorch), fp32 unless noted,
 *     matrices row-major with explicit leading dimensions in elements;
 *   - return 0 on success, a negative hipError_t on a launch error, GM_EINVAL on bad arguments;
 *   - nothing here falls back to a CPU path.
 *
 * "slot" arguments (ctr, mul, add, ring, stride): graph-replayable addressing of per-step data.
 * The effective pointer is  base + ((ctr ? *ctr : 0) * mul + add) % ring * stride  (ring <= 0
 * means no modulo).  `ctr` is a device int64 advanced by gm_tick() once per replayed graph, so a
 * captured graph walks through prefetched index / noise rings and the Adam step table without
 * host involvement.
 */
#ifndef GM_HIP_H
#define GM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GM_EINVAL (-10001)

/* activations (Generator/Discriminator.forward: ns_gan.py:43-46,57-60; w_gp_gan.py:59-62;
 * be_gan.py:73-76) */
enum { GM_ACT_ID = 0, GM_ACT_RELU = 1, GM_ACT_SIGMOID = 2 };

/* GAN loss variants */
enum {
    GM_LOSS_NS = 0,      /* ns_gan.py:191-192,214   (also InfoGAN D/G, DRAGAN first-order, RaGAN G) */
    GM_LOSS_MM = 1,      /* mm_gan.py:215-216,235 */
    GM_LOSS_W = 2,       /* w_gan.py:208,227 ; w_gp_gan.py:218 (first-order part),237 */
    GM_LOSS_LS = 3,      /* ls_gan.py:192-193,213 */
    GM_LOSS_RA = 4,      /* ra_gan.py:204-205 (D) ; :227 (G = NS) */
    GM_LOSS_FISHER = 5,  /* fisher_gan.py:214-223 (D) ; :246 (G) */
    GM_LOSS_F_TV = 6, GM_LOSS_F_FKL = 7, GM_LOSS_F_RKL = 8, GM_LOSS_F_PEARSON = 9,
    GM_LOSS_F_HELLINGER = 10, GM_LOSS_F_JS = 11   /* f_gan.py:99-142 */
};

typedef struct gm_slot {
    const int64_t* ctr;   /* device counter or NULL */
    int32_t mul, add, ring;
    int64_t stride;       /* elements */
} gm_slot;

/* ---- library ------------------------------------------------------------------------- */
int gm_version(void);
const char* gm_arch(void);                 /* "gfx950" */
const char* gm_last_error(void);

/* ---- per-graph tick: *ctr += inc (one thread).  Replaces the Python loop counters of
 * Trainer.train (ns_gan.py:117-126). */
int gm_tick(void* stream, int64_t* ctr, int64_t inc);

/* ---- dst[dst_slot + i] = src[src_slot + i], i < n (fp32 words).  The README extension contract at speed
 * (/root/reference/README.md:29-65): a user's train_D / train_G read ordinary tensors -- compute_noise's result
 * (ns_gan.py:218-220) -- and return a 0-dim loss; the graph that replays them copies the iteration's noise out of
 * the prefetched ring and its loss into the per-step history (the .item() of ns_gan.py:142,154) with this. */
int gm_copy_slot_f32(void* stream, const float* src, gm_slot src_slot, float* dst, gm_slot dst_slot, int64_t n);

/* ---- K1: batch gather.  Replaces DataLoader collate in process_batch (ns_gan.py:222-226):
 * out[b,:] = data[idx[b],:].  idx is int64 [B] at slot `idx_slot`. */
int gm_gather_rows(void* stream, const float* data, int64_t n_rows, const int64_t* idx,
                   gm_slot idx_slot, float* out, int64_t ld_out, int B, int row_elems);

/* ---- K2/K3: Y[M,N] = act(X[M,K] * W[N,K]^T + bias[N]).  Replaces nn.Linear + F.relu /
 * torch.sigmoid (ns_gan.py:44-45,58-59).  X may live in a ring (noise: ns_gan.py:218-220). */
int gm_linear_fwd(void* stream, const float* X, int64_t ldx, gm_slot x_slot, const float* W,
                  const float* bias, float* Y, int64_t ldy, int M, int K, int N, int act);

/* ---- K5/K6: dX[M,K] = dA[M,N] * W[N,K], then times act'(below) where `below` is the output
 * of the layer that produced X (epi: GM_ACT_ID none, GM_ACT_RELU mask below>0,
 * GM_ACT_SIGMOID below*(1-below)).  Replaces autograd's AddmmBackward + Relu/SigmoidBackward
 * inside Tensor.backward() (ns_gan.py:138,155). */
int gm_linear_bwd_dx(void* stream, const float* dA, int64_t lda, const float* W, float* dX,
                     int64_t ldx, const float* below, int64_t ld_below, int M, int K, int N,
                     int epi);

/* ---- K5: dW[N,K] (=|+=) dA[M,N]^T * X[M,K];  db[N] (=|+=) sum_m dA[m,:] (db may be NULL).
 * Replaces autograd's weight/bias gradient accumulation (ns_gan.py:138,155). */
int gm_linear_bwd_dw(void* stream, const float* dA, int64_t lda, const float* X, int64_t ldx,
                     gm_slot x_slot, float* dW, float* db, int M, int K, int N, int accumulate);

/* gm_linear_bwd_dw with the optimizer folded into the gradient epilogue: dW/db are written as
 * usual and Adam (same arithmetic as gm_adam) is applied to (pW,mW,vW)/(pb,mb,vb)
 * by the thread that produced the gradient element -- optim.Adam.step (ns_gan.py:139,156) without
 * its own launch.  Single-GPU fast path only (under data parallelism the all-reduce sits between
 * gradient and optimizer). */
int gm_linear_bwd_dw_adam(void* stream, const float* dA, int64_t lda, const float* X, int64_t ldx,
                          gm_slot x_slot, float* dW, float* db, int M, int K, int N, float* pW,
                          float* mW, float* vW, float* pb, float* mb, float* vb, const float* sched,
                          gm_slot sched_slot, double beta1, double beta2, double eps,
                          double weight_decay, float clamp);

/* ---- K4: adversarial loss + its gradient w.r.t. the critic's PRE-activation output.
 * sx,sg: [B] post-activation scores D(x), D(G(z)) (sx NULL in generator mode).
 * out_act: activation that produced the scores (sigmoid, or relu for WGAN-GP).
 * hyper: host array of up to 8 floats (LS: a,b,c ; Fisher: rho ; [7]: weight of the penalty rows).
 * inv_b: fp32 1/B used for every mean and its gradient (1/B_global under data parallelism).
 * loss_out[slot]: scalar loss.  dax/dag: [B] gradients (dax NULL in generator mode).
 * aux_io: device state/extra terms (Fisher: lambda, moments; WGAN-GP: penalty rows) or NULL.
 * db_out: optional [1] gradient of the critic's output bias (sum of dax + sum of dag, each half
 *   accumulated in fp64 so that exactly-cancelling gradients stay exactly zero).
 * Replaces the torch elementwise/mean ops in train_D / train_G (appendix A.2 table). */
int gm_gan_loss(void* stream, int variant, int gen_mode, const float* sx, const float* sg, int B,
                int out_act, const float* hyper, int n_hyper, float inv_b, float* loss_out,
                gm_slot loss_slot, float* dax, float* dag, float* aux_io, float* db_out);
/* Data-parallel form of the two critic losses that are not a mean of per-sample terms:
 * RaGAN's mean(D(G(z))) inside the sigmoid (ra_gan.py:204) and Fisher's moments / lambda ascent
 * (fisher_gan.py:214-223,155-156) span the GLOBAL batch.  The loss runs in phases around
 * gm_allreduce_scalars of `pre` (device float[8]): Ra 1 | exchange pre[0] | 2 | exchange pre[1] | 3;
 * Fisher 1 | exchange pre[0..3] | 2.  loss_scale: 1 on the rank that reports a loss every rank
 * computes identically (Fisher), 0 elsewhere -- per-rank loss slots are summed over ranks. */
int gm_gan_loss_phase(void* stream, int variant, int gen_mode, const float* sx, const float* sg, int B,
                      int out_act, const float* hyper, int n_hyper, float inv_b, float* loss_out,
                      gm_slot loss_slot, float* dax, float* dag, float* aux_io, float* db_out,
                      int phase, float* pre, float loss_scale);

/* ---- K7 (+K8): Adam over one flat parameter buffer, exactly torch's _single_tensor_adam
 * (SURVEY.md section 3.5).  sched: device float2 table {step_size = lr/bc1, bc2_sqrt} indexed by
 * slot (the optimizer step number); clamp > 0 applies p = clamp(p, -clamp, clamp) afterwards
 * (w_gan.py:241-243).  betas/eps/weight_decay arrive as the Python doubles torch receives;
 * 1-beta is formed in double and then cast to fp32, like torch's scalar operands.
 * Replaces optim.Adam.step (ns_gan.py:139,156). */
int gm_adam(void* stream, float* p, const float* g, float* m, float* v, int64_t n,
            const float* sched, gm_slot sched_slot, double beta1, double beta2, double eps,
            double weight_decay, float clamp);

/* ---- K9: WGAN-GP interpolation  x_hat = eps*x + (1-eps)*G(z), eps [B] in a ring
 * (w_gp_gan.py:197-201). */
int gm_interp(void* stream, const float* eps, gm_slot eps_slot, const float* x, int64_t ldx,
              const float* g, int64_t ldg, float* out, int64_t ldo, int B, int I);

/* ---- K10 prologue: u[b,n] = [s_b>0]*[h[b,n]>0]*w2[n]: the vector that autograd.grad(D(x_hat),
 * x_hat) (w_gp_gan.py:207-212) pushes through the ReLU critic; grad = u * W1 is then one
 * gm_linear_bwd_dx call. */
int gm_gp_u(void* stream, const float* s, const float* h, int64_t ldh, const float* w2, float* u,
            int64_t ldu, int B, int H);

/* ---- K11: per-row ||g||_2, penalty rows pen[b] = (n_b - k)^2 and
 * gamma = d(lambda*mean((n-k)^2))/dg (0 where n_b == 0)  (w_gp_gan.py:215, dra_gan.py:220). */
int gm_gp_norm(void* stream, const float* g, int64_t ldg, float* gamma, int64_t ldm, float* pen,
               float lambda, float inv_b, float k, int B, int I);

/* ---- K12 tail: gw2[n] += sum_b [s_b>0][h[b,n]>0]*t[b,n]  (second backward into w2; the dW1 term
 * is gm_linear_bwd_dw(u, gamma, accumulate) and t = gamma*W1^T is gm_linear_fwd). */
int gm_gp_dw2(void* stream, const float* s, const float* h, int64_t ldh, const float* t,
              int64_t ldt, float* gw2, int B, int H);
/* same, STORING the sum (out[n] = ...) instead of accumulating */
int gm_gp_dw2_store(void* stream, const float* s, const float* h, int64_t ldh, const float* t,
                    int64_t ldt, float* out, int B, int H);
/* K10 prologue in one launch: the critic's N = 1 output layer on x_hat and the vector the input
 * gradient starts from -- s[b] = relu(h[b,:].w2 + b2), u[b,n] = [s_b>0][h[b,n]>0] w2[n]
 * (w_gp_gan.py:202-212: D(x_hat) and autograd.grad's seed through the ReLU critic). */
int gm_head_gp(void* stream, const float* h, int64_t ldh, const float* w2, const float* b2, float* s,
               float* u, int64_t ldu, int B, int H);

/* ---- BIR-VAE (bir_vae.py:86-97, 180-221).  reparam: z = mu + eps with the
 * host-drawn numpy noise (scale = set_var, :92-94).  mmd: partial[m] = row m's share of
 * sum k(x,x) + sum k(z,z) - 2 sum k(x,z) with the Gaussian kernel exp(-mean_d((a-b)^2)/dim) over the
 * prior sample x = torch.randn(z.shape) (:203), and (dz != NULL) dz = d(lambda*mmd)/dz. */
int gm_bir_reparam(void* stream, const float* mu, int64_t ldmu, const float* eps, gm_slot eps_slot,
                   float* z, int64_t ldz, int B, int Z);
int gm_bir_mmd(void* stream, const float* z, int64_t ldz, const float* prior, gm_slot prior_slot,
               float* partial, float* dz, int64_t lddz, int B, int Z, float lambda);

/* ---- K14: VAE.  ml = [mu | log_var] (B x 2Z, the two encoder heads packed side by side).
 * reparam: z = mu + eps*exp(lv/2) (vae.py:100-106), kl_out[slot] = sum 0.5*(mu^2+exp(lv)-lv-1)
 * (vae.py:210-212).  reparam_bwd: d(recon+kl)/d[mu|lv] from dz.  sqerr: per-row sums of
 * (x-xr)^2 (vae.py:203) + gradient w.r.t. the decoder's pre-sigmoid output. */
int gm_vae_reparam(void* stream, const float* ml, int64_t ldml, const float* eps, gm_slot eps_slot,
                   float* z, int64_t ldz, float* kl_out, gm_slot kl_slot, int B, int Z);
/* Wide form of gm_vae_reparam: many workgroups, the KL term left as ceil(B*Z/256) per-workgroup partial
 * sums in kl_part[0..n_part) for gm_sum_finalize2_tick to add up. */
int gm_vae_reparam_wide(void* stream, const float* ml, int64_t ldml, const float* eps, gm_slot eps_slot,
                        float* z, int64_t ldz, float* kl_part, int n_part, int B, int Z);
/* gm_vae_reparam_wide AND the decoder's first layer H = act(z W^T + b) (W: [N, Z], Z <= 32, Z % 4 == 0) as ONE
 * launch: vae.py:100-106 + the `F.relu(self.linear(z))` of the decoder (:113).  The reparameterisation workgroups
 * store z and the KL partials exactly as gm_vae_reparam_wide does; the GEMM workgroups form z from (mu, log_var, eps)
 * themselves.  Bit-identical to gm_vae_reparam_wide + gm_linear_fwd. */
int gm_vae_reparam_fwd(void* stream, const float* ml, int64_t ldml, const float* eps, gm_slot eps_slot,
                       float* z, int64_t ldz, float* kl_part, int n_part, int B, int Z, const float* W,
                       const float* bias, float* H, int64_t ldh, int N, int act);
/* The two narrow GEMMs in the middle of the VAE's backward pass as ONE launch (round 4): dz = dHdec W_d1
 * (W_d1: [Hd, Z], decoder layer 1), d loss / d [mu | log_var] from dz exactly as gm_linear_bwd_dx_reparam's epilogue
 * forms it (dml: [B, 2Z], written), dHe = (dml W_ml) . [He > 0] (W_ml: [2Z, Hd], the encoder's mu / log_var layer).
 * One workgroup per 16 rows; same summation orders as gm_linear_bwd_dx_reparam followed by gm_linear_bwd_dx.
 * Z <= 32, Hd % 4 == 0.  dz itself is not stored. */
int gm_vae_bwd_mid(void* stream, const float* dHdec, int64_t lddh, const float* Wd1, const float* ml,
                   int64_t ldml, const float* eps, gm_slot eps_slot, float* dml, int64_t lddml,
                   const float* Wml, const float* He, int64_t ldhe, float* dHe, int64_t lddhe, int B, int Hd, int Z);
int gm_vae_reparam_bwd(void* stream, const float* ml, int64_t ldml, const float* eps,
                       gm_slot eps_slot, const float* dz, int64_t lddz, float* dml, int64_t ldd,
                       int B, int Z);
int gm_sqerr_sigmoid_bwd(void* stream, const float* x, int64_t ldx, const float* xr, int64_t ldr,
                         float* dA, int64_t lda, float* partial, int B, int I);
/* out[slot] = scale * sum(partial[0..n)) in a fixed order (deterministic loss reductions). */
int gm_sum_finalize(void* stream, const float* partial, int n, float scale, float* out,
                    gm_slot out_slot);
/* The same as the LAST launch of a step: also advances the device step counter `tick` by one (every
 * slot of the step has been resolved by then), saving the separate gm_tick launch. */
int gm_sum_finalize_tick(void* stream, const float* partial, int n, float scale, float* out,
                         gm_slot out_slot, int64_t* tick);
/* Two such sums in one launch (vae.py:203 and :212 of one batch), tick optional (may be NULL). */
int gm_sum_finalize2_tick(void* stream, const float* pa, int na, float scale_a, float* out_a, gm_slot slot_a,
                          const float* pb, int nb, float scale_b, float* out_b, gm_slot slot_b,
                          int64_t* tick);

/* ---- fused critic head (output_dim == 1, separable loss variants): replaces the N=1 GEMV
 * (`self.discriminate`, ns_gan.py:59), the loss lines of train_D / train_G (appendix A.2) and, in
 * the backward, the N=1 dW + K=1 dX launches.  H: hidden activations [R,Hd] (critic mode R=2B,
 * rows 0..B-1 real then B generated; generator mode R=B).  head_fwd_loss writes scores S[R],
 * dS[R] = d loss / d pre-activation, per-row loss terms and -- when dH is given -- dH = dS (x) w2
 * masked by H>0 while the row is hot; head_bwd writes gw2 = dS^T H, gb2 (fp64 half-sums), the loss
 * scalar, and dH when head_fwd_loss did not (its dH may be NULL; all-NULL outputs = scalars only). */
int gm_head_fwd_loss(void* stream, int variant, int gen_mode, const float* H, int64_t ldh,
                     const float* w2, const float* b2, int out_act, int B, int Hd,
                     const float* hyper, int n_hyper, float inv_b, const float* pen, float* S,
                     float* dS, float* rowloss, float* dH_or_null, int64_t lddh);
/* Same, and the LAST workgroup to finish also writes loss_out[slot] = inv_b * sum(rowloss) (fixed
 * order, fp64) and, when tick != NULL, advances the iteration counter: the generator step needs no
 * head_bwd launch.  done_ctr: one zero-initialised device word, re-armed by the kernel. */
int gm_head_fwd_loss_final(void* stream, int variant, int gen_mode, const float* H, int64_t ldh,
                           const float* w2, const float* b2, int out_act, int B, int Hd,
                           const float* hyper, int n_hyper, float inv_b, const float* pen, float* S,
                           float* dS, float* rowloss, float* dH_or_null, int64_t lddh,
                           float* loss_out, gm_slot loss_slot, unsigned int* done_ctr,
                           int64_t* tick_or_null);
int gm_head_bwd(void* stream, const float* H, int64_t ldh, const float* dS, const float* w2,
                const float* rowloss, float* dH, int64_t lddh, float* gw2, float* gb2,
                float* loss_out, gm_slot loss_slot, float inv_b, int gen_mode, int B, int Hd);

/* gm_head_bwd + optional Adam on (w2, b2) + optional per-graph tick (*tick += 1 once the loss
 * slot is written; later kernels of the same iteration then address slots with add - mul). */
int gm_head_bwd_fused(void* stream, const float* H, int64_t ldh, const float* dS, float* w2,
                      float* b2, const float* rowloss, float* dH, int64_t lddh, float* gw2,
                      float* gb2, float* loss_out, gm_slot loss_slot, float inv_b, int gen_mode,
                      int B, int Hd, int with_adam, float* mW, float* vW, float* mb, float* vb,
                      const float* sched, gm_slot sched_slot, double beta1, double beta2, double eps,
                      double weight_decay, float clamp, int64_t* tick);

/* ---- K16: BEGAN (be_gan.py:189-195, 212-258).  l1_rows: per-row L1 reconstruction error and its
 * gradient (coefficient 1/B for the first B rows, -K/B for the rest when K_dev is given);
 * began_dloss: DX, DG, D_loss = DX - K*DG from the row sums; began_update: convergence measure,
 * proportional control of K, the two ReduceLROnPlateau steps and the graph tick -- all on device
 * state (layout documented in csrc/gm_fused.hip), so the algorithm's per-step .item() syncs vanish. */
int gm_l1_rows(void* stream, const float* Y, int64_t ldy, const float* X, int64_t ldx, int R, int I,
               int B, const float* K_dev, float* dY, int64_t lddy, float* rowsum);
int gm_began_dloss(void* stream, const float* rows, int B, float* state, float* loss_out,
                   gm_slot loss_slot);
/* data parallel: B rows of this rank out of B_global (the means' denominator); DX / DG in the state
 * are then PARTIAL means: gm_allreduce_scalars(state + 1, 2) makes them global before began_update */
int gm_l1_rows_dp(void* stream, const float* Y, int64_t ldy, const float* X, int64_t ldx, int R, int I,
                  int B, int B_global, const float* K_dev, float* dY, int64_t lddy, float* rowsum);
int gm_began_dloss_dp(void* stream, const float* rows, int B, int B_global, float* state,
                      float* loss_out, gm_slot loss_slot);
int gm_began_update(void* stream, float* state, double* dstate, int64_t* istate, float gamma,
                    float lambda, int64_t patience, int64_t* tick);
/* gm_adam with a device-resident learning-rate scale (a power of two: exact). */
int gm_adam_scaled(void* stream, float* p, const float* g, float* m, float* v, int64_t n,
                   const float* sched, gm_slot sched_slot, double beta1, double beta2, double eps,
                   double weight_decay, float clamp, const float* lr_scale);
/* The critic head's backward and the first layer's weight gradient are independent once
 * gm_head_fwd_loss has written dH: this entry point runs gm_linear_bwd_dw_adam AND gm_head_bwd_fused
 * as ONE launch (the head workgroups ride in the GEMM's grid; 25 + 169 workgroups for the 784-400-1
 * critic at B = 256 -- one round of the 256 CUs).  `head` mirrors gm_head_bwd_fused's arguments.
 * sched == NULL (and head->with_adam == 0): plain gradients, no optimizer step (data-parallel runs
 * all-reduce the gradients first). */
typedef struct gm_head_bwd_args {
    const float* H; int64_t ldh;
    const float* dS; float* w2; float* b2; const float* rowloss;
    float* dH; int64_t lddh;              /* NULL when gm_head_fwd_loss wrote it */
    float* gw2; float* gb2;
    float* loss_out; gm_slot loss_slot;
    float inv_b; int gen_mode, B, Hd;
    int with_adam; float* mW; float* vW; float* mb; float* vb;
    const float* sched; gm_slot sched_slot;
    double beta1, beta2, eps, weight_decay; float clamp;
    int64_t* tick;
    const float* gw2_add;                 /* optional [Hd]: added to gw2 before it is stored / stepped */
    /* optional (pen_t != NULL): gw2[c] += sum_{r < pen_rows} [pen_s[r] > 0][pen_h[r][c] > 0] pen_t[r][c] -- the
     * gradient penalty's second-backward share of w2's gradient (w_gp_gan.py:215) summed by the
     * head's own workgroups instead of a launch of gm_gp_dw2_store */
    const float* pen_s; const float* pen_h; int64_t pen_ldh; const float* pen_t; int64_t pen_ldt; int pen_rows;
    const float* gb2_add;                 /* optional [1]: added to gb2 before it is stored / stepped (DRAGAN's sigma''
                                           * path reaches the head bias, dra_gan.py:207-223; gm_dragan_head_bwd_store) */
} gm_head_bwd_args;
int gm_linear_bwd_dw_adam_head(void* stream, const float* dA, int64_t lda, const float* X,
                               int64_t ldx, gm_slot x_slot, float* dW, float* db, int M, int K, int N,
                               float* pW, float* mW, float* vW, float* pb, float* mb, float* vb,
                               const float* sched, gm_slot sched_slot, double beta1, double beta2,
                               double eps, double weight_decay, float clamp,
                               const gm_head_bwd_args* head);
/* Same with a STACKED reduction: the first `ones_from` rows of dA / X contribute to dW but not to db
 * (WGAN-GP, w_gp_gan.py:207-218: dW1 = [u ; dH]^T [gamma ; X] in one GEMM -- the penalty's second
 * backward has no bias term). */
int gm_linear_bwd_dw_adam_head_ex(void* stream, const float* dA, int64_t lda, const float* X,
                                  int64_t ldx, gm_slot x_slot, float* dW, float* db, int M, int K, int N,
                                  float* pW, float* mW, float* vW, float* pb, float* mb, float* vb,
                                  const float* sched, gm_slot sched_slot, double beta1, double beta2,
                                  double eps, double weight_decay, float clamp,
                                  const gm_head_bwd_args* head, int ones_from);
/* gm_linear_bwd_dx carrying the head's backward workgroups (generator step: the single scalar
 * workgroup that writes the loss and ticks the iteration counter). */
int gm_linear_bwd_dx_head(void* stream, const float* dA, int64_t lda, const float* W, float* dX,
                          int64_t ldx, const float* below, int64_t ld_below, int M, int K, int N,
                          int epi, const gm_head_bwd_args* head);
/* ---- FOLDED critic head (round 3): no launch for the N = 1 layer at all.
 * Discriminator.forward's second layer (ns_gan.py:59: `discrimination = sigmoid(self.discriminate(
 * activated))`, a [R, 400] x [400] product) is split over the launches on either side of it:
 *  - gm_linear_fwd_headpart = gm_linear_fwd of the hidden layer whose epilogue also leaves, per
 *    32-column tile j of the hidden layer, part[r * ldp + j] = sum_{n in tile j} Y[r, n] * w2[n]
 *    (nparts = ceil(N / 32) <= 16 entries per row, rows ldp floats apart, ldp % 4 == 0, the unused
 *    entries of a row stay zero) and a snapshot snap[0..N) = w2, snap[N] = b2[0] of the head
 *    parameters it used (the consumers below may run in a launch that also steps w2 / b2);
 *  - the consumers rebuild, per row, score = act(sum_j part[r][j] + b2), the row's loss term and
 *    dS (the train_D / train_G loss lines: ns_gan.py:191-192,214 and the siblings gm_head_fwd_loss
 *    lists) in a prologue, and form dH[r, n] = dS_r * w2[n] * [Y[r, n] > 0] in registers while loading
 *    their A operand: gm_linear_bwd_dw_adam_head_fold (critic step: layer-1 weight gradient + Adam,
 *    head backward workgroups riding: gw2, gb2, loss, Adam on (w2, b2)) and gm_linear_bwd_dx_head_fold
 *    (generator step: dX through layer 1 + the loss / tick workgroup).  Both take the hidden
 *    activations H where the unfolded entry points take dH; head->dS / head->rowloss are not read.
 * Summation order of a score: 32-lane butterfly inside a tile, then tiles j = 0, 1, ... : fixed, so
 * results are bitwise reproducible run to run (they differ from gm_head_fwd_loss's order in the last
 * bits).  fold->S / dS / rowloss (optional, [R]) receive the per-row values for inspection. */
typedef struct gm_head_fold_args {
    const float* part; int64_t ldp; int nparts;
    const float* snap;
    int variant, out_act;
    float hyper[8]; int n_hyper;
    const float* pen;                     /* optional penalty rows added to the x rows' loss terms */
    float* S; float* dS; float* rowloss;  /* optional outputs */
} gm_head_fold_args;
int gm_linear_fwd_headpart(void* stream, const float* X, int64_t ldx, gm_slot x_slot, const float* W,
                           const float* bias, float* Y, int64_t ldy, int M, int K, int N, int act,
                           const float* w2, const float* b2, float* part, int64_t ldp, float* snap);
int gm_linear_bwd_dw_adam_head_fold(void* stream, const float* H, int64_t ldh, const float* X,
                                    int64_t ldx, gm_slot x_slot, float* dW, float* db, int M, int K, int N,
                                    float* pW, float* mW, float* vW, float* pb, float* mb, float* vb,
                                    const float* sched, gm_slot sched_slot, double beta1, double beta2,
                                    double eps, double weight_decay, float clamp,
                                    const gm_head_bwd_args* head, const gm_head_fold_args* fold);
int gm_linear_bwd_dx_head_fold(void* stream, const float* H, int64_t ldh, const float* W, float* dX,
                               int64_t ldx, const float* below, int64_t ld_below, int M, int K, int N,
                               int epi, const gm_head_bwd_args* head, const gm_head_fold_args* fold);
/* VAE / AE reconstruction loss where the reconstruction is produced (vae.py:196-203: `recon_loss =
 * torch.sum((images - outputs)**2)` behind Decoder.forward's sigmoid, vae.py:75-77; ae.py:147-160):
 * gm_linear_fwd of the decoder's last layer (sigmoid) whose epilogue also writes
 *   dA[m][n] = d loss / d (pre-sigmoid output) = (-2 (x - x_hat) (1 - x_hat)) x_hat   (as gm_sqerr_sigmoid_bwd)
 *   part[m * ldp + j] = sum_{n in 32-column tile j} (x[m][n] - x_hat[m][n])^2,  j < ceil(N / 32) <= ldp
 * (entries j >= ceil(N / 32) of a row are not written; gm_sum_finalize* over the whole `part` array adds
 * them up in a fixed order).  Replaces the separate gm_sqerr_sigmoid_bwd launch. */
int gm_linear_fwd_sqerr(void* stream, const float* X, int64_t ldx, const float* W, const float* bias,
                        float* Y, int64_t ldy, int M, int K, int N, const float* target,
                        int64_t ld_target, float* dA, int64_t lda, float* part, int64_t ldp);
/* VAE reparameterisation backward where dz is produced (vae.py:100-106 `z = mu + eps * exp(log_var/2)`
 * and kl_divergence :210-212, autograd of both): gm_linear_bwd_dx through the decoder's first layer
 * (dZ = dA W, W: [N, Z]) whose epilogue also writes, with the expressions of gm_vae_reparam_bwd,
 *   dml[m][c] = dz + mu,   dml[m][Z + c] = dz eps exp(lv/2) / 2 + (exp(lv) - 1) / 2.
 * Replaces the separate gm_vae_reparam_bwd launch (bit-identical results). */
int gm_linear_bwd_dx_reparam(void* stream, const float* dA, int64_t lda, const float* W, float* dZ,
                             int64_t ldz, int M, int Z, int N, const float* ml, int64_t ldml,
                             const float* eps, gm_slot eps_slot, float* dml, int64_t ldd);
/* gm_linear_fwd that also writes WGAN-GP's interpolate for its first `rows` output rows
 * (w_gp_gan.py:197-201: x_hat = eps * x + (1 - eps) * G(z), computed where G(z) is produced):
 *   x_hat[m][n] = eps[m] * x_real[m][n] + (1 - eps[m]) * Y[m][n],  m < rows
 * eps: per-row uniforms (ring base + eps_slot).  Saves the separate gm_interp launch. */
int gm_linear_fwd_interp(void* stream, const float* X, int64_t ldx, gm_slot x_slot, const float* W,
                         const float* bias, float* Y, int64_t ldy, int M, int K, int N, int act,
                         const float* eps, gm_slot eps_slot, const float* x_real, int64_t ld_real,
                         float* x_hat, int64_t ld_hat, int rows);
/* gm_linear_fwd and gm_gather_rows as ONE launch: the gather workgroups ride in the GEMM's grid.  The
 * gather only reads the index ring and the resident dataset, so any forward launch that does not
 * touch `out` can carry it (the engine uses the generator's first layer, ns_gan.py:44 + :222-226). */
int gm_linear_fwd_gather(void* stream, const float* X, int64_t ldx, gm_slot x_slot, const float* W,
                         const float* bias, float* Y, int64_t ldy, int M, int K, int N, int act,
                         const float* data, int64_t n_rows, const int64_t* idx, gm_slot idx_slot,
                         float* out, int64_t ld_out, int B, int row_elems);
/* Same with the BIT-PACKED resident dataset (utils.py:31 binarises MNIST): row r
 * = bits[r*words_per_row ...], pixel i = bit (i & 31) of word i >> 5; the gather expands to fp32 rows. */
int gm_linear_fwd_gather_bits(void* stream, const float* X, int64_t ldx, gm_slot x_slot, const float* W,
                              const float* bias, float* Y, int64_t ldy, int M, int K, int N, int act,
                              const uint32_t* bits, int words_per_row, int64_t n_rows, const int64_t* idx,
                              gm_slot idx_slot, float* out, int64_t ld_out, int B, int row_elems);
int gm_gather_rows_bits(void* stream, const uint32_t* bits, int words_per_row, int64_t n_rows,
                        const int64_t* idx, gm_slot idx_slot, float* out, int64_t ld_out, int B,
                        int row_elems);
/* Bit-packed rows AS A GEMM OPERAND (the data is process_batch's, ns_gan.py:222-226, binarised by
 * utils.py:31): the gather copies the selected rows as WORDS (out_bits[b * words_per_row ..), 100 B per MNIST row
 * instead of 3136), and the folded critic step's two launches read rows [0, rows) of X from that copy, expanding to
 * 0.0f / 1.0f in registers -- the fp32 rows X[0 .. rows) are neither written nor read.  Same MFMA sequence on the same
 * values: results are bit-identical to the fp32-operand entry points.  rows % 32 == 0, K % 4 == 0 (a forward tile /
 * reduction chunk is packed as a whole); one tile shape per launch (32x32 forward, 32x48 weight gradient, operands
 * through registers); anything else returns GM_EINVAL -- there is no fp32 copy to fall back to. */
int gm_gather_rows_bits_packed(void* stream, const uint32_t* bits, int words_per_row, int64_t n_rows,
                               const int64_t* idx, gm_slot idx_slot, uint32_t* out_bits, int B);
int gm_linear_fwd_gather_bits_packed(void* stream, const float* X, int64_t ldx, gm_slot x_slot, const float* W,
                                     const float* bias, float* Y, int64_t ldy, int M, int K, int N, int act,
                                     const uint32_t* bits, int words_per_row, int64_t n_rows, const int64_t* idx,
                                     gm_slot idx_slot, uint32_t* out_bits, int B);
int gm_linear_fwd_headpart_bits(void* stream, const float* X, int64_t ldx, gm_slot x_slot, const float* W,
                                const float* bias, float* Y, int64_t ldy, int M, int K, int N, int act,
                                const float* w2, const float* b2, float* part, int64_t ldp, float* snap,
                                const uint32_t* xbits, int words_per_row, int rows);
int gm_linear_bwd_dw_adam_head_fold_bits(void* stream, const float* H, int64_t ldh, const float* X,
                                         int64_t ldx, gm_slot x_slot, float* dW, float* db, int M, int K, int N,
                                         float* pW, float* mW, float* vW, float* pb, float* mb, float* vb,
                                         const float* sched, gm_slot sched_slot, double beta1, double beta2,
                                         double eps, double weight_decay, float clamp,
                                         const gm_head_bwd_args* head, const gm_head_fold_args* fold,
                                         const uint32_t* xbits, int words_per_row, int rows);
/* Two gm_linear_bwd_dw_adam calls over the same batch rows as ONE launch (the generator step's two
 * weight gradients are independent once d loss / d hidden is known).  Falls back to two launches
 * when the pair cannot share a tile configuration.  sched == NULL in an argument block: plain
 * gradient for that GEMM (no optimizer step). */
typedef struct gm_dw_adam_args {
    const float* dA; int64_t lda; const float* X; int64_t ldx; gm_slot x_slot;
    float* dW; float* db; int M, K, N;
    float* pW; float* mW; float* vW; float* pb; float* mb; float* vb;
    const float* sched; gm_slot sched_slot;
    double beta1, beta2, eps, weight_decay; float clamp;
} gm_dw_adam_args;
int gm_linear_bwd_dw_adam_pair(void* stream, const gm_dw_adam_args* first,
                               const gm_dw_adam_args* second);
/* The pair as the LAST launch of a VAE batch (vae.py:162 + the loss sums of :203 / :212): one more workgroup adds up
 * the two partial arrays exactly as gm_sum_finalize2_tick does, and the last workgroup of the launch to finish
 * advances `tick` (every slot of the batch has been resolved by then).  done: one zero-initialised unsigned int the
 * launch counts its workgroups on and re-arms.  Falls back to separate launches when the pair cannot share a tile. */
typedef struct gm_finalize2_args {
    const float* pa; int na; float scale_a; float* out_a; gm_slot slot_a;
    const float* pb; int nb; float scale_b; float* out_b; gm_slot slot_b;
    int64_t* tick; unsigned int* done;
} gm_finalize2_args;
int gm_linear_bwd_dw_adam_pair_finalize(void* stream, const gm_dw_adam_args* first, const gm_dw_adam_args* second,
                                        const gm_finalize2_args* fin);
/* gm_linear_bwd_dx with an additive term before the activation gradient:
 * dX = (dA*W + add_scale*add) * act'(below)   (BEGAN's generator sees G(z) both through D and
 * directly in |D(G(z)) - G(z)|, be_gan.py:256). */
int gm_linear_bwd_dx_add(void* stream, const float* dA, int64_t lda, const float* W, float* dX,
                         int64_t ldx, const float* below, int64_t ld_below, int M, int K, int N,
                         int epi, const float* add, int64_t ldadd, float add_scale);

/* ---- K13: DRAGAN penalty (dra_gan.py:198-223; derivation in SURVEY.md A.3).  std_all: unbiased std
 * of the whole real batch; xhat: delta*x + (1-delta)*(x + C*std*U); rows: per-row norm of the input
 * gradient s'*v (v = (m1.w2) W1 from gm_gp_u + gm_linear_bwd_dx), penalty rows, dv and da2 of the
 * second backward; head_bwd: the w2/b2 accumulations and da1.  Penalty rows are added to the loss by
 * gm_head_fwd_loss / gm_gan_loss with weight hyper[7]. */
/* ws: GM_STD_WS_BYTES of device memory (8-byte aligned), zeroed ONCE by the caller and then owned by these
 * calls (partial sums of the launch's 64 workgroups + their arrival counter; launches sharing a ws must be
 * stream-ordered). */
#define GM_STD_WS_BYTES 1088
int gm_std_all(void* stream, const float* X, int64_t ldx, int R, int I, float* out, void* ws);
/* data parallel: (sum x, sum x^2) of this rank's rows -> scalar all-reduce -> std of the global batch */
int gm_std_sums(void* stream, const float* X, int64_t ldx, int R, int I, float* out2, void* ws);
int gm_std_from_sums(void* stream, const float* sums2, int64_t n_total, float* out);
int gm_dragan_xhat(void* stream, const float* x, int64_t ldx, const float* delta, gm_slot delta_slot,
                   const float* U, gm_slot u_slot, const float* std_dev, float C, float* out,
                   int64_t ldo, int B, int I);
int gm_dragan_rows(void* stream, const float* s, const float* V, int64_t ldv, float* dv, int64_t lddv,
                   float* da2, float* pen, float lambda, float inv_b, float K_norm, int B, int I);
int gm_dragan_head_bwd(void* stream, const float* H, int64_t ldh, const float* T, int64_t ldt,
                       const float* da2, const float* w2, float* gw2, float* gb2, float* dA1,
                       int64_t ldd, int B, int Hd);
/* Fisher GAN (fisher_gan.py:155-156) on the folded critic head: gm_linear_bwd_dw_adam_head_fold with variant
 * GM_LOSS_FISHER reads lambda = aux[0] in every workgroup and leaves lambda + rho * d loss / d lambda in aux[5]
 * (and the four moments in aux[1..4]); this launch makes it lambda.  aux: 8 floats. */
int gm_fisher_commit(void* stream, float* aux);
/* The same with gw2 / gb2 STORED instead of accumulated: the penalty's share of the head's gradient on its own, for
 * gm_head_bwd_args.gw2_add / gb2_add of the stacked critic step (the head's backward adds them before Adam). */
int gm_dragan_head_bwd_store(void* stream, const float* H, int64_t ldh, const float* T, int64_t ldt,
                             const float* da2, const float* w2, float* gw2, float* gb2, float* dA1,
                             int64_t ldd, int B, int Hd);

/* ---- K15: InfoGAN mutual-information loss (train_Q, info_gan.py:269-304): cross-entropy of the
 * categorical code + mean-squared error of the continuous code, and d loss / d q.  noise rows are
 * [z | one-hot c1 | c2] as built by compute_noise (info_gan.py:306-325). */
int gm_info_q_loss(void* stream, const float* q, int64_t ldq, const float* noise, gm_slot noise_slot,
                   int64_t ldn, int B, int z_dim, int disc_dim, int cont_dim, float lambda, float* dq,
                   int64_t lddq, float* loss_out, gm_slot loss_slot);
/* data parallel: this rank's B rows of a global batch of B_global (both means' denominator) */
int gm_info_q_loss_dp(void* stream, const float* q, int64_t ldq, const float* noise, gm_slot noise_slot,
                      int64_t ldn, int B, int B_global, int z_dim, int disc_dim, int cont_dim, float lambda,
                      float* dq, int64_t lddq, float* loss_out, gm_slot loss_slot);

/* ---- elementwise activation backward for the general autograd path:
 * dA = dY * act'(Y)  (Relu/SigmoidBackward, ns_gan.py:44-45). */
int gm_act_bwd(void* stream, const float* dY, const float* Y, float* dA, int64_t n, int act);

/* ---- stage-in of the per-iteration inputs the reference moves host->device every step
 * (`to_cuda` of the image batch indices and of the CPU-generated noise: ns_gan.py:220,225,
 * utils.py:10-14).  The host writes them into PINNED rings that mirror the device rings; this kernel
 * (first node of every captured graph) copies the slots [i, i+n_iters) of every segment, i resolved
 * from `slot` (stride ignored), host ring -> device ring.  src: device-visible address of pinned
 * host memory (gm_host_device_ptr). */
#define GM_STAGE_MAX_SEGS 8
typedef struct gm_stage_seg {
    const void* src;
    void* dst;
    int64_t bytes_per_iter;     /* bytes copied per iteration (all of its blocks) */
    /* An iteration's slot may consist of `blocks` equal pieces that are not adjacent in the source or
     * the destination: a data-parallel rank stages only ITS rows of every [B, w] draw of the global
     * batch (host ring: global batch; device ring: the rank's rows).  Piece q of iteration i is
     * bytes_per_iter / blocks bytes at src + (i * blocks + q) * src_block_stride, written to
     * dst + (i * blocks + q) * dst_block_stride.  blocks <= 1 and strides 0: one dense piece per
     * iteration (stride = bytes_per_iter), the single-rank layout. */
    int32_t blocks;
    int32_t reserved;
    int64_t src_block_stride, dst_block_stride;
} gm_stage_seg;
int gm_stage_in(void* stream, const gm_stage_seg* segs, int n_segs, gm_slot slot, int n_iters);
/* The same with a FILL GATE: the launch may be enqueued before the host has finished writing the
 * iterations' slots (ns_gan.py:218-226 draws them on the host one step at a time; here the host
 * writes a sub-chunk and then advances gate[0] = number of iterations written since configure).
 * ONE wave -- its own one-wave launch in front of the copy, same stream -- waits (system-scope loads of pinned
 * host memory) until gate[0] >= it + n_iters, it = index of `it_slot` (the absolute iteration); after
 * timeout_s seconds it raises gate[1] = 1 and the copy proceeds -- the host must check gate[1] before
 * trusting results.  (Polled from every workgroup of the copy the gate cost a serialized PCIe read per
 * workgroup, ~1 us per iteration staged: round 5.)
 * gate: device-visible address (gm_host_device_ptr) of two int64 in pinned host memory.
 * publish (optional, device memory): workgroup (0,0) stores `it` there -- a second stage-in that runs
 * on a forked branch of the graph, concurrently with iterations that advance the step counter, resolves
 * its slots from that word instead.  max_blocks (1..256): workgroups per segment (a concurrent
 * stage-in should leave the CUs to the iteration kernels). */
int gm_stage_in_gated(void* stream, const gm_stage_seg* segs, int n_segs, gm_slot slot, int n_iters,
                      const int64_t* gate, gm_slot it_slot, double timeout_s, int64_t* publish,
                      int max_blocks);
/* Device-side address of a pinned host allocation (hipHostGetDevicePointer). */
int gm_host_device_ptr(void* host_ptr, void** dev_ptr_out);

/* ---- C1 (NEW: the reference has no distributed code): gradient exchange of
 * the data-parallel step between the ranks of one node, as kernels that live inside the iteration's
 * hipGraph (csrc/gm_comm.hip).  Every rank creates a communicator (an exchange region in fine-grained
 * device memory), the 64-byte IPC handles are exchanged by the host (torch.distributed
 * all_gather_object), gm_comm_connect maps the peers.  gm_allreduce_f32: in-place SUM over ranks of
 * buf[0..n) (n % 4 == 0, 16-byte aligned), identical bits on every rank; gm_allreduce_adam_f32: same,
 * with optim.Adam.step (ns_gan.py:139,156) applied to (p, m, v) by the kernel that writes the reduced
 * gradient; gm_allreduce_scalars: up to 16 floats (the pre-reductions of the losses that are not a
 * mean of per-sample terms: ra_gan.py:204, fisher_gan.py:214-223, dra_gan.py:204, be_gan.py:189-195).
 * Waits are bounded; gm_comm_error reports an expired one. */
int gm_comm_create(int rank, int world, int64_t n_floats, void** comm_out, void* handle_out64);
int gm_comm_connect(void* comm, const void* all_handles /* world x 64 bytes, rank order */);
int gm_comm_destroy(void* comm);
int gm_comm_error(void* comm, int* flag_out);
/* *fine_grained_out = 1 when the exchange region is fine-grained device memory (peers on OTHER GPUs
 * may store into it while a kernel polls it), 0 when the runtime refused that allocation and the
 * region is plain device memory: valid only when every rank shares one device. */
int gm_comm_info(void* comm, int* fine_grained_out);
/* Exchange form.  0 (default): an all-reduce is ONE kernel (reduce-scatter, arrival counter, all-gather + Adam) that
 * READS the peers' buckets over the peer mappings; 1: the same as two launches (reduce, gather) -- ranks that share
 * one device must use 1 unless they also lower gm_comm_set_max_blocks: the one-kernel forms keep every rank's
 * workgroups spinning on peer flags, which starves the peers' GEMM workgroups when they run on the same CUs
 * (dp.PeerComm sets it from the ranks' device identities); 2: ONE kernel that moves the data by posted remote
 * WRITES only (every rank deposits its contributions in the slice owners' staging areas, owners write the reduced
 * slices into every rank's `out`): no xGMI read round trips.  All three give bit-identical sums (rank order). */
int gm_comm_set_exchange(void* comm, int form);
/* Upper bound on the workgroups of one exchange launch.  The one-kernel form needs ALL its workgroups co-resident
 * (each spins until every peer's last workgroup has arrived): the default (max_blocks = 0) is what the occupancy API
 * reports for this device or partition (CPX mode, HSA_CU_MASK), at most 320; ranks that share one device pass less. */
int gm_comm_set_max_blocks(void* comm, int max_blocks);
/* Bound (seconds of the 100 MHz wall clock, default 10) of every device-side wait of this communicator's later
 * launches; an expired wait raises gm_comm_error's flag instead of hanging the GPU.  The start-up self-check between
 * ranks on different devices runs with 2 s. */
int gm_comm_set_wait_seconds(void* comm, double seconds);
/* The region's own bucket (n_floats fp32, 256-byte aligned): a gradient buffer placed here is
 * all-reduced without a staging copy. */
int gm_comm_buffer(void* comm, void** ptr_out, int64_t* n_floats_out);
int gm_allreduce_f32(void* comm, void* stream, float* buf, int64_t n);
int gm_allreduce_adam_f32(void* comm, void* stream, float* grad, int64_t n, float* p, float* m,
                          float* v, const float* sched, gm_slot sched_slot, double beta1,
                          double beta2, double eps, double weight_decay, float clamp,
                          const float* lr_scale_or_null);
int gm_allreduce_scalars(void* comm, void* stream, float* vals, int k);
/* The same SUM as an RCCL collective that can be CAPTURED into the iteration's hipGraph (the fallback when peer
 * mappings are not available: GM_DP_COMM=rccl; round 1 launched torch.distributed all-reduces from the host between
 * segment graphs).  RCCL is not linked: its symbols are resolved at first use from the librccl the process already
 * carries (torch's).  uid128: 128 bytes (ncclUniqueId) made on rank 0 by gm_rccl_unique_id and handed to every rank
 * by the host; gm_rccl_comm_create is collective.  In place, fp32. */
int gm_rccl_available(void);          /* 1: ncclGetUniqueId / CommInitRank / AllReduce / CommDestroy all resolve here */
int gm_rccl_unique_id(void* uid128_out);
int gm_rccl_comm_create(int rank, int world, const void* uid128, void** rccl_comm_out);
int gm_rccl_allreduce_f32(void* rccl_comm, void* stream, float* buf, int64_t n);
int gm_rccl_comm_destroy(void* rccl_comm);

/* ---- HOST helper (no device work): first B entries of torch.randperm(n, generator=
 * Generator().manual_seed(seed)) in O(B): RandomSampler.__iter__ (torch/utils/data/sampler.py:160-185)
 * feeding process_batch (ns_gan.py:222-226).  mt19937 seeded with the low 32 bits of `seed`,
 * forward Fisher-Yates `z = random() % (n-i); swap(r[i], r[i+z])` as in ATen randperm_cpu.
 * Bit-exact sampling indices without materialising the 50 000-entry permutation. */
int gm_randperm_prefix(uint64_t seed, int64_t n, int B, int64_t* out_host);
/* HOST: advance a serialized torch CPU generator state (torch.get_rng_state(), 5056 bytes,
 * mt19937) by n 32-bit outputs without producing them.  Data-parallel ranks replay the reference's
 * global draw protocol (ns_gan.py:183,208,220) but materialise only their own rows of each noise
 * tensor; the draws belonging to other ranks' rows are skipped with this call. */
int gm_mt19937_skip(void* torch_cpu_rng_state, int64_t state_bytes, uint64_t n);

/* ---- HOST: replay of the reference's global-CPU-generator protocol for a run of iterations in
 * ONE call (csrc/gm_hostrng.cpp).  A program is the ordered list of draws of one iteration:
 *   GM_DRAW_SAMPLER  process_batch (ns_gan.py:222-226): DataLoader base seed + RandomSampler seed
 *                    (two int64 random_() draws), first n entries of randperm(a) -> int64 dst[n]
 *   GM_DRAW_NORMAL   torch.randn(n) on contiguous fp32, n >= 16 (ns_gan.py:220, vae.py:104)
 *   GM_DRAW_UNIFORM  torch.rand(n) fp32 (w_gp_gan.py:197, dra_gan.py:200,205)
 *   GM_DRAW_INFO     info_gan.py:312-323: [randn(n,a) | one_hot(randint(0,b,(n,))) | randn(n,c)]
 * executed n_iters times; iteration i writes to dst + i*iter_stride (host memory, e.g. pinned
 * staging).  [e0,e1) limits what is materialised (a data-parallel rank's rows) while the stream
 * advances as for the whole tensor.  Returns -10002 for shapes outside the restated ATen paths
 * (the caller then draws through torch).  The serialized state (torch.get_rng_state()) is
 * advanced in place. */
typedef struct gm_draw_op {
    int32_t kind;
    int32_t n;
    int64_t a;
    int32_t b, c;
    void* dst;
    int64_t iter_stride;
    int64_t e0, e1;
} gm_draw_op;
enum { GM_DRAW_SAMPLER = 0, GM_DRAW_NORMAL = 1, GM_DRAW_UNIFORM = 2, GM_DRAW_INFO = 3 };
int gm_host_replay(void* torch_cpu_rng_state, int64_t state_bytes, const gm_draw_op* ops, int n_ops,
                   int n_iters);
/* HOST: which restatement of ATen's float normal_fill the replay uses: 0 = scalar libm
 * (normal_fill_16<float>), 1 = avx_mathfun.h polynomials with the mul+add pairs contracted to FMAs,
 * 2 = same without contraction.  Python picks the one that reproduces torch bit for bit. */
int gm_host_replay_flavour(int flavour);
/* HOST: the same replay as a job for the library's fill worker (one persistent native thread, jobs run
 * in submission order on the caller-owned generator state buffer, which must stay valid until the job
 * is retired).  After the job's writes the worker stores *gate = gate_value (release) -- the fill gate
 * of gm_stage_in_gated -- when gate is non-null.  Returns the job id (> 0) or a negative error.
 * gm_fill_wait(id): block until job id is retired, returns the worker's sticky error (0 = none; after
 * an error later jobs are retired without running and without opening their gates).
 * gm_fill_completed(): id of the last retired job.  gm_fill_reset(): wait for everything submitted and
 * clear the sticky error.  Replaces one Python thread hop + two generator-state copies per sub-chunk
 * of ns_gan.py:183,208,222-226-style draws. */
int64_t gm_fill_submit(void* torch_cpu_rng_state, int64_t state_bytes, const gm_draw_op* ops, int n_ops,
                       int n_iters, int64_t* gate, int64_t gate_value);
int gm_fill_wait(int64_t id);
int64_t gm_fill_completed(void);
int gm_fill_reset(void);
/* HOST: size of the worker pool of gm_host_replay's Box-Muller stage (1 = caller only). */
int gm_host_replay_threads(int n_threads);
/* numpy's LEGACY global generator (np.random.normal of bir_vae.py:92-94): n values loc + scale * legacy_gauss as
 * float32, bit-identical to torch.from_numpy(np.random.normal(loc, scale, n)).float(); key[624] / pos / has_gauss /
 * gauss are np.random.get_state(legacy=True)'s fields, advanced in place.  The log / sqrt stage of the polar
 * method runs on n_threads. */
int gm_numpy_legacy_normal_f32(uint32_t* key, int32_t* pos, int32_t* has_gauss, double* gauss, double loc,
                               double scale, int64_t n, float* out, int n_threads);

/* ---- graph capture helpers (HIP graphs instead of a tracing compiler) ------------------ */
int gm_graph_begin(void* stream);
int gm_graph_end(void* stream, void** graph_exec_out);
int gm_graph_launch(void* graph_exec, void* stream);
int gm_graph_destroy(void* graph_exec);

/* ---- side streams + cross-stream dependencies, so independent kernels of one iteration become
 * parallel branches of the captured hipGraph */
int gm_stream_create(void** stream_out);
int gm_stream_destroy(void* stream);
int gm_stream_wait_event(void* stream, void* ev);

/* ---- diagnostics: dependent v_mfma_f32_32x32x2_f32 chain on 256 workgroups; out2[0] = shader
 * cycles (s_memtime), out2[1] = 100 MHz wall-clock ticks spanned by block 0. */
int gm_clock_probe(void* stream, int iters, unsigned long long* out2, float* sink);

/* ---- timing helpers for bench.py (HIP events on the launch stream) --------------------- */
int gm_event_create(void** ev_out);
int gm_event_record(void* ev, void* stream);
int gm_event_sync(void* ev);
int gm_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out);
int gm_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif /* GM_HIP_H */
