"""Tensor-level wrappers for the variant-specific fused kernels (csrc/gm_fused.hip)."""
import torch

from . import _lib
from ._lib import NO_SLOT
from .ops import _ld, stream_ptr


def interp(eps, eps_slot, x, g, out, stream=None):
    """x_hat = eps*x + (1-eps)*g   (w_gp_gan.py:197-201)."""
    B, I = out.shape
    _lib.call("gm_interp", stream or stream_ptr(), eps.data_ptr(), eps_slot, x.data_ptr(), _ld(x),
              g.data_ptr(), _ld(g), out.data_ptr(), _ld(out), B, I)


def gp_u(s, h, w2, u, stream=None):
    B, H = u.shape
    _lib.call("gm_gp_u", stream or stream_ptr(), s.data_ptr(), h.data_ptr(), _ld(h), w2.data_ptr(),
              u.data_ptr(), _ld(u), B, H)


def gp_norm(g, gamma, pen, lam, inv_b, k=1.0, stream=None):
    B, I = g.shape
    _lib.call("gm_gp_norm", stream or stream_ptr(), g.data_ptr(), _ld(g), gamma.data_ptr(),
              _ld(gamma), pen.data_ptr(), lam, inv_b, k, B, I)


def gp_dw2(s, h, t, gw2, stream=None):
    B, H = h.shape
    _lib.call("gm_gp_dw2", stream or stream_ptr(), s.data_ptr(), h.data_ptr(), _ld(h), t.data_ptr(),
              _ld(t), gw2.data_ptr(), B, H)


def gp_dw2_store(s, h, t, out, stream=None):
    """out[n] = sum_b [s_b>0][h[b,n]>0] t[b,n] (gp_dw2 without the accumulation)."""
    B, H = h.shape
    _lib.call("gm_gp_dw2_store", stream or stream_ptr(), s.data_ptr(), h.data_ptr(), _ld(h), t.data_ptr(),
              _ld(t), out.data_ptr(), B, H)


def head_gp(h, w2, b2, s, u, stream=None):
    """D(x_hat)'s N = 1 layer + the seed of the input gradient in one launch (w_gp_gan.py:202-212)."""
    B, H = u.shape
    _lib.call("gm_head_gp", stream or stream_ptr(), h.data_ptr(), _ld(h), w2.data_ptr(), b2.data_ptr(),
              s.data_ptr(), u.data_ptr(), _ld(u), B, H)


def bir_reparam(mu, eps, z, B, Z, eps_slot=NO_SLOT, stream=None):
    """z = mu + eps (bir_vae.py:86-97; eps drawn on the host from numpy's global RNG)."""
    _lib.call("gm_bir_reparam", stream or stream_ptr(), mu.data_ptr(), _ld(mu), eps.data_ptr(), eps_slot,
              z.data_ptr(), _ld(z), B, Z)


def bir_mmd(z, prior, partial, dz, B, Z, lam, prior_slot=NO_SLOT, stream=None):
    """Row shares of the Gaussian-kernel MMD (bir_vae.py:201-221) and d(lam*mmd)/dz (dz may be None)."""
    _lib.call("gm_bir_mmd", stream or stream_ptr(), z.data_ptr(), _ld(z), prior.data_ptr(), prior_slot,
              partial.data_ptr(), dz.data_ptr() if dz is not None else None,
              _ld(dz) if dz is not None else 0, B, Z, lam)


def vae_reparam(ml, eps, z, kl_out, B, Z, eps_slot=NO_SLOT, kl_slot=NO_SLOT, stream=None):
    _lib.call("gm_vae_reparam", stream or stream_ptr(), ml.data_ptr(), _ld(ml), eps.data_ptr(),
              eps_slot, z.data_ptr(), _ld(z), kl_out.data_ptr(), kl_slot, B, Z)


def vae_reparam_wide(ml, eps, z, kl_part, B, Z, eps_slot=NO_SLOT, stream=None):
    """z = mu + eps*exp(lv/2); KL left as per-workgroup partial sums in kl_part (see sum_finalize2)."""
    _lib.call("gm_vae_reparam_wide", stream or stream_ptr(), ml.data_ptr(), _ld(ml), eps.data_ptr(),
              eps_slot, z.data_ptr(), _ld(z), kl_part.data_ptr(), kl_part.numel(), B, Z)
    return (B * Z + 255) // 256


def vae_bwd_mid(dHdec, Wd1, ml, eps, dml, Wml, He, dHe, B, eps_slot=NO_SLOT, stream=None):
    """dz = dHdec W_d1, d loss / d [mu | log_var] (-> dml), dHe = (dml W_ml) . [He > 0] as ONE launch (gm_vae_bwd_mid)."""
    Hd, Z = Wd1.shape
    if Hd > 512 or Wml.shape[1] != Hd or He.shape[1] < Hd:
        raise ValueError("vae_bwd_mid: one hidden width <= 512 for decoder and encoder (got %d / %d); use "
                         "linear_bwd_dx_reparam + linear_bwd_dx" % (Hd, Wml.shape[1]))
    _lib.call("gm_vae_bwd_mid", stream or stream_ptr(), dHdec.data_ptr(), _ld(dHdec), Wd1.data_ptr(), ml.data_ptr(),
              _ld(ml), eps.data_ptr(), eps_slot, dml.data_ptr(), _ld(dml), Wml.data_ptr(), He.data_ptr(), _ld(He),
              dHe.data_ptr(), _ld(dHe), B, Hd, Z)


def vae_reparam_fwd(ml, eps, z, kl_part, B, Z, W, bias, H, act, eps_slot=NO_SLOT, stream=None):
    """vae_reparam_wide + the decoder's first layer H = act(z W^T + bias) as ONE launch (gm_vae_reparam_fwd)."""
    from .ops import ACT
    _lib.call("gm_vae_reparam_fwd", stream or stream_ptr(), ml.data_ptr(), _ld(ml), eps.data_ptr(), eps_slot,
              z.data_ptr(), _ld(z), kl_part.data_ptr(), kl_part.numel(), B, Z, W.data_ptr(),
              bias.data_ptr() if bias is not None else None, H.data_ptr(), _ld(H), W.shape[0], ACT[act])
    return (B * Z + 255) // 256


def sum_finalize2(pa, na, out_a, slot_a, pb, nb, out_b, slot_b, scale_a=1.0, scale_b=1.0, tick=None, stream=None):
    """Two fixed-order fp64 sums in one launch; tick: device step counter to advance (last launch of a step)."""
    _lib.call("gm_sum_finalize2_tick", stream or stream_ptr(), pa.data_ptr(), na, scale_a, out_a.data_ptr(),
              slot_a, pb.data_ptr(), nb, scale_b, out_b.data_ptr(), slot_b,
              tick.data_ptr() if tick is not None else None)


def vae_reparam_bwd(ml, eps, dz, dml, B, Z, eps_slot=NO_SLOT, stream=None):
    _lib.call("gm_vae_reparam_bwd", stream or stream_ptr(), ml.data_ptr(), _ld(ml), eps.data_ptr(),
              eps_slot, dz.data_ptr(), _ld(dz), dml.data_ptr(), _ld(dml), B, Z)


def sqerr_sigmoid_bwd(x, xr, dA, partial, B, stream=None):
    I = x.shape[1]
    _lib.call("gm_sqerr_sigmoid_bwd", stream or stream_ptr(), x.data_ptr(), _ld(x), xr.data_ptr(),
              _ld(xr), dA.data_ptr(), _ld(dA), partial.data_ptr(), B, I)


def sum_finalize(partial, n, out, scale=1.0, out_slot=NO_SLOT, tick=None, stream=None):
    """out[slot] = scale * sum(partial[:n]) (fp64, fixed order).  tick: device step counter to advance
    (this is then the last launch of the step)."""
    if tick is not None:
        _lib.call("gm_sum_finalize_tick", stream or stream_ptr(), partial.data_ptr(), n, scale,
                  out.data_ptr(), out_slot, tick.data_ptr())
        return
    _lib.call("gm_sum_finalize", stream or stream_ptr(), partial.data_ptr(), n, scale,
              out.data_ptr(), out_slot)


def head_fwd_loss(variant, gen_mode, H, w2, b2, out_act, B, hyper, inv_b, pen, S, dS, rowloss,
                  dH=None, final=None, stream=None):
    """Fused critic head forward + per-row loss + d loss/d pre-activation (separable variants).
    dH: also write the hidden-layer gradient.  final = dict(loss_out, loss_slot, done, tick=None):
    the last workgroup also writes the loss scalar (and ticks) -- no head_bwd needed for scalars."""
    import ctypes
    from ._lib import ACT, LOSS
    h = (ctypes.c_float * 8)(*([float(x) for x in hyper] + [0.0] * (8 - len(hyper))))
    args = [stream or stream_ptr(), LOSS[variant], 1 if gen_mode else 0,
            H.data_ptr(), _ld(H), w2.data_ptr(), b2.data_ptr(), ACT[out_act], B, H.shape[1], h,
            len(hyper), inv_b, pen.data_ptr() if pen is not None else None, S.data_ptr(),
            dS.data_ptr(), rowloss.data_ptr(), dH.data_ptr() if dH is not None else None,
            _ld(dH) if dH is not None else 0]
    if final is None:
        _lib.call("gm_head_fwd_loss", *args)
        return
    done, tick = final["done"], final.get("tick")
    assert done.dtype == torch.int32 and done.numel() >= 1
    _lib.call("gm_head_fwd_loss_final", *args, final["loss_out"].data_ptr(), final["loss_slot"],
              done.data_ptr(), tick.data_ptr() if tick is not None else None)


def head_bwd(H, dS, w2, rowloss, dH, gw2, gb2, loss_out, loss_slot, inv_b, gen_mode, B, lin=None,
             adam=None, tick=None, betas=(0.9, 0.999), eps=1e-8, stream=None):
    """dH = dS (x) w2 masked by H>0; gw2 = dS^T H; gb2; loss scalar (see gm_hip.h).
    adam (dict(sched, sched_slot, clamp)) + lin (engine._Linear of the head): also apply Adam to
    (w2, b2) here.  tick: device int64 counter to advance once the loss slot is written."""
    g = lambda t: t.data_ptr() if t is not None else None
    ldd = _ld(dH) if dH is not None else 0      # dH=None: head_fwd_loss already wrote it
    if adam is None and tick is None:
        _lib.call("gm_head_bwd", stream or stream_ptr(), H.data_ptr(), _ld(H), dS.data_ptr(),
                  w2.data_ptr(), rowloss.data_ptr(), g(dH), ldd, g(gw2), g(gb2),
                  loss_out.data_ptr(), loss_slot, inv_b, 1 if gen_mode else 0, B, H.shape[1])
        return
    with_adam = adam is not None
    _lib.call("gm_head_bwd_fused", stream or stream_ptr(), H.data_ptr(), _ld(H), dS.data_ptr(),
              w2.data_ptr(), lin.b.data_ptr() if with_adam else None, rowloss.data_ptr(),
              g(dH), ldd, g(gw2), g(gb2), loss_out.data_ptr(), loss_slot, inv_b,
              1 if gen_mode else 0, B, H.shape[1], 1 if with_adam else 0,
              lin.mW.data_ptr() if with_adam else None, lin.vW.data_ptr() if with_adam else None,
              lin.mb.data_ptr() if with_adam else None, lin.vb.data_ptr() if with_adam else None,
              adam["sched"].data_ptr() if with_adam else None,
              adam["sched_slot"] if with_adam else NO_SLOT, betas[0], betas[1], eps, 0.0,
              adam.get("clamp", 0.0) if with_adam else 0.0, g(tick))


def info_q_loss(q, noise, noise_slot, B, z_dim, disc_dim, cont_dim, dq, loss_out, loss_slot,
                lam=1.0, B_global=None, stream=None):
    """InfoGAN train_Q loss (info_gan.py:295-302) + d loss / d q."""
    _lib.call("gm_info_q_loss_dp", stream or stream_ptr(), q.data_ptr(), _ld(q), noise.data_ptr(),
              noise_slot, z_dim + disc_dim + cont_dim, B, B if B_global is None else B_global, z_dim,
              disc_dim, cont_dim, lam, dq.data_ptr(), _ld(dq), loss_out.data_ptr(), loss_slot)


def l1_rows(Y, X, R, B, K_dev, dY, rowsum, B_global=None, stream=None):
    """Per-row L1 error + gradient (BEGAN, be_gan.py:225-236,256).  B_global: the mean's
    denominator when this rank holds B of B_global rows."""
    _lib.call("gm_l1_rows_dp", stream or stream_ptr(), Y.data_ptr(), _ld(Y), X.data_ptr(), _ld(X), R,
              Y.shape[1], B, B if B_global is None else B_global,
              K_dev.data_ptr() if K_dev is not None else None, dY.data_ptr(), _ld(dY), rowsum.data_ptr())


def began_dloss(rows, B, state, loss_out, loss_slot, B_global=None, stream=None):
    _lib.call("gm_began_dloss_dp", stream or stream_ptr(), rows.data_ptr(), B,
              B if B_global is None else B_global, state.data_ptr(), loss_out.data_ptr(), loss_slot)


def began_update(state, dstate, istate, gamma, lam, patience, tick, stream=None):
    _lib.call("gm_began_update", stream or stream_ptr(), state.data_ptr(), dstate.data_ptr(),
              istate.data_ptr(), gamma, lam, patience, tick.data_ptr() if tick is not None else None)


STD_WS_BYTES = 1088                  # include/gm_hip.h GM_STD_WS_BYTES


def std_workspace(device):
    """Zeroed workspace of the multi-workgroup std kernels (allocate once, outside graph capture)."""
    import torch
    return torch.zeros(STD_WS_BYTES // 8, dtype=torch.float64, device=device)


def std_sums(X, R, out2, ws=None, stream=None):
    ws = std_workspace(X.device) if ws is None else ws
    _lib.call("gm_std_sums", stream or stream_ptr(), X.data_ptr(), _ld(X), R, X.shape[1], out2.data_ptr(),
              ws.data_ptr())


def std_from_sums(sums2, n_total, out, stream=None):
    _lib.call("gm_std_from_sums", stream or stream_ptr(), sums2.data_ptr(), n_total, out.data_ptr())


def std_all(X, R, out, ws=None, stream=None):
    ws = std_workspace(X.device) if ws is None else ws
    _lib.call("gm_std_all", stream or stream_ptr(), X.data_ptr(), _ld(X), R, X.shape[1], out.data_ptr(),
              ws.data_ptr())


def dragan_xhat(x, delta, delta_slot, U, u_slot, std_dev, out, B, C=1.0, stream=None):
    _lib.call("gm_dragan_xhat", stream or stream_ptr(), x.data_ptr(), _ld(x), delta.data_ptr(),
              delta_slot, U.data_ptr(), u_slot, std_dev.data_ptr(), C, out.data_ptr(), _ld(out), B,
              out.shape[1])


def dragan_rows(s, V, dv, da2, pen, lam, inv_b, B, K=1.0, stream=None):
    _lib.call("gm_dragan_rows", stream or stream_ptr(), s.data_ptr(), V.data_ptr(), _ld(V),
              dv.data_ptr(), _ld(dv), da2.data_ptr(), pen.data_ptr(), lam, inv_b, K, B, V.shape[1])


def fisher_commit(aux, stream=None):
    """aux[0] = aux[5]: lambda's successor, computed by the folded Fisher critic step, becomes lambda."""
    _lib.call("gm_fisher_commit", stream or stream_ptr(), aux.data_ptr())


def dragan_head_bwd(H, T, da2, w2, gw2, gb2, dA1, B, store=False, stream=None):
    """store: gw2 / gb2 receive the penalty's share alone (for the head backward's gw2_add / gb2_add)."""
    _lib.call("gm_dragan_head_bwd_store" if store else "gm_dragan_head_bwd", stream or stream_ptr(), H.data_ptr(), _ld(H), T.data_ptr(), _ld(T),
              da2.data_ptr(), w2.data_ptr(), gw2.data_ptr(), gb2.data_ptr(), dA1.data_ptr(), _ld(dA1),
              B, H.shape[1])
