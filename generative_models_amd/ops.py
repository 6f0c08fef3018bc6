"""Tensor-level wrappers over the C-ABI (include/gm_hip.h) + autograd Functions.

Everything here launches hand-written gfx950 kernels from libgm_hip.so on torch's current HIP
stream.  PyTorch only owns the memory.  No op has a torch-eager fallback."""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import ACT, LOSS, NO_SLOT, Slot, slot  # noqa: F401  (re-exported)


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise _lib.GMError("%s must be a float32 device tensor (got %s on %s)" %
                           (name, t.dtype, t.device))
    return t


def _ld(t):
    """Leading dimension of a 2-D (or 1-D treated as a column) row-major view."""
    if t.dim() == 1:
        return 1
    if t.stride(1) != 1:
        raise _lib.GMError("inner dimension must be contiguous")
    return t.stride(0)


# ---- raw ops (pointers may be offset views; M/K/N given explicitly) -------------------------
def linear_fwd(x, W, b, y, act, M=None, x_slot=NO_SLOT, stream=None):
    """y[M,N] = act(x[M,K] @ W[N,K]^T + b).  ns_gan.py:44-45,58-59."""
    N, K = W.shape
    M = x.shape[0] if M is None else M
    _lib.call("gm_linear_fwd", stream or stream_ptr(), _chk(x, "x").data_ptr(), _ld(x), x_slot,
              _chk(W, "W").data_ptr(), b.data_ptr() if b is not None else None,
              _chk(y, "y").data_ptr(), _ld(y), M, K, N, ACT[act] if not isinstance(act, int) else act)
    return y


def linear_fwd_interp(x, W, b, y, act, eps, eps_slot, x_real, x_hat, rows, M=None, x_slot=NO_SLOT,
                      stream=None):
    """linear_fwd + WGAN-GP's x_hat = eps*x_real + (1-eps)*y for the first `rows` output rows
    (w_gp_gan.py:197-201), written by the same launch."""
    N, K = W.shape
    M = x.shape[0] if M is None else M
    _lib.call("gm_linear_fwd_interp", stream or stream_ptr(), _chk(x, "x").data_ptr(), _ld(x), x_slot,
              _chk(W, "W").data_ptr(), b.data_ptr() if b is not None else None,
              _chk(y, "y").data_ptr(), _ld(y), M, K, N, ACT[act] if not isinstance(act, int) else act,
              eps.data_ptr(), eps_slot, _chk(x_real, "x_real").data_ptr(), _ld(x_real),
              _chk(x_hat, "x_hat").data_ptr(), _ld(x_hat), rows)
    return y


def linear_bwd_dx(dA, W, dX, below=None, epi="id", M=None, add=None, add_scale=1.0, stream=None):
    """dX[M,K] = (dA[M,N] @ W[N,K] + add_scale*add) (* act'(below))."""
    N, K = W.shape
    M = dA.shape[0] if M is None else M
    if add is not None:
        _lib.call("gm_linear_bwd_dx_add", stream or stream_ptr(), _chk(dA, "dA").data_ptr(), _ld(dA),
                  W.data_ptr(), _chk(dX, "dX").data_ptr(), _ld(dX),
                  below.data_ptr() if below is not None else None,
                  _ld(below) if below is not None else 0, M, K, N,
                  ACT[epi] if not isinstance(epi, int) else epi, add.data_ptr(), _ld(add), add_scale)
        return dX
    _lib.call("gm_linear_bwd_dx", stream or stream_ptr(), _chk(dA, "dA").data_ptr(), _ld(dA),
              W.data_ptr(), _chk(dX, "dX").data_ptr(), _ld(dX),
              below.data_ptr() if below is not None else None,
              _ld(below) if below is not None else 0, M, K, N,
              ACT[epi] if not isinstance(epi, int) else epi)
    return dX


def linear_bwd_dw(dA, X, dW, db, M=None, accumulate=False, x_slot=NO_SLOT, stream=None):
    """dW[N,K] (+)= dA[M,N]^T @ X[M,K]; db[N] (+)= colsum(dA)."""
    N, K = dW.shape
    M = dA.shape[0] if M is None else M
    _lib.call("gm_linear_bwd_dw", stream or stream_ptr(), _chk(dA, "dA").data_ptr(), _ld(dA),
              _chk(X, "X").data_ptr(), _ld(X), x_slot, dW.data_ptr(),
              db.data_ptr() if db is not None else None, M, K, N, 1 if accumulate else 0)


def linear_bwd_dw_adam(dA, X, lin, adam, M=None, x_slot=NO_SLOT, betas=(0.9, 0.999), eps=1e-8,
                       weight_decay=0.0, stream=None):
    """dW/db as linear_bwd_dw (written into lin.gW/lin.gb) with Adam applied to the layer's
    parameters in the same kernel.  lin: engine._Linear (W, b, gW, gb, mW, vW, mb, vb views);
    adam: dict(sched, sched_slot, clamp)."""
    N, K = lin.gW.shape
    M = dA.shape[0] if M is None else M
    _lib.call("gm_linear_bwd_dw_adam", stream or stream_ptr(), _chk(dA, "dA").data_ptr(), _ld(dA),
              _chk(X, "X").data_ptr(), _ld(X), x_slot, lin.gW.data_ptr(), lin.gb.data_ptr(), M, K, N,
              lin.W.data_ptr(), lin.mW.data_ptr(), lin.vW.data_ptr(), lin.b.data_ptr(),
              lin.mb.data_ptr(), lin.vb.data_ptr(), adam["sched"].data_ptr(), adam["sched_slot"],
              betas[0], betas[1], eps, weight_decay, adam.get("clamp", 0.0))


def linear_fwd_gather(x, W, b, y, act, data, idx, out, M=None, B=None, x_slot=NO_SLOT,
                      idx_slot=NO_SLOT, stream=None):
    """linear_fwd(x, W, b, y, act) and gather_rows(data, idx, out) as ONE launch (the gather
    workgroups ride in the GEMM's grid; `out` must not be an operand of this GEMM)."""
    N, K = W.shape
    M = x.shape[0] if M is None else M
    n_rows, row = data.shape
    B = out.shape[0] if B is None else B
    if isinstance(data, PackedData) and out.dtype == torch.int32:     # ... and the rows stay packed (out: [B, wpr] words)
        assert out.is_contiguous() and out.shape[1] == data.wpr
        _lib.call("gm_linear_fwd_gather_bits_packed", stream or stream_ptr(), _chk(x, "x").data_ptr(), _ld(x),
                  x_slot, _chk(W, "W").data_ptr(), b.data_ptr() if b is not None else None,
                  _chk(y, "y").data_ptr(), _ld(y), M, K, N, ACT[act], data.data_ptr(), data.wpr,
                  n_rows, idx.data_ptr(), idx_slot, out.data_ptr(), B)
        return y
    if isinstance(data, PackedData):                 # 1 bit / pixel resident dataset
        _lib.call("gm_linear_fwd_gather_bits", stream or stream_ptr(), _chk(x, "x").data_ptr(), _ld(x),
                  x_slot, _chk(W, "W").data_ptr(), b.data_ptr() if b is not None else None,
                  _chk(y, "y").data_ptr(), _ld(y), M, K, N, ACT[act], data.data_ptr(), data.wpr,
                  n_rows, idx.data_ptr(), idx_slot, out.data_ptr(), _ld(out), B, row)
        return y
    _lib.call("gm_linear_fwd_gather", stream or stream_ptr(), _chk(x, "x").data_ptr(), _ld(x),
              x_slot, _chk(W, "W").data_ptr(), b.data_ptr() if b is not None else None,
              _chk(y, "y").data_ptr(), _ld(y), M, K, N, ACT[act], _chk(data, "data").data_ptr(),
              n_rows, idx.data_ptr(), idx_slot, out.data_ptr(), _ld(out), B, row)
    return y


def _dw_adam_args(dA, X, lin, adam, M, x_slot, betas, eps, weight_decay):
    from ._lib import DwAdamArgs
    N, K = lin.gW.shape
    a = DwAdamArgs()
    a.dA, a.lda, a.X, a.ldx, a.x_slot = _chk(dA, "dA").data_ptr(), _ld(dA), _chk(X, "X").data_ptr(), _ld(X), x_slot
    a.dW, a.db, a.M, a.K, a.N = lin.gW.data_ptr(), lin.gb.data_ptr(), (dA.shape[0] if M is None else M), K, N
    if adam is not None:                        # None: plain gradients, no optimizer step
        a.pW, a.mW, a.vW = lin.W.data_ptr(), lin.mW.data_ptr(), lin.vW.data_ptr()
        a.pb, a.mb, a.vb = lin.b.data_ptr(), lin.mb.data_ptr(), lin.vb.data_ptr()
        a.sched, a.sched_slot = adam["sched"].data_ptr(), adam["sched_slot"]
        a.clamp = adam.get("clamp", 0.0)
    a.beta1, a.beta2, a.eps, a.weight_decay = betas[0], betas[1], eps, weight_decay
    return a


def linear_bwd_dw_adam_pair(first, second, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                            stream=None):
    """Two linear_bwd_dw_adam calls over the same batch rows as ONE launch.  first / second:
    dict(dA, X, lin, adam, M=None, x_slot=NO_SLOT)."""
    import ctypes
    mk = lambda d: _dw_adam_args(d["dA"], d["X"], d["lin"], d.get("adam"), d.get("M"),
                                 d.get("x_slot", NO_SLOT), betas, eps, weight_decay)
    a, b = mk(first), mk(second)
    _lib.call("gm_linear_bwd_dw_adam_pair", stream or stream_ptr(), ctypes.byref(a), ctypes.byref(b))


def linear_bwd_dw_adam_pair_finalize(first, second, fin, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                                     stream=None):
    """linear_bwd_dw_adam_pair as the LAST launch of a VAE batch: one more workgroup runs sum_finalize2's two sums and
    the last workgroup to finish advances the step counter.  fin: dict(pa, na, out_a, slot_a, pb, nb, out_b, slot_b,
    done (one zeroed int32 the launch counts its workgroups on), tick=None, scale_a=1.0, scale_b=1.0)."""
    import ctypes
    from ._lib import Finalize2Args
    mk = lambda d: _dw_adam_args(d["dA"], d["X"], d["lin"], d.get("adam"), d.get("M"),
                                 d.get("x_slot", NO_SLOT), betas, eps, weight_decay)
    a, b = mk(first), mk(second)
    f = Finalize2Args()
    f.pa, f.na, f.scale_a, f.out_a, f.slot_a = fin["pa"].data_ptr(), fin["na"], fin.get("scale_a", 1.0), \
        fin["out_a"].data_ptr(), fin["slot_a"]
    f.pb, f.nb, f.scale_b, f.out_b, f.slot_b = fin["pb"].data_ptr(), fin["nb"], fin.get("scale_b", 1.0), \
        fin["out_b"].data_ptr(), fin["slot_b"]
    tick = fin.get("tick")
    f.tick = tick.data_ptr() if tick is not None else None
    assert fin["done"].dtype == torch.int32 and fin["done"].numel() >= 1
    f.done = fin["done"].data_ptr()
    _lib.call("gm_linear_bwd_dw_adam_pair_finalize", stream or stream_ptr(), ctypes.byref(a), ctypes.byref(b),
              ctypes.byref(f))


def _head_args(head, betas=(0.9, 0.999), eps=1e-8):
    """gm_head_bwd_args from dict(H, dS, lin (head _Linear), rowloss, loss_out, loss_slot, inv_b, B,
    gen_mode=False, adam=None (dict(sched, sched_slot, clamp)), tick=None, grads=True).  dS / rowloss
    may be None for the folded head (they are rebuilt from the forward's partial dots)."""
    from ._lib import HeadBwdArgs
    hl, ha, H = head["lin"], head.get("adam"), head["H"]
    gen = bool(head.get("gen_mode", False))
    a = HeadBwdArgs()
    ptr = lambda t: t.data_ptr() if t is not None else None
    a.H, a.ldh, a.dS = H.data_ptr(), _ld(H), ptr(head.get("dS"))
    a.w2, a.b2, a.rowloss = hl.W.data_ptr(), hl.b.data_ptr(), ptr(head.get("rowloss"))
    a.dH, a.lddh = None, 0                      # written by head_fwd_loss
    if not gen:
        a.gw2, a.gb2 = hl.gW.data_ptr(), hl.gb.data_ptr()
    a.loss_out, a.loss_slot = head["loss_out"].data_ptr(), head["loss_slot"]
    a.inv_b, a.gen_mode, a.B, a.Hd = head["inv_b"], 1 if gen else 0, head["B"], H.shape[1]
    a.with_adam = 1 if ha is not None else 0
    if ha is not None:
        a.mW, a.vW, a.mb, a.vb = hl.mW.data_ptr(), hl.vW.data_ptr(), hl.mb.data_ptr(), hl.vb.data_ptr()
        a.sched, a.sched_slot = ha["sched"].data_ptr(), ha["sched_slot"]
        a.clamp = ha.get("clamp", 0.0)
    a.beta1, a.beta2, a.eps, a.weight_decay = betas[0], betas[1], eps, 0.0
    tick = head.get("tick")
    a.tick = tick.data_ptr() if tick is not None else None
    add = head.get("gw2_add")
    a.gw2_add = add.data_ptr() if add is not None else None
    addb = head.get("gb2_add")
    a.gb2_add = addb.data_ptr() if addb is not None else None
    pen = head.get("pen")                       # dict(s=[rows], h=[rows, Hd], t=[rows, Hd]): see gm_hip.h
    if pen is not None:
        a.pen_s, a.pen_h, a.pen_ldh = pen["s"].data_ptr(), pen["h"].data_ptr(), _ld(pen["h"])
        a.pen_t, a.pen_ldt, a.pen_rows = pen["t"].data_ptr(), _ld(pen["t"]), pen["t"].shape[0]
    return a


def linear_fwd_sqerr(x, W, b, y, target, dA, part, M=None, stream=None):
    """Sigmoid output layer + the reconstruction loss in its epilogue (vae.py:203): y = sigmoid(xW^T+b),
    dA = d sum((target - y)^2) / d (pre-sigmoid), part[m, j] = the row's squared error inside 32-column
    tile j (part: [>= M, >= ceil(N/32)], zero-initialised; sum it with ops_fused.sum_finalize*)."""
    N, K = W.shape
    M = x.shape[0] if M is None else M
    assert part.dim() == 2 and part.shape[0] >= M and part.shape[1] >= (N + 31) // 32
    _lib.call("gm_linear_fwd_sqerr", stream or stream_ptr(), _chk(x, "x").data_ptr(), _ld(x),
              _chk(W, "W").data_ptr(), b.data_ptr() if b is not None else None, _chk(y, "y").data_ptr(), _ld(y),
              M, K, N, _chk(target, "target").data_ptr(), _ld(target), _chk(dA, "dA").data_ptr(), _ld(dA),
              part.data_ptr(), part.shape[1])
    return y


def linear_bwd_dx_reparam(dA, W, dZ, ml, eps, dml, M=None, eps_slot=NO_SLOT, stream=None):
    """dZ = dA W through the decoder's first layer (W: [N, Z]) + the reparameterisation / KL backward in
    the epilogue: dml = [dZ + mu | dZ*eps*exp(lv/2)/2 + (exp(lv)-1)/2]   (vae.py:100-106,210-212)."""
    N, Z = W.shape
    M = dA.shape[0] if M is None else M
    _lib.call("gm_linear_bwd_dx_reparam", stream or stream_ptr(), _chk(dA, "dA").data_ptr(), _ld(dA),
              _chk(W, "W").data_ptr(), _chk(dZ, "dZ").data_ptr(), _ld(dZ), M, Z, N, ml.data_ptr(), _ld(ml),
              eps.data_ptr(), eps_slot, dml.data_ptr(), _ld(dml))
    return dZ


def lds_min_m():
    """Rows from which forward / dX launches take the LDS macro-tile kernel (csrc/gm_gemm.hip)."""
    import os
    return 1024


class HeadFold:
    """Buffers of the folded critic head (gm_head_fold_args): the hidden layer's forward leaves per-
    column-tile partial dots with w2 and a snapshot of (w2, b2); the launches either side of the N = 1
    layer rebuild scores / losses / dS from them (ns_gan.py:57-60,191-192,214)."""

    def __init__(self, rows, hidden, device):
        self.nparts = (hidden + 31) // 32
        assert self.nparts <= 16, "folded head: hidden layers up to 512 wide"
        # part[r, j]: a row's partial dots are contiguous (one 64-byte line for <= 16 column tiles);
        # entries j >= nparts are never written and stay zero
        self.part = torch.zeros(rows, _al4(self.nparts), device=device)
        self.snap = torch.zeros(_al4(hidden + 1), device=device)
        self.hidden = hidden

    def args(self, variant, out_act, hyper=(), pen=None, S=None, dS=None, rowloss=None):
        from ._lib import HeadFoldArgs
        a = HeadFoldArgs()
        a.part, a.ldp, a.nparts, a.snap = self.part.data_ptr(), self.part.shape[1], self.nparts, self.snap.data_ptr()   # ldp = row stride
        a.variant = LOSS[variant] if not isinstance(variant, int) else variant
        a.out_act = ACT[out_act] if not isinstance(out_act, int) else out_act
        hy = tuple(hyper)
        assert len(hy) <= 8
        for i, h in enumerate(hy):
            a.hyper[i] = float(h)
        a.n_hyper = len(hy)
        ptr = lambda t: t.data_ptr() if t is not None else None
        a.pen, a.S, a.dS, a.rowloss = ptr(pen), ptr(S), ptr(dS), ptr(rowloss)
        return a


def _al4(n):
    return (n + 3) // 4 * 4


def linear_fwd_headpart(x, W, b, y, act, head_lin, fold, M=None, x_slot=NO_SLOT, stream=None, xbits=None):
    """linear_fwd of the critic's hidden layer that also leaves the folded head's partial dots and
    the (w2, b2) snapshot in `fold` (HeadFold).  xbits = (words, words_per_row, rows): the first `rows` rows of x
    are read from the packed copy (gather_rows_packed) instead."""
    N, K = W.shape
    M = x.shape[0] if M is None else M
    assert fold.hidden == N and fold.part.shape[0] >= M
    if xbits is not None:
        _lib.call("gm_linear_fwd_headpart_bits", stream or stream_ptr(), _chk(x, "x").data_ptr(), _ld(x), x_slot,
                  _chk(W, "W").data_ptr(), b.data_ptr() if b is not None else None, _chk(y, "y").data_ptr(),
                  _ld(y), M, K, N, ACT[act] if not isinstance(act, int) else act, head_lin.W.data_ptr(),
                  head_lin.b.data_ptr(), fold.part.data_ptr(), fold.part.shape[1], fold.snap.data_ptr(),
                  xbits[0].data_ptr(), xbits[1], xbits[2])
        return y
    _lib.call("gm_linear_fwd_headpart", stream or stream_ptr(), _chk(x, "x").data_ptr(), _ld(x), x_slot,
              _chk(W, "W").data_ptr(), b.data_ptr() if b is not None else None, _chk(y, "y").data_ptr(),
              _ld(y), M, K, N, ACT[act] if not isinstance(act, int) else act, head_lin.W.data_ptr(),
              head_lin.b.data_ptr(), fold.part.data_ptr(), fold.part.shape[1], fold.snap.data_ptr())
    return y


def linear_bwd_dx_head_fold(H, W, dX, head, fold_args, below=None, epi="id", M=None, stream=None):
    """linear_bwd_dx_head in the folded form: A operand = the hidden activations H (dH is formed in
    registers from the rows' dS and the snapshot of w2); head: see _head_args (dS / rowloss unused)."""
    import ctypes
    N, K = W.shape
    M = H.shape[0] if M is None else M
    a = _head_args(head)
    _lib.call("gm_linear_bwd_dx_head_fold", stream or stream_ptr(), _chk(H, "H").data_ptr(), _ld(H),
              _chk(W, "W").data_ptr(), _chk(dX, "dX").data_ptr(), _ld(dX),
              below.data_ptr() if below is not None else None, _ld(below) if below is not None else 0,
              M, K, N, ACT[epi] if not isinstance(epi, int) else epi, ctypes.byref(a), ctypes.byref(fold_args))
    return dX


def linear_bwd_dw_adam_head_fold(H, X, lin, adam, head, fold_args, M=None, x_slot=NO_SLOT,
                                 betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, stream=None, xbits=None):
    """linear_bwd_dw_adam_head in the folded form (see linear_bwd_dx_head_fold).  xbits: as linear_fwd_headpart."""
    import ctypes
    N, K = lin.gW.shape
    M = H.shape[0] if M is None else M
    a = _head_args(head, betas, eps)
    name = "gm_linear_bwd_dw_adam_head_fold" + ("_bits" if xbits is not None else "")
    tail = (xbits[0].data_ptr(), xbits[1], xbits[2]) if xbits is not None else ()
    if adam is None:
        _lib.call(name, stream or stream_ptr(), _chk(H, "H").data_ptr(),
                  _ld(H), _chk(X, "X").data_ptr(), _ld(X), x_slot, lin.gW.data_ptr(),
                  lin.gb.data_ptr(), M, K, N, None, None, None, None, None, None, None, NO_SLOT,
                  betas[0], betas[1], eps, weight_decay, 0.0, ctypes.byref(a), ctypes.byref(fold_args), *tail)
        return
    _lib.call(name, stream or stream_ptr(), _chk(H, "H").data_ptr(),
              _ld(H), _chk(X, "X").data_ptr(), _ld(X), x_slot, lin.gW.data_ptr(),
              lin.gb.data_ptr(), M, K, N, lin.W.data_ptr(), lin.mW.data_ptr(), lin.vW.data_ptr(),
              lin.b.data_ptr(), lin.mb.data_ptr(), lin.vb.data_ptr(), adam["sched"].data_ptr(),
              adam["sched_slot"], betas[0], betas[1], eps, weight_decay, adam.get("clamp", 0.0),
              ctypes.byref(a), ctypes.byref(fold_args), *tail)


def linear_bwd_dx_head(dA, W, dX, head, below=None, epi="id", M=None, stream=None):
    """linear_bwd_dx with the head's backward workgroups riding in the launch (generator step: the
    scalar workgroup that writes the loss and ticks).  head: see _head_args."""
    import ctypes
    N, K = W.shape
    M = dA.shape[0] if M is None else M
    a = _head_args(head)
    _lib.call("gm_linear_bwd_dx_head", stream or stream_ptr(), _chk(dA, "dA").data_ptr(), _ld(dA),
              _chk(W, "W").data_ptr(), _chk(dX, "dX").data_ptr(), _ld(dX),
              below.data_ptr() if below is not None else None, _ld(below) if below is not None else 0,
              M, K, N, ACT[epi] if not isinstance(epi, int) else epi, ctypes.byref(a))
    return dX


def linear_bwd_dw_adam_head(dA, X, lin, adam, head, M=None, x_slot=NO_SLOT, betas=(0.9, 0.999),
                            eps=1e-8, weight_decay=0.0, ones_from=0, stream=None):
    """linear_bwd_dw_adam and the critic head's backward (ops_fused.head_bwd with Adam on the head
    layer) as ONE launch.  head: dict(H, dS, lin (head _Linear), rowloss, loss_out, loss_slot,
    inv_b, B, adam (dict(sched, sched_slot, clamp)), gw2_add=None); dH must already be written by
    head_fwd_loss.  ones_from: the first rows of the (stacked) reduction that do not reach db."""
    import ctypes
    N, K = lin.gW.shape
    M = dA.shape[0] if M is None else M
    a = _head_args(head, betas, eps)
    if adam is None:                            # plain gradients (the head's dict has adam=None too)
        _lib.call("gm_linear_bwd_dw_adam_head_ex", stream or stream_ptr(), _chk(dA, "dA").data_ptr(),
                  _ld(dA), _chk(X, "X").data_ptr(), _ld(X), x_slot, lin.gW.data_ptr(),
                  lin.gb.data_ptr(), M, K, N, None, None, None, None, None, None, None, NO_SLOT,
                  betas[0], betas[1], eps, weight_decay, 0.0, ctypes.byref(a), ones_from)
        return
    _lib.call("gm_linear_bwd_dw_adam_head_ex", stream or stream_ptr(), _chk(dA, "dA").data_ptr(),
              _ld(dA), _chk(X, "X").data_ptr(), _ld(X), x_slot, lin.gW.data_ptr(),
              lin.gb.data_ptr(), M, K, N, lin.W.data_ptr(), lin.mW.data_ptr(), lin.vW.data_ptr(),
              lin.b.data_ptr(), lin.mb.data_ptr(), lin.vb.data_ptr(), adam["sched"].data_ptr(),
              adam["sched_slot"], betas[0], betas[1], eps, weight_decay, adam.get("clamp", 0.0),
              ctypes.byref(a), ones_from)


class PackedData:
    """Device-resident dataset at 1 bit per pixel.  utils.py:31,34 binarises
    MNIST with torch.bernoulli, so every pixel is exactly 0.0 or 1.0: a row of 784 pixels is 25
    uint32 words (100 B) instead of 3136 B.  The gather kernels expand the selected rows back to the
    fp32 rows the GEMMs read; nothing else ever touches the dataset."""

    def __init__(self, images_2d):
        """images_2d: [N, I] tensor (any device) holding only 0.0 / 1.0."""
        x = images_2d.detach()
        self.shape = tuple(x.shape)
        n, i = self.shape
        self.wpr = (i + 31) // 32
        b = (x.to("cpu") != 0).numpy()
        packed = np.packbits(b, axis=1, bitorder="little")            # pixel i -> bit (i & 7) of byte i >> 3
        pad = self.wpr * 4 - packed.shape[1]
        if pad:
            packed = np.concatenate([packed, np.zeros((n, pad), dtype=np.uint8)], axis=1)
        words = np.ascontiguousarray(packed).view("<u4")              # little-endian: pixel i = bit i & 31
        dev = images_2d.device if images_2d.is_cuda else torch.device("cuda")
        self.bits = torch.from_numpy(words.view(np.int32).copy()).to(dev)

    @staticmethod
    def is_binary(images):
        return bool(((images == 0) | (images == 1)).all())

    def data_ptr(self):
        return self.bits.data_ptr()

    def nbytes(self):
        return self.bits.numel() * 4


def gather_rows(data, idx, out, B=None, idx_slot=NO_SLOT, stream=None):
    """out[b,:] = data[idx[b],:]  (process_batch, ns_gan.py:222-226).  data: fp32 [N, I] or PackedData."""
    n_rows, row = data.shape
    B = out.shape[0] if B is None else B
    assert idx.dtype == torch.int64 and idx.is_cuda
    if isinstance(data, PackedData) and out.dtype == torch.int32:     # rows copied as words: out [B, wpr]
        assert out.is_contiguous() and out.shape[1] == data.wpr
        _lib.call("gm_gather_rows_bits_packed", stream or stream_ptr(), data.data_ptr(), data.wpr, n_rows,
                  idx.data_ptr(), idx_slot, out.data_ptr(), B)
        return out
    if isinstance(data, PackedData):
        _lib.call("gm_gather_rows_bits", stream or stream_ptr(), data.data_ptr(), data.wpr, n_rows,
                  idx.data_ptr(), idx_slot, out.data_ptr(), _ld(out), B, row)
        return out
    _lib.call("gm_gather_rows", stream or stream_ptr(), data.data_ptr(), n_rows, idx.data_ptr(),
              idx_slot, out.data_ptr(), _ld(out), B, row)
    return out


def gan_loss(variant, gen_mode, sx, sg, B, out_act, loss_out, dax, dag, hyper=(), inv_b=None,
             loss_slot=NO_SLOT, aux=None, db=None, stream=None, phase=0, pre=None, loss_scale=1.0):
    """Adversarial loss + d(loss)/d(pre-activation score).  SURVEY.md appendix A.2.
    phase / pre / loss_scale: the data-parallel, phased form of the RaGAN / Fisher critic losses
    (gm_gan_loss_phase)."""
    h = (ctypes.c_float * 8)(*([float(x) for x in hyper] + [0.0] * (8 - len(hyper))))
    inv_b = float(np.float32(1.0) / np.float32(B)) if inv_b is None else float(inv_b)
    if phase:
        _lib.call("gm_gan_loss_phase", stream or stream_ptr(), LOSS[variant] if isinstance(variant, str) else variant,
                  1 if gen_mode else 0, sx.data_ptr() if sx is not None else None, sg.data_ptr(), B,
                  ACT[out_act] if not isinstance(out_act, int) else out_act, h, len(hyper), inv_b,
                  loss_out.data_ptr(), loss_slot, dax.data_ptr() if dax is not None else None,
                  dag.data_ptr() if dag is not None else None,
                  aux.data_ptr() if aux is not None else None, db.data_ptr() if db is not None else None,
                  phase, pre.data_ptr(), loss_scale)
        return
    _lib.call("gm_gan_loss", stream or stream_ptr(), LOSS[variant] if isinstance(variant, str) else variant,
              1 if gen_mode else 0, sx.data_ptr() if sx is not None else None, sg.data_ptr(), B,
              ACT[out_act] if not isinstance(out_act, int) else out_act, h, len(hyper), inv_b,
              loss_out.data_ptr(), loss_slot, dax.data_ptr() if dax is not None else None,
              dag.data_ptr() if dag is not None else None,
              aux.data_ptr() if aux is not None else None,
              db.data_ptr() if db is not None else None)


def adam(p, g, m, v, sched, sched_slot=NO_SLOT, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
         clamp=0.0, lr_scale=None, stream=None):
    """Flat Adam step, torch _single_tensor_adam order.  lr_scale: device float
    multiplying the learning rate (BEGAN's plateau schedulers)."""
    if lr_scale is not None:
        _lib.call("gm_adam_scaled", stream or stream_ptr(), p.data_ptr(), g.data_ptr(), m.data_ptr(),
                  v.data_ptr(), p.numel(), sched.data_ptr(), sched_slot, betas[0], betas[1], eps,
                  weight_decay, clamp, lr_scale.data_ptr())
        return
    _lib.call("gm_adam", stream or stream_ptr(), p.data_ptr(), g.data_ptr(), m.data_ptr(),
              v.data_ptr(), p.numel(), sched.data_ptr(), sched_slot, betas[0], betas[1], eps,
              weight_decay, clamp)


def act_bwd(dY, Y, dA, act, stream=None):
    _lib.call("gm_act_bwd", stream or stream_ptr(), dY.data_ptr(), Y.data_ptr(), dA.data_ptr(),
              dY.numel(), ACT[act])
    return dA


def copy_slot(src, dst, n, src_slot=NO_SLOT, dst_slot=NO_SLOT, stream=None):
    """dst[dst_slot + i] = src[src_slot + i] for i < n fp32 words (gm_copy_slot_f32)."""
    _lib.call("gm_copy_slot_f32", stream or stream_ptr(), src.data_ptr(), src_slot, dst.data_ptr(), dst_slot, n)


def tick(ctr, inc=1, stream=None):
    _lib.call("gm_tick", stream or stream_ptr(), ctr.data_ptr(), inc)


def adam_schedule(lr, n_steps, betas=(0.9, 0.999), start=1):
    """Per-step scalars exactly as torch computes them in Python doubles (adam.py:531-536):
    step_size = lr / (1 - b1**step), bc2_sqrt = (1 - b2**step) ** 0.5; cast to fp32 like the
    scalar operands of addcdiv_/div.  Returns float32 [n_steps, 2]."""
    out = np.empty((n_steps, 2), dtype=np.float32)
    for i in range(n_steps):
        step = start + i
        bc1 = 1 - betas[0] ** step
        bc2 = 1 - betas[1] ** step
        out[i, 0] = lr / bc1
        out[i, 1] = bc2 ** 0.5
    return out


def randperm_prefix(seed, n, B, out=None):
    """HOST: first B entries of torch.randperm(n, generator=Generator().manual_seed(seed))."""
    if out is None:
        out = np.empty(B, dtype=np.int64)
    _lib.call("gm_randperm_prefix", seed & 0xFFFFFFFFFFFFFFFF, n, B, out.ctypes.data)
    return out


# ---- HIP graph wrapper --------------------------------------------------------------------
class Graph:
    """A captured sequence of gm_* launches (hipGraph).  Capture happens on a side stream."""

    def __init__(self):
        self.exec = None
        self._stream = torch.cuda.Stream()

    def capture(self, fn):
        s = self._stream
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            _lib.call("gm_graph_begin", s.cuda_stream)
            try:
                fn(s.cuda_stream)
            finally:
                out = ctypes.c_void_p()
                _lib.call("gm_graph_end", s.cuda_stream, ctypes.byref(out))
            self.exec = out
        torch.cuda.current_stream().wait_stream(s)
        return self

    def launch(self, stream=None):
        _lib.call("gm_graph_launch", self.exec, stream or stream_ptr())

    def __del__(self):
        try:
            if self.exec:
                _lib.load().gm_graph_destroy(self.exec)
        except Exception:
            pass


class Event:
    """HIP event on an explicit stream (bench.py roofline timing)."""

    def __init__(self):
        self.h = ctypes.c_void_p()
        _lib.call("gm_event_create", ctypes.byref(self.h))

    def record(self, stream=None):
        _lib.call("gm_event_record", self.h, stream or stream_ptr())

    def sync(self):
        _lib.call("gm_event_sync", self.h)

    def elapsed_ms(self, stop):
        ms = ctypes.c_float()
        _lib.call("gm_event_elapsed_ms", self.h, stop.h, ctypes.byref(ms))
        return ms.value

    def __del__(self):
        try:
            if self.h:
                _lib.load().gm_event_destroy(self.h)
        except Exception:                             # noqa: BLE001  (interpreter teardown)
            pass


# ---- general autograd path (user-overridden train_D / train_G; README.md:29-31) -----------
class _MM(torch.autograd.Function):
    """C = op(A, B) for the three GEMM layouts; closed under differentiation, so arbitrary-order
    autograd (WGAN-GP style create_graph=True) works on top of the HIP kernels.
    kind: 'nt' A[M,K] B[N,K]^T ; 'nn' A[M,N] B[N,K] ; 'tn' A[M,N]^T B[M,K]."""

    @staticmethod
    def forward(ctx, A, B, kind):
        A, B = A.contiguous(), B.contiguous()
        ctx.kind = kind
        ctx.save_for_backward(A, B)
        if kind == "nt":
            C = torch.empty(A.shape[0], B.shape[0], device=A.device)
            linear_fwd(A, B, None, C, "id")
        elif kind == "nn":
            C = torch.empty(A.shape[0], B.shape[1], device=A.device)
            linear_bwd_dx(A, B, C)
        else:
            C = torch.empty(A.shape[1], B.shape[1], device=A.device)
            linear_bwd_dw(A, B, C, None)
        return C

    @staticmethod
    def backward(ctx, G):
        A, B = ctx.saved_tensors
        k = ctx.kind
        gA = gB = None
        if k == "nt":      # C = A B^T
            if ctx.needs_input_grad[0]: gA = _MM.apply(G, B, "nn")
            if ctx.needs_input_grad[1]: gB = _MM.apply(G, A, "tn")
        elif k == "nn":    # C = A B
            if ctx.needs_input_grad[0]: gA = _MM.apply(G, B, "nt")
            if ctx.needs_input_grad[1]: gB = _MM.apply(A, G, "tn")
        else:              # C = A^T B
            if ctx.needs_input_grad[0]: gA = _MM.apply(B, G, "nt")
            if ctx.needs_input_grad[1]: gB = _MM.apply(A, G, "nn")
        return gA, gB, None


class _FusedLinear(torch.autograd.Function):
    """y = act(x W^T + b) with the fused forward kernel; first-order backward on HIP kernels,
    higher-order through _MM."""

    @staticmethod
    def forward(ctx, x, W, b, act):
        x = x.contiguous()
        y = torch.empty(x.shape[0], W.shape[0], device=x.device)
        linear_fwd(x, W, b, y, act)
        ctx.act = act
        ctx.save_for_backward(x, W, y)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, W, y = ctx.saved_tensors
        act = ctx.act
        if torch.is_grad_enabled():        # create_graph=True: stay differentiable
            if act == "relu":
                dA = gy * (y > 0).to(gy.dtype)
            elif act == "sigmoid":
                dA = gy * (1 - y) * y
            else:
                dA = gy
            gx = _MM.apply(dA, W, "nn") if ctx.needs_input_grad[0] else None
            gW = _MM.apply(dA, x, "tn") if ctx.needs_input_grad[1] else None
            gb = dA.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
            return gx, gW, gb, None
        gy = gy.contiguous()
        dA = gy if act in ("id", None) else act_bwd(gy, y, torch.empty_like(gy), act)
        gx = gW = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            linear_bwd_dx(dA, W, gx)
        if ctx.needs_input_grad[1]:
            gW = torch.empty_like(W)
            gb = torch.empty(W.shape[0], device=W.device) if ctx.has_bias else None
            linear_bwd_dw(dA, x, gW, gb)
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            gb = dA.sum(0)
        return gx, gW, gb, None


def fused_linear(x, weight, bias, act):
    """Drop-in for act(F.linear(x, weight, bias)) on device tensors."""
    return _FusedLinear.apply(x, weight, bias, act)
