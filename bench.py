#!/usr/bin/env python
"""bench.py -- images/sec of the NSGAN D+G step (BASELINE.json metric), bs=256 per GPU, fp32.

    python bench.py --gpus N --steps K --warmup W
    (N>1: one rank per GPU; started under torch.distributed.run it uses the RANK/WORLD_SIZE it is
    given, started from a bare shell it re-launches itself under torch.distributed.run)

A "step" is one full reference iteration (ns_gan.py:122-156): process_batch -> train_D ->
backward -> Adam(D) -> train_G -> backward -> Adam(G), in PARITY mode (reference RNG protocol,
bit-exact sampling indices).  The dataset (synthetic 50 000 x 28x28 Bernoulli images, seed 3435)
is resident in HBM before the timed region; the host-side RNG prefetch is inside it.
Weak scaling: every rank runs bs=256 (global batch 256*N), gradients all-reduced over RCCL.
The K-step timed region (barrier + synchronize on both sides, max over ranks) is repeated
`--reps` times on fresh draws; `ms_per_step` / `value` are the MEDIAN repetition (all of them are
listed in config.reps_ms_per_step).  The host draws of a timed step happen inside its timed region.
At N=1 the same line carries `configs`: BASELINE.json configs 3/4/5 (WGAN-GP bs=256, VAE bs=512,
NSGAN/LSGAN bs=1024) measured the same way, each with its dominant GEMM's roofline fraction and a
bounded CPU-oracle baseline.  Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "generative_models_amd", "src"))

FLOP_PER_IMAGE = 6_326_400        # 10 U GEMMs + small, algorithmic (no wasted bwd)
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md:41
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md:35 (spec)
B_PER_GPU = 256
LDS_MIN_M = 1024    # csrc/gm_gemm.hip lds_cfg_for
IMG, HID, Z, N_TRAIN = 784, 400, 20, 50000
PROFILE_ROUND = "r06"             # profiles/<round>_* hold the rocprofv3 / PMC passes of the kernels named below
FOLD_HEAD_DEFAULT = os.environ.get("GM_FOLD_HEAD", "1") != "0"   # engine default (folded critic head)


_DATASET = None


def synthetic_dataset():
    """seed 3435, Bernoulli(0.1307) 50 000 x 1 x 28 x 28 (built once per process)."""
    global _DATASET
    if _DATASET is None:
        st = torch.get_rng_state()
        torch.manual_seed(3435)
        img = torch.bernoulli(torch.full((N_TRAIN, 1, 28, 28), 0.1307))
        _DATASET = torch.utils.data.TensorDataset(img, torch.zeros(N_TRAIN, dtype=torch.int64))
        torch.set_rng_state(st)
    return _DATASET


def gemm_shapes(B, fused_head=True, batch_gen=True, group_head=True, ride_gather=True, pair_dw=True,
                ride_head_dx=True, fold_head=False):
    """Every GEMM launch of one NSGAN iteration: (kind, M, K, N) in layer terms.  With the fused
    critic-head kernels (default) the N=1 layer is not a GEMM launch any more; with the batched
    generator forward (default at D_steps=1) G(zD) and G(zG) are one 2B-row launch pair."""
    d_head = [] if fused_head else [("fwd", 2 * B, HID, 1), ("dw", 2 * B, HID, 1), ("dx", 2 * B, HID, 1)]
    g_head = [] if fused_head else [("fwd", B, HID, 1), ("dx", B, HID, 1)]
    gen = [("fwd", 2 * B, Z, HID), ("fwd", 2 * B, HID, IMG)] if batch_gen else \
        [("fwd", B, Z, HID), ("fwd", B, HID, IMG)] * 2
    # "dwh": the first critic layer's weight gradient carrying the head's backward workgroups
    dw1 = "dwh" if (fused_head and group_head) else "dw"
    if ride_gather:                      # "fwdg": the batch gather rides in this launch's grid
        gen[0] = ("fwdg",) + gen[0][1:]
    # "dwp": both generator weight gradients as one launch (second GEMM: dW1, [HID, Z])
    g_dw = [("dwp", B, HID, IMG)] if pair_dw else [("dw", B, HID, IMG), ("dw", B, Z, HID)]
    if fold_head:
        # folded critic head: "fwdp" = hidden-layer forward that also leaves the head's partial dots,
        # "dwhf" / "dxhf" = the riding launches that rebuild dS and form dH in registers
        return (gen + [("fwdp", 2 * B, IMG, HID), ("dwhf", 2 * B, IMG, HID), ("fwdp", B, IMG, HID),
                       ("dxhf", B, IMG, HID), ("dx", B, HID, IMG)] + g_dw)
    return (gen + [("fwd", 2 * B, IMG, HID)] + d_head + [(dw1, 2 * B, IMG, HID)] +
            [("fwd", B, IMG, HID)] + g_head +
            # "dxh": the generator-mode head's scalar workgroup (loss + tick) rides in this launch
            [("dxh" if (fused_head and ride_head_dx) else "dx", B, IMG, HID), ("dx", B, HID, IMG)] + g_dw)


def gemm_shapes_wgp(B, fold_g=True):
    """Every GEMM launch of one WGAN-GP iteration at D_steps = 1 (engine._issue_gp_forward / _D_rest with
    the stacked layer-1 weight gradient): "dwhs" = [u ; dH]^T [gamma ; x ; G(z)] over 3B rows + head."""
    g = [("fwdp", B, IMG, HID), ("dxhf", B, IMG, HID)] if fold_g else [("fwd", B, IMG, HID), ("dxh", B, IMG, HID)]
    return ([("fwdg", 2 * B, Z, HID), ("fwd", 2 * B, HID, IMG),           # G(zD), G(zG) (+ gather, + x_hat epilogue)
             ("fwd", 3 * B, IMG, HID),                                    # D layer 1 on [x_hat ; x ; G(z)]: one launch
             ("dx", B, IMG, HID), ("fwd", B, IMG, HID),                   # g = u W1 ; t = gamma W1^T
             ("dwhs", 3 * B, IMG, HID)] + g + [("dx", B, HID, IMG), ("dwp", B, HID, IMG)])


def gemm_shapes_vae(B):
    """Every GEMM launch of one VAE training batch (engine.VAEEngine._issue): "fwds" = decoder output
    layer + reconstruction loss epilogue, "fwdz" = reparameterisation + the decoder's first layer (one launch; its GEMM
    workgroups form z from mu, log_var, eps), "bmid" = the two narrow GEMMs between the decoder's and the encoder's wide
    layers + the reparameterisation backward as one launch (round 4), "dxr" = dX through the decoder's first layer + reparameterisation
    backward epilogue; the two weight-gradient pairs carry their second GEMM's (N2, K2)."""
    return [("fwd", B, IMG, HID), ("fwdg", B, HID, 2 * Z), ("fwdz", B, Z, HID), ("fwds", B, HID, IMG),
            ("dx", B, HID, IMG), ("bmid", B, Z, HID), ("dwp", B, HID, IMG, (HID, Z)),
            ("dwp", B, IMG, HID, (2 * Z, HID))]


def gemm_variant(kind, M, K, N, extra=None):
    """Name of the kernel instantiation csrc/gm_gemm.hip launches for this layer shape with the
    default settings (mirrors launch<MODE>(): v_mfma_f32_16x16x4_f32 kernel, 16 waves, per-chunk
    load/consume schedule, 16-byte paths by alignment, tile shape from the tile count) -- the name rocprofv3 reports.  Template order: MODE, VEC, WAVES, G, XV, MI, NI, DMA, SL."""
    if kind == "fwdz":
        return "vae_reparam_fwd_kernel"
    if kind == "bmid":
        return "vae_bwd_mid_kernel"
    if kind in ("fwd", "fwdg", "fwdp", "fwds"):
        mode, Mg, Ng, Kr, vec, xv = 0, M, N, K, K % 4 == 0, False
    elif kind in ("dx", "dxh", "dxhf", "dxr"):
        mode, Mg, Ng, Kr, vec, xv = 1, M, K, N, N % 4 == 0, K % 4 == 0
    else:
        mode, Mg, Ng, Kr, vec = 2, N, K + 1, M, False
        xv = N % 4 == 0 and K % 4 == 0 and N >= 4 and K >= 4
    if kind in ("fwd", "dx", "dxh", "fwdg", "dxr") and Mg >= LDS_MIN_M and Kr >= 64 and Ng >= 32 and vec and (mode == 0 or xv):
        # many-row launches: the LDS-staged macro-tile kernel (riders get their own launch)
        cands = [(64, 64, "64, 64, 32, 64, 4, 1, 4"), (32, 64, "32, 64, 32, 32, 4, 1, 4")]
        cost = lambda c: (-(-(-(-Mg // c[0]) * -(-Ng // c[1])) // 256)) * c[0] * c[1]
        best = min(cands, key=lambda c: (cost(c), -c[0] * c[1]))
        ns = -(-Kr // 32)
        return "gemm_lds_kernel<%d, %s, %d>" % (mode, best[2], ns if Kr in (784, 400) else 0)
    if kind in ("fwd", "fwdg") and Kr <= 32 and vec:
        # forwards over a reduction of at most 32 (the generator's first layer): one wave per 16 x 32 piece, no cross-wave
        # reduction.  Arguments: ring slot on the operand (the noise ring: true in the engines), gather riders
        return "gemm16_k32_fwd_kernel<true, %s>" % ("true" if kind == "fwdg" else "false")
    nw = 16
    chunks = -(-Kr // 16)
    g = 1                                  # per-chunk load/consume schedule
    tm, tn = -(-Mg // 32), -(-Ng // 32)
    mi, ni = 2, 2
    if tm * tn > 256 and chunks >= 32:                 # wide tiles: one round of workgroups
        mi, ni = (2, 4) if tn >= tm else (4, 2)
    elif tm * tn <= 128 and Mg > 16:                   # 16-row tiles: twice the workgroups
        mi, ni = 1, 2
    if mode == 0 and (mi, ni) == (4, 2) and kind == "fwd" and tn * -(-Mg // 48) <= 256:
        mi, ni = 3, 2                                  # 3B-row forward: 208 tiles of 48x32 instead of 156 of 64x32
    if mode == 2:
        # weight gradients: 32x48 / 48x32 wherever that covers the output in one round of <= 256 workgroups
        if (mi, ni) == (2, 4) and tm * -(-Ng // 48) <= 256:
            mi, ni = 2, 3
        elif (mi, ni) == (4, 2) and tn * -(-Mg // 48) <= 256:
            mi, ni = 3, 2
        elif (mi, ni) == (2, 2) and tm * tn > 256:
            if tn >= tm and tm * -(-Ng // 48) <= 256:
                mi, ni = 2, 3
            elif tn < tm and tn * -(-Mg // 48) <= 256:
                mi, ni = 3, 2
    b = lambda v: "true" if v else "false"
    # weight gradients over >= 768 rows on 32x48 / 48x32 tiles: the LDS-DMA instantiations (gemm16_dw_dma)
    dma = mode == 2 and xv and Kr >= 768 and (mi, ni) in ((2, 3), (3, 2))
    if kind in ("dwh", "dwhf", "dwhs"):
        # last three arguments: ones column with a row offset (WGAN-GP's stacked weight gradient only); folded
        # head; LDS-DMA
        return "gemm16_dw_head_kernel<false, %d, %s, %d, %d, %s, %s, %s>" % (
            g, b(xv), mi, ni, b(kind == "dwhs"), b(kind == "dwhf"), b(dma))
    if kind in ("dxh", "dxhf"):
        assert vec and xv and (mi, ni) in ((2, 2), (1, 2))
        return "gemm16_dx_head_kernel<%d, %d, %d, %s>" % (g, mi, ni, b(kind == "dxhf"))
    if kind == "fwdg":
        assert vec and (mi, ni) in ((2, 2), (1, 2))
        return "gemm16_fwd_gather_kernel<true, %d, %d, %d>" % (g, mi, ni)
    if kind == "dwp":
        assert xv and (mi, ni) != (1, 2)
        # last two arguments (round 6): which of the pair resolves ring slots on its operands -- the generator's second
        # GEMM reads its X from the noise ring (extra is None), the VAE's pairs (extra = the second layer's shape) neither
        return "gemm16_dw_pair_kernel<%d, true, %d, %d, %s, false, %s>" % (g, mi, ni, b(dma), b(extra is None))
    # (last argument, round 6: the instantiation that resolves ring slots on its operands -- false for every launch this
    # function names, only the generator's first layer reads through one and it rides in the gather / pair kernels)
    return "gemm16_kernel<%d, %s, %d, %d, %s, %d, %d, %s, false>" % (mode, b(vec), nw, g, b(xv), mi, ni, b(dma))


def clock_probe():
    """Effective shader clock (MHz) while every CU runs a dependent fp32-MFMA chain."""
    from generative_models_amd import _lib, ops
    out = torch.zeros(2, dtype=torch.int64, device="cuda")
    sink = torch.zeros(1, device="cuda")
    for _ in range(3):
        _lib.call("gm_clock_probe", ops.stream_ptr(), 4000, out.data_ptr(), sink.data_ptr())
    torch.cuda.synchronize()
    cyc, wall = [int(x) for x in out.cpu()]
    return cyc / max(wall, 1) * 100.0, cyc / 4000.0


def _holder(N, K, dev):
    """Stand-in for engine._Linear: parameter / gradient / Adam-moment views of one layer."""
    from types import SimpleNamespace
    z = lambda *s: torch.zeros(*s, device=dev)
    return SimpleNamespace(W=torch.randn(N, K, device=dev) / K ** 0.5, b=z(N), gW=z(N, K), gb=z(N),
                           mW=z(N * K), vW=z(N * K), mb=z(N), vb=z(N))


def time_kernels_isolated(B, reps=100, fused_head=True, batch_gen=True, group_head=True,
                          ride_gather=True, pair_dw=True, ride_head_dx=True, fused_adam=True, shapes=None,
                          fold_head=False):
    """HIP-event timing (on the launch stream) of each GEMM launch shape of the step, run back to
    back `reps` times.  Returns {kernel instantiation name: (total_us_per_step,
    total_flop_per_step, n_launches_per_step)}."""
    from generative_models_amd import ops
    dev = "cuda"
    out = {}
    st = ops.stream_ptr()
    data = idx = xr = None
    for shape in (shapes or gemm_shapes(B, fused_head, batch_gen, group_head, ride_gather,
                                        pair_dw, ride_head_dx, fold_head)):
        kind, M, K, N = shape[:4]
        extra = shape[4] if len(shape) > 4 else None
        x = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) / K ** 0.5
        dA = torch.randn(M, N, device=dev)
        y = torch.empty(M, N, device=dev)
        dX = torch.empty(M, K, device=dev)
        dW = torch.empty(N, K, device=dev)
        db = torch.empty(N, device=dev)
        b = torch.zeros(N, device=dev)
        flop = 2.0 * M * K * N
        if kind == "fwd":
            fn = lambda: ops.linear_fwd(x, W, b, y, "relu", stream=st)
        elif kind == "fwds":                    # sigmoid output layer + reconstruction loss epilogue
            tgt = (torch.rand(M, N, device=dev) < 0.13).float()
            part2 = torch.zeros(M, (-(-N // 32) + 3) // 4 * 4, device=dev)
            fn = lambda: ops.linear_fwd_sqerr(x, W, b, y, tgt, dA, part2, stream=st)
        elif kind == "fwdz":                    # reparameterisation + decoder layer 1 (K = z_dim) in one launch
            from generative_models_amd import ops_fused
            mlz, epz = torch.randn(M, 2 * K, device=dev) * 0.5, torch.randn(M * K, device=dev)
            zz_, pk = torch.empty(M, K, device=dev), torch.empty((M * K + 255) // 256, device=dev)
            fn = lambda: ops_fused.vae_reparam_fwd(mlz, epz, zz_, pk, M, K, W, b, y, "relu", stream=st)
        elif kind == "bmid":                    # dz, d[mu | log_var], dHe: K = z_dim, N = hidden
            from generative_models_amd import ops_fused
            mlz, epz = torch.randn(M, 2 * K, device=dev) * 0.5, torch.randn(M * K, device=dev)
            dmlz, Wml_ = torch.empty(M, 2 * K, device=dev), torch.randn(2 * K, N, device=dev) / N ** 0.5
            Hez, dHez = torch.relu(torch.randn(M, N, device=dev)), torch.empty(M, N, device=dev)
            fn = lambda: ops_fused.vae_bwd_mid(dA, W, mlz, epz, dmlz, Wml_, Hez, dHez, M, stream=st)
            flop = 2.0 * M * K * N + 2.0 * M * 2 * K * N
        elif kind == "dxr":                     # dX (layer [N, K]: K = z_dim) + reparameterisation backward
            mlz, epz, dmlz = torch.randn(M, 2 * K, device=dev), torch.randn(M * K, device=dev), torch.empty(M, 2 * K, device=dev)
            fn = lambda: ops.linear_bwd_dx_reparam(dA, W, dX, mlz, epz, dmlz, stream=st)
        elif kind in ("fwdp", "dwhf", "dxhf"):
            # the folded head's launches on a consistent state: forward first (partial dots, snapshot)
            L1, L2 = _holder(N, K, dev), _holder(1, N, dev)
            fold = ops.HeadFold(M, N, dev)
            Hh, lo = torch.empty(M, N, device=dev), torch.zeros(1, device=dev)
            ops.linear_fwd_headpart(x, L1.W, L1.b, Hh, "relu", L2, fold, stream=st)
            sched = torch.from_numpy(ops.adam_schedule(2e-4, 4)).to(dev)
            ad = dict(sched=sched, sched_slot=ops.slot(0, 0, 1, 0, 1), clamp=0.0) if fused_adam else None
            if kind == "fwdp":
                fn = lambda: ops.linear_fwd_headpart(x, L1.W, L1.b, Hh, "relu", L2, fold, stream=st)
            elif kind == "dwhf":
                fa = fold.args("ns", "sigmoid")
                head = dict(H=Hh, lin=L2, loss_out=lo, loss_slot=ops.NO_SLOT, inv_b=2.0 / M, B=M // 2, adam=ad)
                # (Adam steps L1 / L2 between timed launches; the partial dots stay those of the first forward)
                fn = lambda: ops.linear_bwd_dw_adam_head_fold(Hh, x, L1, ad, head, fa, stream=st)
            else:
                fa = fold.args("ns", "sigmoid")
                head = dict(H=Hh, lin=L2, loss_out=lo, loss_slot=ops.NO_SLOT, inv_b=1.0 / M, B=M, gen_mode=True)
                fn = lambda: ops.linear_bwd_dx_head_fold(Hh, L1.W, dX, head, fa, below=x, epi="sigmoid", stream=st)
        elif kind == "fwdg":
            if data is None:
                data = (torch.rand(N_TRAIN, IMG, device=dev) > 0.5).float()
                idx = torch.randint(0, N_TRAIN, (B,), device=dev)
                xr = torch.empty(B, IMG, device=dev)
            fn = lambda: ops.linear_fwd_gather(x, W, b, y, "relu", data, idx, xr, stream=st)
        elif kind == "dxh":
            L2 = _holder(1, N, dev)
            Hh = torch.relu(torch.randn(M, N, device=dev))
            dS, rl, lo = torch.randn(M, device=dev) / M, torch.rand(M, device=dev), torch.zeros(1, device=dev)
            head = dict(H=Hh, dS=dS, lin=L2, rowloss=rl, loss_out=lo, loss_slot=ops.NO_SLOT,
                        inv_b=1.0 / M, B=M, gen_mode=True)
            fn = lambda: ops.linear_bwd_dx_head(dA, W, dX, head, below=x, epi="relu", stream=st)
        elif kind == "dwp":
            N2, K2 = extra if extra is not None else (K, Z)       # second GEMM: dW of a [N2, K2] layer
            L2, L1 = _holder(N, K, dev), _holder(N2, K2, dev)
            dH2, zz = torch.randn(M, N2, device=dev), torch.randn(M, K2, device=dev)
            sched = torch.from_numpy(ops.adam_schedule(2e-4, 4)).to(dev)
            ad = dict(sched=sched, sched_slot=ops.slot(0, 0, 1, 0, 1), clamp=0.0) if fused_adam else None
            # the generator's pair reads the second GEMM's X through a ring slot (the noise ring), as the step does
            zctr = torch.zeros(1, dtype=torch.int64, device=dev)         # (captured by the lambda: stays alive)
            zslot = {} if extra is not None else dict(x_slot=ops.slot(zctr.data_ptr(), 1, 0, 1, 0))
            fn = lambda: ops.linear_bwd_dw_adam_pair(dict(dA=dA, X=x, lin=L2, adam=ad),
                                                     dict(dA=dH2, X=zz, lin=L1, adam=ad, keep=zctr, **zslot), stream=st)
            flop += 2.0 * M * N2 * K2
        elif kind == "dx":
            fn = lambda: ops.linear_bwd_dx(dA, W, dX, below=x, epi="relu", stream=st)
        elif kind in ("dwh", "dwhs"):
            L1, L2 = _holder(N, K, dev), _holder(1, N, dev)
            Hh = torch.relu(torch.randn(M, N, device=dev))
            dS, rl, lo = torch.randn(M, device=dev) / M, torch.rand(M, device=dev), torch.zeros(1, device=dev)
            sched = torch.from_numpy(ops.adam_schedule(2e-4, 4)).to(dev)
            ad = dict(sched=sched, sched_slot=ops.slot(0, 0, 1, 0, 1), clamp=0.0) if fused_adam else None
            if kind == "dwhs":                  # stacked reduction: rows [0, M/3) do not reach db; head over 2M/3 rows
                Hh, dS, rl = Hh[:2 * M // 3], dS[:2 * M // 3], rl[:2 * M // 3]
                head = dict(H=Hh, dS=dS, lin=L2, rowloss=rl, loss_out=lo, loss_slot=ops.NO_SLOT,
                            inv_b=3.0 / M, B=M // 3, adam=ad, gw2_add=torch.zeros(N, device=dev))
                fn = lambda: ops.linear_bwd_dw_adam_head(dA, x, L1, ad, head, ones_from=M // 3, stream=st)
            else:
                head = dict(H=Hh, dS=dS, lin=L2, rowloss=rl, loss_out=lo, loss_slot=ops.NO_SLOT,
                            inv_b=2.0 / M, B=M // 2, adam=ad)
                fn = lambda: ops.linear_bwd_dw_adam_head(dA, x, L1, ad, head, stream=st)
        else:                                   # as in the step: Adam in the gradient epilogue
            L1 = _holder(N, K, dev)
            sched = torch.from_numpy(ops.adam_schedule(2e-4, 4)).to(dev)
            ad = dict(sched=sched, sched_slot=ops.slot(0, 0, 1, 0, 1), clamp=0.0)
            fn = (lambda: ops.linear_bwd_dw_adam(dA, x, L1, ad, stream=st)) if fused_adam else \
                (lambda: ops.linear_bwd_dw(dA, x, L1.gW, L1.gb, stream=st))
        fn()
        torch.cuda.synchronize()
        # capture `reps` back-to-back launches into one hipGraph so that the host-side launch
        # cost (python+ctypes, ~8 us) is not what the events measure
        def body(gst, fn=fn, kind=kind):
            nonlocal st
            st = gst
            for _ in range(reps):
                fn()
        g = ops.Graph().capture(body)
        st = ops.stream_ptr()
        g.launch(); g.launch()
        e0, e1 = ops.Event(), ops.Event()
        e0.record(st)
        g.launch()
        e1.record(st)
        e1.sync()
        us = e0.elapsed_ms(e1) * 1e3 / reps
        log('  %-4s M=%4d K=%4d N=%4d : %7.2f us  %6.2f TFLOP/s' % (kind, M, K, N, us, flop / us / 1e6))
        name = gemm_variant(kind, M, K, N, extra)
        t, f, n = out.get(name, (0.0, 0.0, 0))
        out[name] = (t + us, f + flop, n + 1)
    return out


def log(msg):
    if os.environ.get("GM_BENCH_VERBOSE"):
        print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def _probe_threads(fn_step, cands=(8, 16, 32)):
    """Fastest OpenMP thread count for the CPU oracle in a short probe (hundreds of threads on
    256x400 GEMMs only add barrier time); the count used is reported as `cores`."""
    ncpu = os.cpu_count() or 1
    best = None
    for cand in [c for c in cands if c <= ncpu] or [min(ncpu, 4)]:
        torch.set_num_threads(cand)
        fn_step(3)
        t0 = time.perf_counter()
        fn_step(8)
        ps = (time.perf_counter() - t0) / 8
        log("cpu probe %.1f ms/step on %d threads" % (ps * 1e3, cand))
        if best is None or ps < best[0]:
            best = (ps, cand)
    torch.set_num_threads(best[1])
    return best


def cpu_baseline_gan(variant="ns", B=B_PER_GPU, seconds_target=12.0, compute_only=False, cores=None):
    """Oracle port (CPU restatement of the reference trainer, oracle/port.py, pinned bit-exact to
    the unmodified reference) timed on this host's cores on the same synthetic data.
    as-written: DataLoader reshuffle + collate per step included (the north-star "reference CPU
    Trainer"); compute_only: process_batch returns one pre-fetched batch (separates
    the data path from the math)."""
    from oracle import port
    ds = synthetic_dataset()
    loader = torch.utils.data.DataLoader(ds, batch_size=B, shuffle=True)
    model = port.build(variant, IMG, HID, Z)
    tr = port.GANPort(variant, model, loader)
    kw = {"D_steps": 1} if variant == "wgp" else {}
    if compute_only:
        fixed = tr.process_batch()
        tr.process_batch = lambda: fixed
    step = lambda n: tr.train(1, max_steps=n, **kw)
    if cores is None:
        per_step, cores = _probe_threads(step)
    else:
        torch.set_num_threads(cores)
        step(2)
        t0 = time.perf_counter(); step(4); per_step = (time.perf_counter() - t0) / 4
    steps_per_epoch = int(np.ceil(len(loader)))
    total, done, dt = int(max(8, min(2000, seconds_target / per_step))), 0, 0.0
    while done < total:
        n = min(steps_per_epoch, total - done)
        t0 = time.perf_counter()
        step(n)
        dt += time.perf_counter() - t0
        done += n
    return {"value": done * B / dt, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d %s bs=%d D+G steps of oracle/port.py (torch %s CPU, %d threads), %s, %.1f s"
                      % (done, variant, B, torch.__version__, cores,
                         "compute only (fixed pre-fetched batch)" if compute_only
                         else "as-written incl. DataLoader reshuffle", dt) +
                      "; kind 'port' = the oracle's restatement, pinned bit for bit to the unmodified reference "
                      "(tests/test_oracle_pin.py), used where the reference is not mounted (the GPU box); the two timed "
                      "side by side on one host: profiles/r06_cpu_reference_vs_port.json"}


def cpu_baseline_reference(B=B_PER_GPU, seconds_target=12.0, cores=None):
    """SURVEY.md 8(d) as written: the UNMODIFIED reference's NSGANTrainer.train (/root/reference/src/ns_gan.py:94-170,
    loaded by oracle/ref_harness.py) timed on this host's cores on the same synthetic data -- only where the reference is
    mounted (GM_REFERENCE_ROOT, default /root/reference: the build container; never on the GPU box, which falls back
    to the port).  Returns None when it is not."""
    from oracle import ref_harness
    if not ref_harness.available():
        return None
    mod = ref_harness.load("ns_gan")
    ds = synthetic_dataset()

    class Capped(torch.utils.data.DataLoader):
        cap = 1

        def __len__(self):
            return self.cap
    mk = lambda: torch.utils.data.DataLoader(ds, batch_size=B, shuffle=True)
    train_iter = Capped(ds, batch_size=B, shuffle=True)
    torch.manual_seed(1234)
    model = mod.NSGAN(image_size=IMG, hidden_dim=HID, z_dim=Z)
    tr = mod.NSGANTrainer(model, train_iter, mk(), mk(), viz=False)

    def step(n):
        train_iter.cap = n
        with ref_harness.quiet():
            tr.train(num_epochs=1)
    if cores is None:
        per_step, cores = _probe_threads(step)
    else:
        torch.set_num_threads(cores)
        step(2)
        t0 = time.perf_counter(); step(4); per_step = (time.perf_counter() - t0) / 4
    total = int(max(8, min(2000, seconds_target / per_step)))
    t0 = time.perf_counter()
    step(total)
    dt = time.perf_counter() - t0
    return {"value": total * B / dt, "unit": "images/sec", "cores": cores, "kind": "reference",
            "sample": "%d NSGAN bs=%d D+G steps of the unmodified reference's NSGANTrainer.train (ns_gan.py:94-170 via "
                      "oracle/ref_harness.py, torch %s CPU, %d threads), as-written incl. the per-step DataLoader "
                      "reshuffle, %.1f s" % (total, B, torch.__version__, cores, dt)}


def cpu_baseline_vae(B=512, seconds_target=5.0, cores=16):
    from oracle import port
    torch.set_num_threads(min(cores, os.cpu_count() or 1))
    ds = synthetic_dataset()
    mk = lambda: torch.utils.data.DataLoader(ds, batch_size=B, shuffle=True)
    tr = port.VAEPort(port.VAEModel(IMG, HID, Z), mk(), mk(), mk())
    tr.train(1, max_steps=3, do_eval=False)
    t0 = time.perf_counter(); tr.train(1, max_steps=4, do_eval=False); ps = (time.perf_counter() - t0) / 4
    n = int(max(8, min(98, seconds_target / ps)))
    t0 = time.perf_counter()
    tr.train(1, max_steps=n, do_eval=False)
    dt = time.perf_counter() - t0
    return {"value": n * B / dt, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d VAE bs=%d training batches of oracle/port.py (as-written DataLoader "
                      "iteration), %.1f s" % (n, B, dt)}


# ------------------------------------------------------------------------------------------------
def fence(world):
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
        torch.cuda.synchronize()


def timed_reps(run_rep, reps, K, world, dev):
    """run_rep(r) enqueues repetition r (K steps).  Returns seconds per repetition (max over
    ranks), barrier + synchronize on both sides of each."""
    out = []
    for r in range(reps):
        fence(world)
        t0 = time.perf_counter()
        run_rep(r)
        fence(world)
        dt = time.perf_counter() - t0
        if world > 1:
            gloo = torch.distributed.get_backend() == "gloo"
            t = torch.tensor([dt], device="cpu" if gloo else dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = t.item()
        out.append(dt)
    return out


SUSTAINED_MAX_STEPS = 120000      # room in the schedule / loss buffers for the sustained window below


def bench_gan(variant, B_global, W, K, reps, dev, world=1, rank=0, use_graph=True, force_dp=False,
              lrs=(2e-4, 2e-4), D_steps=1, solo=False, long_steps=0, sustained_s=0.0):
    """W warm-up + reps x K timed iterations of the fused engine.  Returns (engine, [seconds]).
    solo: a 1-rank engine timed on this rank alone inside a multi-rank job (no barrier / max)."""
    import importlib
    from generative_models_amd import engine as gm_engine
    mod, cls = {"ns": ("ns_gan", "NSGAN"), "ls": ("ls_gan", "LSGAN"), "wgp": ("w_gp_gan", "WGPGAN"),
                "dra": ("dra_gan", "DRAGAN")}[variant]
    m = importlib.import_module(mod)
    ds = synthetic_dataset()
    loader = torch.utils.data.DataLoader(ds, batch_size=B_global, shuffle=True)
    torch.manual_seed(1234)
    model = getattr(m, cls)(image_size=IMG, hidden_dim=HID, z_dim=Z)
    trainer = getattr(m, cls + "Trainer")(model, loader, None, None, viz=False)
    data = ds.tensors[0].reshape(N_TRAIN, -1).to(dev).contiguous()          # resident in HBM
    eng = gm_engine.GANEngine(variant, trainer.model, data, B_global, dev, use_graph=use_graph,
                              world_size=world, rank=rank, force_dp=force_dp)
    eng.configure(W + reps * K + long_steps + (SUSTAINED_MAX_STEPS if sustained_s else 0), lrs[0], lrs[1], D_steps)
    eng.run(W, it_start=0)
    marks = []

    def rep(r):
        marks.append(time.perf_counter())
        eng.run(K, it_start=W + r * K)
        marks.append(time.perf_counter())
    secs = timed_reps(rep, reps, K, 1 if solo else world, dev)
    if eng._trace:                                   # GM_TRACE_RUN=1: host timeline of each repetition
        for r in range(reps):
            t0, t1 = marks[2 * r], marks[2 * r + 1]
            ev = [(k, a, round((t - t0) * 1e6)) for k, a, t in eng._trace if t0 <= t <= t1 + 1e-3]
            print("[trace rep %d] run() returned at %d us, total %d us: %s"
                  % (r, (t1 - t0) * 1e6, secs[r] * 1e6, ev), file=sys.stderr, flush=True)
    eng.steady_us_per_step = None
    if long_steps:
        # one long region right behind the timed ones: the steady-state step, so that the fixed cost
        # of a K-step run() (cold start of the host draws, graph boundaries, final sync) can be reported
        ls = timed_reps(lambda r: eng.run(long_steps, it_start=W + reps * K), 1, long_steps,
                        1 if solo else world, dev)
        eng.steady_us_per_step = ls[0] / long_steps * 1e6
    eng.sustained = None
    n_sus = 0
    if sustained_s and eng.steady_us_per_step:
        # ONE uninterrupted window of >= sustained_s seconds of the same step (host draws inside, as everywhere): long
        # enough for an outside observer sampling GPU activity every few seconds (the driver's gpu_busy) to see it
        n_sus = int(min(SUSTAINED_MAX_STEPS, sustained_s * 1e6 / eng.steady_us_per_step + 1))
        su = timed_reps(lambda r: eng.run(n_sus, it_start=W + reps * K + long_steps), 1, n_sus,
                        1 if solo else world, dev)
        eng.sustained = {"seconds": su[0], "steps": n_sus, "us_per_step": su[0] / n_sus * 1e6,
                         "img_s": n_sus * B_global / su[0]}
    G, D = eng.losses(W, W + reps * K + long_steps + n_sus)
    assert np.isfinite(G).all() and np.isfinite(D).all(), "non-finite losses"
    return eng, secs


def bench_vae(B, dev, with_eval, warm_epochs=1, epochs=5, n_val=10000):
    """Full epochs of vae.py's train loop on the fused engine: 97 batches of 512 + the ragged 336
    (50 000 mod 512); with_eval adds the reference's per-epoch validation pass (vae.py:174-175)."""
    import vae
    from generative_models_amd.engine import VAEEngine
    from generative_models_amd.trainers import _epoch_order
    ds = synthetic_dataset()
    torch.manual_seed(3436)
    vds = torch.utils.data.TensorDataset(torch.bernoulli(torch.full((n_val, 1, 28, 28), 0.1307)),
                                         torch.zeros(n_val, dtype=torch.int64))
    tl = torch.utils.data.DataLoader(ds, batch_size=B, shuffle=True)
    vl = torch.utils.data.DataLoader(vds, batch_size=B, shuffle=True)
    torch.manual_seed(1234)
    tr = vae.VAETrainer(vae.VAE(IMG, HID, Z), tl, vl, vl)
    eng = VAEEngine(tr.model, dev)
    steps = len(tl)
    eng.configure(B, (warm_epochs + epochs) * steps, 1e-3, 1e-5)
    tdata = ds.tensors[0].reshape(N_TRAIN, -1).to(dev).contiguous()
    vdata = vds.tensors[0].reshape(n_val, -1).to(dev).contiguous()
    eng.alloc_val(len(vl))
    # every epoch is timed on its own (synchronize on both sides: the reference reads its losses at every
    # epoch end anyway) and the MEDIAN epoch is reported -- one 5 ms host hiccup in a 10 ms epoch used to
    # move a single 3-epoch timing by 15 %
    per_epoch = []
    for e in range(warm_epochs + epochs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run_pass(tdata, _epoch_order(tl), True, e * steps)
        if with_eval:
            eng.run_pass(vdata, _epoch_order(vl), False, 0)
            eng.vrecon[:len(vl)].cpu()                # the reference reads the validation loss every epoch
        torch.cuda.synchronize()
        if e >= warm_epochs:
            per_epoch.append(time.perf_counter() - t0)
    dt = float(np.median(per_epoch))
    assert np.isfinite(eng.recon.cpu().numpy()).all()
    return N_TRAIN / dt, dt / steps * 1e3, epochs * steps


BYTES_PER_IMAGE_B256 = 82944


def mfma_busy_frac(kernel, launch_us, mhz):
    """MFMA-pipe busy fraction of the dominant kernel from the COMMITTED SQ PMC pass
    (profiles/r02_nsgan_b256_sq_pmc.json: SQ_VALU_MFMA_BUSY_CYCLES per dispatch, summed over the 1024
    SIMDs) over this run's launch duration; None when the pass does not list the kernel."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND + "_nsgan_b256_sq_pmc.json")))
        rows = [v["SQ_VALU_MFMA_BUSY_CYCLES"] for k, v in pmc.items() if k.split("|")[0] == kernel]
        if not rows:
            return None
        return (sum(rows) / len(rows) / 1024.0) / (launch_us * mhz)
    except Exception:                                # noqa: BLE001
        return None


def dominant_gemm_roofline(shapes, B, pmc_tag=None, reps=50):
    """Roofline entry of a configuration: EVERY GEMM launch shape of its step is timed in isolation
    (HIP events over graph-captured back-to-back launches, as for the headline) and the kernel
    instantiation with the largest time share of the step is reported -- not a fixed forward shape.
    traffic: that kernel's bytes per dispatch from the committed PMC pass of this configuration."""
    kt = time_kernels_isolated(B, reps=reps, shapes=shapes)
    name = max(kt, key=lambda k: kt[k][0])
    us, flop, n = kt[name]
    ach = flop / (us * 1e-6) / 1e12
    e = {"bound": "mfma", "kernel": name, "launches_per_step": n, "avg_launch_us": us / n,
         "us_per_step": us, "shapes": ["%s %dx%dx%d" % tuple(sh[:4]) for sh in shapes
                                       if gemm_variant(*sh[:5]) == name],
         "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_MFMA_TFLOPS,
         "per_kernel_us_per_step": {k: round(v[0], 2) for k, v in kt.items()}}
    e["traffic"], e["traffic_source"] = pmc_traffic(name, pmc_tag)
    e["hbm_gbps"] = (e["traffic"] / (e["avg_launch_us"] * 1e-6) / 1e9) if e["traffic"] else None
    if pmc_tag is not None:
        e.update(profile_check(name, e["avg_launch_us"], pmc_tag))
    return e


def profile_check(kernel, live_us, tag):
    """{"profile_kernel_us", "profile_source", "stale_profile"}: the committed rocprofv3 average duration of `kernel`
    (profiles/<round>_<tag>_kernel_stats.csv) beside this run's HIP-event figure; stale when they differ by > 10 % or
    the committed summary does not list the kernel at all."""
    import csv
    src = "%s_%s_kernel_stats.csv" % (PROFILE_ROUND, tag)
    try:
        rows = {r["kernel"]: float(r["avg_us"]) for r in csv.DictReader(open(os.path.join(ROOT, "profiles", src)))}
    except Exception:                                # noqa: BLE001
        return {"profile_kernel_us": None, "profile_source": None, "stale_profile": True}
    us = rows.get(kernel)
    return {"profile_kernel_us": us, "profile_source": "profiles/" + src,
            "stale_profile": us is None or abs(us - live_us) > 0.10 * live_us}


def pmc_traffic(kernel, tag):
    """Bytes per dispatch (2 x FETCH_SIZE gfx950 correction + WRITE_SIZE) of `kernel` from the committed
    PMC pass profiles/<round>_<tag>_pmc_traffic.json (PMC counters cannot be read from inside this
    process); (None, None) when the pass does not list the kernel."""
    if tag is None:
        return None, None
    src = "%s_%s_pmc_traffic.json" % (PROFILE_ROUND, tag)
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", src)))
        rows = [v for k, v in pmc.items() if k.split("|")[0] == kernel]
        if not rows:
            return None, None
        nd = sum(v["dispatches"] for v in rows)
        return sum((v["read_bytes"] + v["write_bytes"]) * v["dispatches"] for v in rows) / nd, "profiles/" + src
    except Exception:                                # noqa: BLE001
        return None, None


def other_configs(dev, steps, warmup, reps, cpu=True, only=None):
    """BASELINE.json configs 3/4/5 on one MI355X, >= 200 timed steps each (+ DRAGAN,
    the variant whose per-step host draw is the largest: B x 784 uniforms).  only: one of
    wgp_b256 / wgp_b256_d5 / ns_b1024 / ls_b1024 / dra_b256 / vae_b512 (profiling runs)."""
    K, W = max(steps, 200), max(warmup, 20)
    out = []
    want = lambda tag: only is None or only == tag
    # every CPU leg runs AFTER the last GPU leg: a 16-thread CPU baseline leaves the host (16-core quota on the GPU
    # boxes) busy for a while, and the GPU legs behind it paid for that -- bs=1024 read 163-167 us with the CPU legs
    # in between against 145-146 us alone or with --no-cpu-baseline (same box, round 4)
    deferred = []

    def gan(name, variant, B, lrs, flop_per_image, cpu_variant):
        eng, secs = bench_gan(variant, B, W, K, reps, dev, lrs=lrs)
        dt = float(np.median(secs))
        e = {"workload": name, "img_s": K * B / dt, "ms_per_step": dt / K * 1e3, "steps": K,
             "reps_ms_per_step": [round(x / K * 1e3, 5) for x in secs],
             "step_mfma_frac": K * B / dt * flop_per_image / (PEAK_FP32_MFMA_TFLOPS * 1e12),
             "roofline": dominant_gemm_roofline(
                 gemm_shapes_wgp(B, eng._fold_head_G()) if variant == "wgp" else
                 gemm_shapes(B, fold_head=eng._fold_head()), B,
                 pmc_tag={"wgp": "wgp_b256", "ns": "ns_b1024", "ls": "ns_b1024"}.get(variant))}   # (LSGAN: NSGAN's launches)
        e["roofline"]["step_frac"] = e["step_mfma_frac"]
        if cpu:
            deferred.append((e, lambda: cpu_baseline_gan(cpu_variant, B, seconds_target=4.0, cores=16)))
        log("%s: %.0f img/s" % (name, e["img_s"]))
        del eng
        out.append(e)

    if want("wgp_b256"):
        gan("WGAN-GP MNIST bs=256 D_steps=1 (BASELINE.json configs[2]; w_gp_gan.py __main__)", "wgp", 256,
            (1e-4, 1e-4), 8_836_000, "wgp")
    if want("wgp_b256_d5"):
        # SURVEY.md 8d config 3 also asks for the reference's D_steps=5 default (w_gp_gan.py:96): images =
        # real images consumed by the critic steps = 5 B per D+G iteration
        eng, secs = bench_gan("wgp", 256, W, K, reps, dev, lrs=(1e-4, 1e-4), D_steps=5)
        dt = float(np.median(secs))
        out.append({"workload": "WGAN-GP MNIST bs=256 D_steps=5 (train() default, w_gp_gan.py:96); images = critic-step images",
                    "img_s": K * 5 * 256 / dt, "ms_per_step": dt / K * 1e3, "steps": K,
                    "reps_ms_per_step": [round(x / K * 1e3, 5) for x in secs]})
        log("WGAN-GP bs=256 D_steps=5: %.0f img/s" % (K * 5 * 256 / dt))
        del eng
    if want("ns_b1024"):
        gan("NSGAN MNIST bs=1024, 1 GPU (BASELINE.json configs[4], single-GPU leg)", "ns", 1024,
            (2e-4, 2e-4), FLOP_PER_IMAGE, "ns")
    if want("ls_b1024"):
        gan("LSGAN MNIST bs=1024, 1 GPU (BASELINE.json configs[4], single-GPU leg)", "ls", 1024,
            (1e-4, 1e-4), FLOP_PER_IMAGE, "ls")
    if want("dra_b256"):
        # not a BASELINE.json config: the variant with the heaviest host protocol (dra_gan.py:200-205
        # draws B x 784 uniforms per critic step on the CPU generator) -- VERDICT r1 item 9
        eng, secs = bench_gan("dra", 256, W, K, reps, dev, lrs=(1e-4, 1e-4))
        dt = float(np.median(secs))
        out.append({"workload": "DRAGAN MNIST bs=256 D_steps=1 (dra_gan.py; host draws 256 x 784 uniforms per step)",
                    "img_s": K * 256 / dt, "ms_per_step": dt / K * 1e3, "steps": K,
                    "reps_ms_per_step": [round(x / K * 1e3, 5) for x in secs],
                    "step_mfma_frac": K * 256 / dt * VARIANT_FLOP["dra"][0] / (PEAK_FP32_MFMA_TFLOPS * 1e12),
                    "roofline": {"bound": "mfma", "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                 "flop_per_image": VARIANT_FLOP["dra"][0], "flop_derivation": VARIANT_FLOP["dra"][1],
                                 "step_frac": K * 256 / dt * VARIANT_FLOP["dra"][0] / (PEAK_FP32_MFMA_TFLOPS * 1e12),
                                 "profile": "profiles/%s_dra_b256_summary.md" % PROFILE_ROUND}})
        log("DRAGAN bs=256: %.0f img/s" % (K * 256 / dt))
        del eng
    for v in ("info", "be"):
        if want(v + "_b256"):
            out.append(bench_variant_trainer(v))
    for with_eval in ((False, True) if want("vae_b512") else ()):
        img_s, ms, n = bench_vae(512, dev, with_eval)
        e = {"workload": "VAE MNIST bs=512 full epochs incl. the ragged 336 batch, %s "
                         "(BASELINE.json configs[3])" % ("train + per-epoch 10k-image validation pass"
                                                         if with_eval else "train loop only"),
             "img_s": img_s, "ms_per_step": ms, "steps": n,
             "step_mfma_frac": img_s * 3_280_000 / (PEAK_FP32_MFMA_TFLOPS * 1e12),
             "roofline": dominant_gemm_roofline(gemm_shapes_vae(512), 512, pmc_tag="vae_b512")}
        e["roofline"]["step_frac"] = e["step_mfma_frac"]
        if cpu and not with_eval:
            deferred.append((e, lambda: cpu_baseline_vae(512)))
        log("%s: %.0f img/s" % (e["workload"], img_s))
        out.append(e)
    for e, fn in deferred:
        e["cpu_baseline"] = fn()
    return out


# Algorithmic FLOP per real image of one D+G iteration at D_steps = 1 for the variants outside BASELINE.json (2 m n k per
# GEMM; U = one 784 x 400 contraction = 627 200 FLOP per row; the reference's wasted backprops -- G's gradients during the
# critic step, D's during the generator step -- are not counted)
VARIANT_FLOP = {
    "dra": (8_836_000, "NSGAN's 10 U + small layers (6 326 400) + the penalty path's D(x_hat) forward, grad wrt x_hat and "
                       "the two second-backward GEMMs = 4 U: as WGAN-GP (SURVEY.md 8d), dra_gan.py:198-223"),
    "info": (10_249_600, "critic step 5 U + 36 800, generator step 5 U + 65 600 (G's first layer is 40 wide: 32 000 per "
                         "pass), MI step (info_gan.py:269-304) G forward, Q forward, Q dW1, Q dX, dH through G's output "
                         "layer, G dW2 = 6 U + 112 000 (G's first layer forward and dW, Q's 400 x 20 head forward / dW / "
                         "dX): 16 U + 214 400"),
    "be": (11_337_600, "the critic is an autoencoder 784-400-784 (be_gan.py:63-76): critic step G 1 U + D forward on "
                       "[x ; G(z)] 4 U + decoder dW 2 U + dH 2 U + encoder dW 2 U = 11 U + 16 000; generator step G 1 U + "
                       "D forward 2 U + dX through decoder and encoder 2 U + dH through G 1 U + G dW2 1 U = 7 U + 32 000: "
                       "18 U + 48 000"),
}


def bench_variant_trainer(v, epochs=2):
    """InfoGAN / BEGAN at bs=256 through their drop-in Trainer.train (default arguments), warm: us per D+G iteration,
    images/s and the step's fraction of the FP32-MFMA roofline from VARIANT_FLOP."""
    import importlib
    mod_name, cls = {"info": ("info_gan", "InfoGAN"), "be": ("be_gan", "BEGAN")}[v]
    mod = importlib.import_module(mod_name)
    ds = synthetic_dataset()
    torch.manual_seed(1234)
    kw = dict(image_size=IMG, hidden_dim=HID, z_dim=Z)
    if v == "info":
        kw.update(disc_dim=10, cont_dim=10)
    model = getattr(mod, cls)(**kw)
    tr = getattr(mod, cls + "Trainer")(model, torch.utils.data.DataLoader(ds, batch_size=B_PER_GPU, shuffle=True), None, None, viz=False)
    steps = len(tr.train_iter)
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.train(epochs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert np.isfinite(tr.Glosses).all() and tr._engine is not None
    img_s = epochs * steps * B_PER_GPU / dt
    flop, why = VARIANT_FLOP[v]
    frac = img_s * flop / (PEAK_FP32_MFMA_TFLOPS * 1e12)
    log("%s bs=256: %.0f img/s" % (cls, img_s))
    return {"workload": "%s MNIST bs=256 (%s.py defaults) through %sTrainer.train, warm" % (cls, mod_name, cls),
            "img_s": img_s, "ms_per_step": dt / (epochs * steps) * 1e3, "steps": epochs * steps, "step_mfma_frac": frac,
            "roofline": {"bound": "mfma", "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "flop_per_image": flop,
                         "flop_derivation": why, "step_frac": frac,
                         "profile": "profiles/%s_variant_%s_summary.md" % (PROFILE_ROUND, v)}}


def comm_mode_of(eng):
    """How this engine exchanges gradients (config.launch / the DP series entries)."""
    if eng.world == 1 and not eng.force_segments:
        return "none (1 rank)"
    if eng._peer():
        return "in-graph peer kernels over hipIpc/xGMI mappings (%s), %s exchange regions" % (
            eng.exchange_form(), eng.comm_memory)
    why = (" (peer exchange refused: %s)" % eng.comm_fallback) if getattr(eng, "comm_fallback", None) else ""
    if eng._rccl_in_graph():
        return "RCCL all-reduce captured inside the iteration's hipGraph" + why
    return "host-launched RCCL all-reduce between segment graphs" + why


def dp_identity(eng, world):
    """config.ranks: what every rank drives and how it exchanges -- one line of evidence per rank that the job really
    ran one process per GPU over the exchange it claims (device identity, exchange form, self-check outcome)."""
    from generative_models_amd import dp
    mine = {"rank": eng.rank, "device": dp._device_identity() if torch.cuda.is_available() else "cpu",
            "exchange": eng.exchange_form(), "exchange_memory": eng.comm_memory,
            "peer_selfcheck": "refused: %s" % eng.comm_fallback if getattr(eng, "comm_fallback", None)
            else ("passed" if eng._peer() else "not used"),
            # first contact: how long the self-check took under which device-side wait bound, per optimizer bucket
            "peer_selfcheck_detail": getattr(eng, "comm_selfcheck", None),
            "fallback": (("rccl_in_graph" if eng._rccl_in_graph() else "rccl (host-launched)") if not eng._peer() else None)}
    if world <= 1 or not torch.distributed.is_initialized():
        return [mine]
    out = [None] * world
    torch.distributed.all_gather_object(out, mine)
    return out


def dp_series(dev, world, rank, ranks_seen, K, W, reps):
    """BASELINE.json configs[4] / SURVEY.md 8(d)(5): NSGAN and LSGAN at batch 1024 across the `world`
    ranks of this run (loops being sharded: ns_gan.py:122-156, ls_gan.py:95-171), two series each:
      weak   -- 1024 rows PER RANK (global batch 1024 x N); eff = img/s(N) / (N x img/s(1))  [the >= 70 % target]
      strong -- GLOBAL batch 1024 (1024 / N rows per rank); speed-up = img/s(N) / img/s(1)
    img/s(1) is measured IN THIS RUN: rank 0 runs the 1-rank fused engine on its own GPU while the
    other ranks wait at the barrier.  Every timed region: barrier + synchronize on both sides, max
    over ranks, median of `reps` repetitions of K iterations."""
    out = []
    lrs = {"ns": (2e-4, 2e-4), "ls": (1e-4, 1e-4)}
    for variant in ("ns", "ls"):
        one = None
        if rank == 0:
            eng, secs = bench_gan(variant, 1024, W, K, reps, dev, world=1, rank=0, lrs=lrs[variant], solo=True)
            one = K * 1024 / float(np.median(secs))
            del eng
        fence(world)
        entry = {"workload": "%s MNIST batch 1024 data-parallel (BASELINE.json configs[4])"
                             % {"ns": "NSGAN", "ls": "LSGAN"}[variant],
                 "n_gpus": world, "ranks_seen": ranks_seen, "steps": K, "reps": reps,
                 "n1_img_s_same_run": one}
        for kind, Bg in (("weak", 1024 * world), ("strong", 1024)):
            eng, secs = bench_gan(variant, Bg, W, K, reps, dev, world=world, rank=rank, lrs=lrs[variant])
            dt = float(np.median(secs))
            e = {"global_batch": Bg, "rows_per_rank": Bg // world, "img_s": K * Bg / dt,
                 "ms_per_step": dt / K * 1e3, "reps_ms_per_step": [round(x / K * 1e3, 5) for x in secs],
                 "comm": comm_mode_of(eng)}
            if one:
                e["speedup_vs_n1"] = e["img_s"] / one
                e["efficiency"] = e["img_s"] / one / world
            entry[kind] = e
            log("%s %s N=%d: %.0f img/s" % (variant, kind, world, e["img_s"]))
            peer = eng._peer()
            del eng
            fence(world)
            if kind == "weak" and peer:
                # the same weak step with the FALLBACK exchange (RCCL all-reduce captured in the graph), side by side in
                # one invocation: which of the two a node should run is a measurement, not a default
                prev = os.environ.get("GM_DP_COMM")
                os.environ["GM_DP_COMM"] = "rccl"
                try:
                    eng, secs = bench_gan(variant, Bg, W, K, reps, dev, world=world, rank=rank, lrs=lrs[variant])
                    dt = float(np.median(secs))
                    r = {"global_batch": Bg, "img_s": K * Bg / dt, "ms_per_step": dt / K * 1e3, "comm": comm_mode_of(eng)}
                    if one:
                        r["efficiency"] = r["img_s"] / one / world
                    entry["weak_rccl"] = r
                    log("%s weak (rccl arm) N=%d: %.0f img/s" % (variant, world, r["img_s"]))
                    del eng
                finally:
                    if prev is None:
                        del os.environ["GM_DP_COMM"]
                    else:
                        os.environ["GM_DP_COMM"] = prev
                fence(world)
        out.append(entry)
    return out


def bench_trainer(epochs=3):
    """What a user of the reference's API gets (north_star: the path IS Trainer.train, ns_gan.py:94-170):
    `NSGANTrainer.train(epochs)` through the drop-in module -- per-epoch loss read-back, the reference's
    list extends (:159-160) and the epoch-end print included -- after one warm-up epoch on the same
    trainer (graph capture, clocks).  images = real images consumed by the critic steps."""
    import ns_gan
    ds = synthetic_dataset()
    mk = lambda: torch.utils.data.DataLoader(ds, batch_size=B_PER_GPU, shuffle=True)
    torch.manual_seed(1234)
    model = ns_gan.NSGAN(image_size=IMG, hidden_dim=HID, z_dim=Z)
    tr = ns_gan.NSGANTrainer(model, mk(), None, None, viz=False)
    steps = len(tr.train_iter)
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.train(epochs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert len(tr.Glosses) == (1 + epochs) * steps and np.isfinite(tr.Glosses).all()
    return {"what": "ns_gan.NSGANTrainer(model, train_iter, ...).train(%d) through the drop-in module, warm "
                    "(bs=256, %d iterations per epoch)" % (epochs, steps),
            "img_s": epochs * steps * B_PER_GPU / dt, "ms_per_step": dt / (epochs * steps) * 1e3,
            "steps": epochs * steps}


def bench_general_path(epochs=2):
    """The README's extension contract (/root/reference/README.md:29-65; loop ns_gan.py:122-156): a user subclass of
    NSGANTrainer that overrides ONLY train_D / train_G -- the README's own LSGAN edit -- at bs=256.  It cannot run on
    the fused engine; generative_models_amd/captured.py replays the two hooks as one captured graph per D+G iteration
    on the device data path.  One warm-up epoch on the same trainer (eager first iteration, capture), then `epochs`
    timed.  The fully general host loop (DataLoader reshuffle, CPU randn + H2D, .item() per step) is timed beside it
    on 40 iterations."""
    import ns_gan

    class ReadmeLS(ns_gan.NSGANTrainer):
        def train_D(self, images):
            noise = self.compute_noise(images.shape[0], self.model.z_dim)
            G_output = self.model.G(noise)
            DX_score, DG_score = self.model.D(images), self.model.D(G_output)
            return (0.50 * torch.mean((DX_score - 1.) ** 2)) + (0.50 * torch.mean((DG_score - 0.) ** 2))

        def train_G(self, images):
            noise = self.compute_noise(images.shape[0], self.model.z_dim)
            DG_score = self.model.D(self.model.G(noise))
            return 0.50 * torch.mean((DG_score - 1.) ** 2)

    ds = synthetic_dataset()

    def run(n_batches, warm, timed):
        class Capped(torch.utils.data.DataLoader):
            def __len__(self):
                return n_batches
        torch.manual_seed(1234)
        model = ns_gan.NSGAN(image_size=IMG, hidden_dim=HID, z_dim=Z)
        tr = ReadmeLS(model, Capped(ds, batch_size=B_PER_GPU, shuffle=True), None, None, viz=False)
        with contextlib.redirect_stdout(io.StringIO()):
            tr.train(warm)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr.train(timed)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        assert np.isfinite(tr.Glosses).all()
        return tr, dt / (timed * n_batches)

    steps = len(torch.utils.data.DataLoader(ds, batch_size=B_PER_GPU))
    tr, per = run(steps, 1, epochs)
    cap = getattr(tr, "_captured", None)
    out = {"what": "README.md:29-65 LSGAN override of ns_gan.NSGANTrainer (train_D / train_G only), bs=256, "
                   "%d iterations per epoch, %d epochs timed after one warm epoch" % (steps, epochs),
           "mode": (cap.mode if cap is not None and cap.done else "host loop"),
           "us_per_step": per * 1e6, "img_s": B_PER_GPU / per, "steps": epochs * steps}
    os.environ["GM_CAPTURED_GENERAL"] = "0"
    try:
        _, per_h = run(40, 1, 1)
    finally:
        del os.environ["GM_CAPTURED_GENERAL"]
    out["host_loop_us_per_step"] = per_h * 1e6
    return out


def self_launch(args):
    """`python bench.py --gpus N` from a bare shell: re-run under torch.distributed.run."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the K-step timed region (median reported)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs 3/4/5 section")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--sustained", type=float, default=6.0,
                    help="seconds of ONE uninterrupted window of the headline step behind the timed regions (0: skip)")
    ap.add_argument("--general-path", action="store_true", help="run only the README-override leg and print its entry")
    ap.add_argument("--only", default=None, help="profiling: run ONE of the extra configs (wgp_b256, ns_b1024, "
                    "ls_b1024, dra_b256, info_b256, be_b256, vae_b512) and print its entry instead of the contract line")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # dry run of the N > 1 code path on a box with ONE GPU (every rank on device 0, gloo as control
    # plane -- what tests/test_gpu_dp.py does): GM_BENCH_ONE_DEVICE=1.  Not a measurement.
    one_device = os.environ.get("GM_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    if world != args.gpus:
        sys.exit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    torch.cuda.set_device(local_rank)
    # diagnostic: GM_FORCE_DP=1 runs the data-parallel launch structure (segment graphs + RCCL
    # all-reduces of the gradient buckets) on ONE rank, to price its host/launch overhead
    force_dp = world == 1 and os.environ.get("GM_FORCE_DP") == "1"
    ranks_seen = 1
    if force_dp or world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        ranks_seen = dist.get_world_size()

    dev = torch.device("cuda", local_rank)
    W, K, reps = args.warmup, args.steps, max(1, args.reps)
    if args.general_path:
        print(json.dumps(bench_general_path()))
        return
    if args.only:
        print(json.dumps(other_configs(dev, min(K, 400), W, min(reps, 3), cpu=False, only=args.only)))
        return
    B_global = B_PER_GPU * world
    eng, secs = bench_gan("ns", B_global, W, K, reps, dev, world=world, rank=rank,
                          use_graph=not args.no_graph, force_dp=force_dp, long_steps=512,
                          sustained_s=(args.sustained if world == 1 else 0.0))
    dt = float(np.median(secs))
    log('timed regions done: %s' % ["%.4f" % x for x in secs])
    ranks_info = dp_identity(eng, world) if (world > 1 or force_dp) else None          # (collective: every rank)
    img_s = K * B_global / dt
    n1_img_s, series = None, None
    if world > 1:
        # N = 1 figure of THIS run (rank 0 alone, same K / W / reps), then the batch-1024 series
        if rank == 0:
            e1, s1 = bench_gan("ns", B_PER_GPU, W, K, reps, dev, world=1, rank=0,
                               use_graph=not args.no_graph, solo=True)
            n1_img_s = K * B_PER_GPU / float(np.median(s1))
            del e1
        fence(world)
        if not args.no_configs:
            series = dp_series(dev, world, rank, ranks_seen, max(K, 100), max(W, 10), min(reps, 3))

    if rank == 0:
        kt = time_kernels_isolated(B_PER_GPU, fused_head=eng.fuse_head, batch_gen=eng._batch_gen(),
                                   group_head=eng.group_head, fold_head=eng._fold_head(),
                                   ride_gather=eng._gather_rides(),
                                   pair_dw=eng.pair_dw, fused_adam=eng._adam_in_epilogue("G"),
                                   ride_head_dx=eng.ride_head_dx)
        mhz, cyc_per_mfma = clock_probe()
        log('clock probe: %.0f MHz effective, %.1f cycles per dependent v_mfma_f32_32x32x2_f32' % (mhz, cyc_per_mfma))
        dom = max(kt, key=lambda k: kt[k][0])
        t_us, flop, n = kt[dom]
        # HBM/fabric bytes per launch of that kernel from the COMMITTED PMC pass (profiles/: FETCH_SIZE
        # x2 gfx950 correction + WRITE_SIZE per dispatch; PMC counters cannot be read from inside this
        # process) -- the source file is named next to the number
        traffic, traffic_src = pmc_traffic(dom, "nsgan_b256")
        achieved = flop / (t_us * 1e-6) / 1e12
        line = {
            "metric": "images/sec (28x28 MNIST) per D+G step, NSGAN bs=256",
            "value": img_s, "unit": "images/sec", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "NSGAN MNIST bs=256 fp32 per GPU (BASELINE.json configs[1]); "
                                   "784-400-20 MLPs, N=50000 synthetic Bernoulli images, parity-mode "
                                   "RNG protocol, Adam 2e-4, D_steps=1",
                       "global_batch": B_global,
                       "launch": (("hipGraphs of up to %d iterations (8 launches each), draws staged in from a %d-slot "
                                   "pinned ring by the graph's first two nodes" % (eng.graph_iters, eng.R))
                                  if (world == 1 and not force_dp) else
                                  (("hipGraphs of up to %d iterations incl. 2 in-graph peer all-reduces (+Adam) per "
                                    "iteration over hipIpc/xGMI mappings" % eng.graph_iters)
                                   if eng._peer() else "hipGraph per segment + 2 RCCL all-reduces/iteration"))
                       if eng.use_graph else "eager",
                       "parallelism": "dp%d" % world, "ranks_seen": ranks_seen,
                       "gradient_exchange": comm_mode_of(eng),
                       "ranks": ranks_info,
                       "timing": "median of %d repetitions of the %d-step timed region" % (reps, K),
                       # a short timed region is COLD-START-INCLUSIVE: every repetition starts with no draws ahead of it
                       # (the host's first two iterations of draws, two graph hand-offs and the final synchronisation
                       # sit inside it: run_fixed_cost_us); the steady step is steady_us_per_step
                       "cold_start_inclusive": bool(K <= 200),
                       "reps_ms_per_step": [round(x / K * 1e3, 5) for x in secs],
                       "host_rng": "C replay (gm_host_replay)" if eng._replay_ok else "torch per-draw",
                       "host_threads": __import__("generative_models_amd").host_thread_plan(),
                       # SURVEY 8d's ">= 200 timed steps after warm-up" figure of the SAME engine, right behind the
                       # K-step regions (one 512-step region), and one uninterrupted multi-second window
                       "steady_512_steps": ({"steps": 512, "us_per_step": eng.steady_us_per_step,
                                             "img_s": B_global / eng.steady_us_per_step * 1e6}
                                            if eng.steady_us_per_step else None),
                       "sustained_window": eng.sustained,
                       # steady-state step of the same engine (the 512-step region) and what a K-step run() costs on
                       # top of K of those: cold start of the host draws, graph boundaries, final synchronize
                       "steady_us_per_step": eng.steady_us_per_step,
                       # (meaningful for short runs: at thousands of steps it is the noise of two measurements)
                       "run_fixed_cost_us": (dt * 1e6 - K * eng.steady_us_per_step)
                       if (eng.steady_us_per_step and K <= 200) else None},
            # (top-level copies of config.steady_us_per_step / run_fixed_cost_us / roofline.step_frac: tools read them)
            "steady_us_per_step": eng.steady_us_per_step,
            "run_fixed_cost_us": (dt * 1e6 - K * eng.steady_us_per_step) if (eng.steady_us_per_step and K <= 200) else None,
            "step_mfma_frac": img_s / world * FLOP_PER_IMAGE / (PEAK_FP32_MFMA_TFLOPS * 1e12),
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved,
                         "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "shader_clock_mhz": round(mhz),
                         "mfma_busy_frac": mfma_busy_frac(dom, t_us / n, mhz),
                         # fabric / HBM-side bytes of the committed PMC pass over this run's launch duration
                         "hbm_gbps": (traffic / (t_us / n * 1e-6) / 1e9) if traffic else None,
                         "hbm_frac_of_8tbs": (traffic / (t_us / n * 1e-6) / 8.0e12) if traffic else None,
                         "launches_per_step": n, "avg_launch_us": t_us / n,
                         "per_kernel_us_per_step": {k: round(v[0], 2) for k, v in kt.items()},
                         # the WHOLE step against the same peak (6 326 400 FLOP per image), and its compulsory HBM bytes
                         # (82 944 B per image at B = 256) against 8 TB/s: nowhere near the HBM roof
                         "step_frac": img_s / world * FLOP_PER_IMAGE / (PEAK_FP32_MFMA_TFLOPS * 1e12),
                         "step_hbm_frac": img_s / world * BYTES_PER_IMAGE_B256 / 8.0e12},
        }
        # the committed rocprofv3 average of the same kernel beside the live figure: a profile that no longer
        # describes the built kernels shows here, not in a reviewer's diff
        line["roofline"].update(profile_check(dom, t_us / n, "nsgan_b256"))
        del eng
        if world > 1:
            line["config"]["n1_img_s_same_run"] = n1_img_s
            line["config"]["efficiency_vs_n1_same_run"] = img_s / (world * n1_img_s) if n1_img_s else None
            if series is not None:
                line["dp_series"] = series
        if world == 1 and not force_dp:
            if not args.no_configs:
                # INSIDE `config`: the driver's record keeps the contract keys + config / roofline / cpu_baseline
                line["config"]["trainer"] = bench_trainer()
                gp = bench_general_path()
                gp["over_fast_step"] = gp["us_per_step"] / (dt / K * 1e6)
                line["config"]["general_path"] = gp
                # (GPU legs of the configs section first, every CPU baseline after them: see other_configs); each
                # entry carries its own roofline and cpu_baseline
                line["config"]["other_configs"] = other_configs(dev, min(K, 400), W, min(reps, 3),
                                                                cpu=not args.no_cpu_baseline)
            if not args.no_cpu_baseline:
                ref = cpu_baseline_reference(B_PER_GPU)          # the unmodified reference where it is mounted
                line["cpu_baseline"] = ref if ref is not None else cpu_baseline_gan("ns", B_PER_GPU)
                if ref is not None:
                    line["cpu_baseline"]["port"] = cpu_baseline_gan("ns", B_PER_GPU, seconds_target=6.0, cores=ref["cores"])
                line["cpu_baseline"]["compute_only"] = cpu_baseline_gan(
                    "ns", B_PER_GPU, seconds_target=5.0, compute_only=True, cores=line["cpu_baseline"]["cores"])
                line["cpu_baseline"]["gpu_over_cpu"] = img_s / line["cpu_baseline"]["value"]
        out_line = json.dumps(line)
    else:
        out_line = None
    if force_dp or world > 1:
        torch.distributed.barrier()          # rank 0 may still be in its reporting section
        torch.distributed.destroy_process_group()
    if out_line is not None:
        # RCCL writes its version banner through C stdio, which would otherwise be flushed at exit,
        # AFTER the result: push it out first so that the JSON line is the last thing on stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:                            # noqa: BLE001
            pass
        print(out_line, flush=True)


if __name__ == "__main__":
    main()
