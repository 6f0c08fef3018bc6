#!/usr/bin/env python
"""Isolated timing of GEMM launch shapes (hipGraph of back-to-back launches, HIP events on the launch
stream): python tools/gemm_shapes_bench.py fwd:2048:784:400 dx:1024:784:400 ...   (kind:M:K:N in
layer terms).  Kernel-selection knobs (GM_QUAD_MIN_TILES, GM_WIDE_TILES, ...) are read once per
process, so A/B runs are separate invocations."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from generative_models_amd import ops  # noqa: E402


def time_shape(kind, M, K, N, reps=50):
    dev = "cuda"
    x = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    dA = torch.randn(M, N, device=dev)
    y, dX = torch.empty(M, N, device=dev), torch.empty(M, K, device=dev)
    dW, db, b = torch.empty(N, K, device=dev), torch.empty(N, device=dev), torch.zeros(N, device=dev)
    st = [ops.stream_ptr()]
    if kind == "dwadam":
        # weight gradient with the optimizer in its epilogue (gm_linear_bwd_dw_adam), as the training steps launch it
        import torch.nn as nn
        from generative_models_amd.engine import FlatParams, _Linear
        net = nn.Sequential(nn.Linear(K, N))
        fp = FlatParams(net.parameters(), dev)
        lin = _Linear(fp, net[0])
        sched = torch.from_numpy(ops.adam_schedule(2e-4, 4)).to(dev)
        adam = dict(sched=sched, sched_slot=ops.slot(0, 0, 1, 0, 1), clamp=0.0)
        fn = lambda: ops.linear_bwd_dw_adam(dA, x, lin, adam, stream=st[0])
    elif kind == "fwdsig":
        fn = lambda: ops.linear_fwd(x, W, b, y, "sigmoid", stream=st[0])
    elif kind == "fwd":
        fn = lambda: ops.linear_fwd(x, W, b, y, "relu", stream=st[0])
    elif kind == "dx":
        fn = lambda: ops.linear_bwd_dx(dA, W, dX, below=x, epi="relu", stream=st[0])
    else:
        fn = lambda: ops.linear_bwd_dw(dA, x, dW, db, stream=st[0])
    fn()
    torch.cuda.synchronize()
    # correctness of this launch against torch fp32 (relative to the output's scale)
    if kind == "dwadam":
        ref, got = dA.t() @ x, lin.gW
    elif kind == "fwdsig":
        ref, got = torch.sigmoid(x @ W.t() + b), y
    elif kind == "fwd":
        ref, got = torch.relu(x @ W.t() + b), y
    elif kind == "dx":
        ref, got = (dA @ W) * (x > 0), dX
    else:
        ref, got = dA.t() @ x, dW
    err = float((got - ref).abs().max() / ref.abs().max())
    ablated = os.environ.get("GM_ABLATED_LIB", "0") != "0"    # timing-experiment build: results are meaningless
    assert ablated or err < 2e-5, (kind, M, K, N, err)
    if kind == "dw" and not ablated:
        rb = dA.sum(0)
        errb = float((db - rb).abs().max() / rb.abs().max())
        assert errb < 2e-5, (kind, M, K, N, "db", errb)

    def body(gst):
        st[0] = gst
        for _ in range(reps):
            fn()
    g = ops.Graph().capture(body)
    s = ops.stream_ptr()
    for _ in range(3):
        g.launch()
    e0, e1 = ops.Event(), ops.Event()
    e0.record(s)
    g.launch()
    e1.record(s)
    e1.sync()
    us = e0.elapsed_ms(e1) * 1e3 / reps
    # the vendor library on the same contraction (torch -> hipBLASLt / rocBLAS), same timing method;
    # fwd: bias fused by the library, relu not included; dx: without the relu mask; dw: without db
    if kind in ("fwd", "fwdsig"):
        vfn = lambda: torch.addmm(b, x, W.t(), out=y)
    elif kind == "dx":
        vfn = lambda: torch.mm(dA, W, out=dX)
    else:
        vfn = lambda: torch.mm(dA.t(), x, out=dW)
    vfn()
    torch.cuda.synchronize()
    tg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(tg):
        for _ in range(reps):
            vfn()
    for _ in range(3):
        tg.replay()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    tg.replay()
    t1.record()
    t1.synchronize()
    vus = t0.elapsed_time(t1) * 1e3 / reps
    return us, 2.0 * M * K * N / us / 1e6, vus


if __name__ == "__main__":
    for a in sys.argv[1:]:
        kind, M, K, N = a.split(":")
        us, tf, vus = time_shape(kind, int(M), int(K), int(N))
        print("%-3s M=%5s K=%4s N=%4s : %8.2f us  %6.2f TFLOP/s   | vendor GEMM (torch) %8.2f us  %6.2f TFLOP/s"
              % (kind, M, K, N, us, tf, vus, 2.0 * int(M) * int(K) * int(N) / vus / 1e6), flush=True)
