#!/usr/bin/env python
"""bench.py -- images/sec of the NSGAN D+G step (BASELINE.json metric), bs=256 per GPU, fp32.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" is one full reference iteration (ns_gan.py:122-156): process_batch -> train_D ->
backward -> Adam(D) -> train_G -> backward -> Adam(G), in PARITY mode (reference RNG protocol,
bit-exact sampling indices).  The dataset (synthetic 50 000 x 28x28 Bernoulli images, seed 3435)
is resident in HBM before the timed region; the host-side RNG prefetch is inside it.
Weak scaling: every rank runs bs=256 (global batch 256*N), gradients all-reduced over RCCL.
Prints ONE JSON line on rank 0.
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
IMG, HID, Z, N_TRAIN = 784, 400, 20, 50000


def synthetic_dataset():
    torch.manual_seed(3435)
    img = torch.bernoulli(torch.full((N_TRAIN, 1, 28, 28), 0.1307))
    ds = torch.utils.data.TensorDataset(img, torch.zeros(N_TRAIN, dtype=torch.int64))
    return ds


def gemm_shapes(B, fused_head=True, batch_gen=True, group_head=True, ride_gather=True, pair_dw=True,
                ride_head_dx=True):
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
    return (gen + [("fwd", 2 * B, IMG, HID)] + d_head + [(dw1, 2 * B, IMG, HID)] +
            [("fwd", B, IMG, HID)] + g_head +
            # "dxh": the generator-mode head's scalar workgroup (loss + tick) rides in this launch
            [("dxh" if (fused_head and ride_head_dx) else "dx", B, IMG, HID), ("dx", B, HID, IMG)] + g_dw)


def gemm_variant(kind, M, K, N):
    """Name of the kernel instantiation csrc/gm_gemm.hip launches for this layer shape with the
    default settings (mirrors launch<MODE>(): v_mfma_f32_16x16x4_f32 kernel, 16 waves, per-chunk
    load/consume schedule, 16-byte paths by alignment, tile shape from the tile count) -- the name rocprofv3 reports.  Template order: MODE, VEC, WAVES, G, XV, MI, NI."""
    if kind in ("fwd", "fwdg"):
        mode, Mg, Ng, Kr, vec, xv = 0, M, N, K, K % 4 == 0, False
    elif kind in ("dx", "dxh"):
        mode, Mg, Ng, Kr, vec, xv = 1, M, K, N, N % 4 == 0, K % 4 == 0
    else:
        mode, Mg, Ng, Kr, vec = 2, N, K + 1, M, False
        xv = N % 4 == 0 and K % 4 == 0 and N >= 4 and K >= 4
    nw = 16
    chunks = -(-Kr // 16)
    g = 1                                  # per-chunk load/consume schedule
    tm, tn = -(-Mg // 32), -(-Ng // 32)
    mi, ni = 2, 2
    if tm * tn > 256 and chunks >= 32:                 # wide tiles: one round of workgroups
        mi, ni = (2, 4) if tn >= tm else (4, 2)
    elif tm * tn <= 128 and Mg > 16:                   # 16-row tiles: twice the workgroups
        mi, ni = 1, 2
    b = lambda v: "true" if v else "false"
    if kind == "dwh":
        return "gemm16_dw_head_kernel<false, %d, %s, %d, %d>" % (g, b(xv), mi, ni)
    if kind == "dxh":
        assert vec and xv and (mi, ni) in ((2, 2), (1, 2))
        return "gemm16_dx_head_kernel<%d, %d, %d>" % (g, mi, ni)
    if kind == "fwdg":
        assert vec and (mi, ni) in ((2, 2), (1, 2))
        return "gemm16_fwd_gather_kernel<true, %d, %d, %d>" % (g, mi, ni)
    if kind == "dwp":
        assert xv and (mi, ni) != (1, 2)
        return "gemm16_dw_pair_kernel<%d, true, %d, %d>" % (g, mi, ni)
    return "gemm16_kernel<%d, %s, %d, %d, %s, %d, %d>" % (mode, b(vec), nw, g, b(xv), mi, ni)


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
                          ride_gather=True, pair_dw=True, ride_head_dx=True, fused_adam=True):
    """HIP-event timing (on the launch stream) of each GEMM launch shape of the step, run back to
    back `reps` times.  Returns {kernel instantiation name: (total_us_per_step,
    total_flop_per_step, n_launches_per_step)}."""
    from generative_models_amd import ops
    dev = "cuda"
    out = {}
    st = ops.stream_ptr()
    data = idx = xr = None
    for kind, M, K, N in gemm_shapes(B, fused_head, batch_gen, group_head, ride_gather, pair_dw,
                                     ride_head_dx):
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
            L2, L1 = _holder(N, K, dev), _holder(K, Z, dev)
            dH2, zz = torch.randn(M, K, device=dev), torch.randn(M, Z, device=dev)
            sched = torch.from_numpy(ops.adam_schedule(2e-4, 4)).to(dev)
            ad = dict(sched=sched, sched_slot=ops.slot(0, 0, 1, 0, 1), clamp=0.0) if fused_adam else None
            fn = lambda: ops.linear_bwd_dw_adam_pair(dict(dA=dA, X=x, lin=L2, adam=ad),
                                                     dict(dA=dH2, X=zz, lin=L1, adam=ad), stream=st)
            flop += 2.0 * M * Z * K
        elif kind == "dx":
            fn = lambda: ops.linear_bwd_dx(dA, W, dX, below=x, epi="relu", stream=st)
        elif kind == "dwh":
            L1, L2 = _holder(N, K, dev), _holder(1, N, dev)
            Hh = torch.relu(torch.randn(M, N, device=dev))
            dS, rl, lo = torch.randn(M, device=dev) / M, torch.rand(M, device=dev), torch.zeros(1, device=dev)
            sched = torch.from_numpy(ops.adam_schedule(2e-4, 4)).to(dev)
            ad = dict(sched=sched, sched_slot=ops.slot(0, 0, 1, 0, 1), clamp=0.0) if fused_adam else None
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
        name = gemm_variant(kind, M, K, N)
        t, f, n = out.get(name, (0.0, 0.0, 0))
        out[name] = (t + us, f + flop, n + 1)
    return out


def log(msg):
    if os.environ.get("GM_BENCH_VERBOSE"):
        print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def cpu_baseline(seconds_target=12.0):
    """Oracle port (CPU restatement of the reference trainer, oracle/port.py) timed as-written
    (DataLoader reshuffle included) on this host's cores: NSGAN bs=256, same synthetic data.
    Thread count: the fastest of {8,16,32} in a short probe (hundreds of OpenMP threads on
    256x400 GEMMs only add barrier time); the count used is reported as `cores`."""
    from oracle import port
    ds = synthetic_dataset()
    loader = torch.utils.data.DataLoader(ds, batch_size=B_PER_GPU, shuffle=True)
    model = port.build("ns", IMG, HID, Z)
    tr = port.GANPort("ns", model, loader)
    ncpu = os.cpu_count() or 1
    best = None
    for cand in [c for c in (8, 16, 32) if c <= ncpu] or [min(ncpu, 4)]:
        torch.set_num_threads(cand)
        tr.train(1, max_steps=3)                    # warm-up at this thread count
        t0 = time.perf_counter()
        tr.train(1, max_steps=12)
        ps = (time.perf_counter() - t0) / 12
        log("cpu probe %.1f ms/step on %d threads" % (ps * 1e3, cand))
        if best is None or ps < best[0]:
            best = (ps, cand)
    per_step, cores = best
    torch.set_num_threads(cores)
    steps_per_epoch = int(np.ceil(len(loader)))
    total, done, dt = int(max(10, min(2000, seconds_target / per_step))), 0, 0.0
    while done < total:
        n = min(steps_per_epoch, total - done)
        t0 = time.perf_counter()
        tr.train(1, max_steps=n)
        dt += time.perf_counter() - t0
        done += n
    return {"value": done * B_PER_GPU / dt, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d NSGAN bs=256 D+G steps of oracle/port.py (torch %s CPU, %d threads), "
                      "as-written incl. DataLoader reshuffle, %.1f s" % (done, torch.__version__,
                                                                          cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node "
                     "%d bench.py --gpus %d ..." % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    pg = None
    # diagnostic: GM_FORCE_DP=1 runs the data-parallel launch structure (segment graphs + RCCL
    # all-reduces of the gradient buckets) on ONE rank, to price its host/launch overhead
    force_dp = world == 1 and os.environ.get("GM_FORCE_DP") == "1"
    if force_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from generative_models_amd import engine as gm_engine, ops
    import ns_gan

    ds = synthetic_dataset()
    B_global = B_PER_GPU * world
    loader = torch.utils.data.DataLoader(ds, batch_size=B_global, shuffle=True)
    torch.manual_seed(1234)
    model = ns_gan.NSGAN(image_size=IMG, hidden_dim=HID, z_dim=Z)
    trainer = ns_gan.NSGANTrainer(model, loader, None, None, viz=False)
    dev = torch.device("cuda", local_rank)
    data = ds.tensors[0].reshape(N_TRAIN, -1).to(dev).contiguous()          # resident in HBM
    eng = gm_engine.GANEngine("ns", trainer.model, data, B_global, dev,
                              use_graph=not args.no_graph,
                              world_size=world, rank=rank)
    eng.force_segments = force_dp
    W, K = args.warmup, args.steps
    log('engine built')
    eng.configure(W + K, 2e-4, 2e-4, 1)
    eng.run(W, it_start=0)
    log('warmup issued')

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    eng.run(K, it_start=W)
    fence()
    dt = time.perf_counter() - t0
    log('timed region done: %.3f s' % dt)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    G, D = eng.losses(W, W + K)
    assert np.isfinite(G).all() and np.isfinite(D).all(), "non-finite losses"
    img_s = K * B_global / dt

    if rank == 0:
        kt = time_kernels_isolated(B_PER_GPU, fused_head=eng.fuse_head, batch_gen=eng._batch_gen(),
                                   group_head=eng.group_head,
                                   ride_gather=eng._gather_rides(),
                                   pair_dw=eng.pair_dw, fused_adam=eng._adam_in_epilogue("G"),
                                   ride_head_dx=eng.ride_head_dx and not eng.head_final)
        mhz, cyc_per_mfma = clock_probe()
        log('clock probe: %.0f MHz effective, %.1f cycles per dependent v_mfma_f32_32x32x2_f32' % (mhz, cyc_per_mfma))
        log('isolated kernel timing done')
        dom = max(kt, key=lambda k: kt[k][0])
        t_us, flop, n = kt[dom]
        # HBM/fabric bytes per launch of that kernel from the committed PMC pass
        # (profiles/r01_pmc_fetch_write.md: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; PMC counters
        # cannot be read from inside this process, so the per-dispatch averages are loaded)
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            rows = [v for k, v in pmc.items() if k.split("|")[0] == dom]
            if rows:
                nd = sum(v["dispatches"] for v in rows)
                traffic = sum((v["read_bytes"] + v["write_bytes"]) * v["dispatches"] for v in rows) / nd
        except Exception:
            traffic = None
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
                       "launch": ("hipGraph/iteration" if (world == 1 and not force_dp) else "hipGraph per segment + 2 RCCL all-reduces/iteration") if eng.use_graph else "eager",
                       "parallelism": "dp%d" % world},
            "step_mfma_frac": img_s / world * FLOP_PER_IMAGE / (PEAK_FP32_MFMA_TFLOPS * 1e12),
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved,
                         "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                         "shader_clock_mhz": round(mhz),
                         "launches_per_step": n, "avg_launch_us": t_us / n,
                         "per_kernel_us_per_step": {k: round(v[0], 2) for k, v in kt.items()}},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if force_dp:
        torch.distributed.destroy_process_group()
    if world > 1:
        torch.distributed.barrier()          # rank 0 may still be in its reporting section
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
