"""Data-parallel path on the MI355X: the peer-to-peer gradient exchange
(csrc/gm_comm.hip, dp.PeerComm) and the N-rank engine, exercised with SEVERAL RANKS ON ONE GPU --
the exchange goes through hipIpc mappings exactly as it does between GPUs of a node (there over
xGMI), the control plane is gloo.  N ranks must reproduce 1 rank up to fp32 summation order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(os.path.dirname(HERE), "generative_models_amd", "src")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port, real=False):
    """real: one GPU per rank and RCCL as the control plane (a node with >= `world` GPUs: the exchange
    then crosses xGMI links); otherwise every rank on GPU 0 with gloo."""
    local = rank if real else 0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world),
                      RANK=str(rank), LOCAL_RANK=str(local), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, SRC)
    import torch.distributed as dist
    torch.cuda.set_device(local)
    if real:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


needs_2_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2,
                                  reason="needs >= 2 GPUs (peer mappings over xGMI)")


def _comm_worker(rank, world, port, q, real=False):
    dist = _init(rank, world, port, real)
    from generative_models_amd import dp, ops
    dev = torch.device("cuda", torch.cuda.current_device())
    n = 314404
    try:
        comm = dp.PeerComm(n, world, rank)
    except Exception as e:                            # noqa: BLE001  (collective: every rank raises)
        q.put((rank, False, ["PeerComm: %r" % (e,)]))
        dist.destroy_process_group()
        return
    ok = comm.selfcheck(dev)
    g = torch.Generator().manual_seed(100)
    res = []
    for rnd in range(3):                               # changing data: a stale mapping would show
        bufs = [torch.randn(n, generator=g) for _ in range(world)]
        mine = bufs[rank].to(dev)
        comm.allreduce(mine)
        torch.cuda.synchronize()
        want = bufs[0].clone()
        for r in range(1, world):
            want += bufs[r]                            # rank order, fp32: bit-identical expected
        res.append(bool(torch.equal(mine.cpu(), want)))
    # all-reduce + Adam == Adam on the summed gradient
    grads = [torch.randn(n, generator=g) * 1e-2 for _ in range(world)]
    p0 = torch.randn(n, generator=g)
    gsum = grads[0].clone()
    for r in range(1, world):
        gsum += grads[r]
    sched = torch.from_numpy(ops.adam_schedule(2e-4, 2)).to(dev)
    pa, ma, va = p0.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    pb, mb, vb = p0.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    ga = grads[rank].to(dev)
    for step in range(2):
        comm.allreduce_adam(ga, pa, ma, va, sched, ops.slot(0, 0, step, 0, 1), clamp=0.01)
        ops.adam(pb, gsum.to(dev), mb, vb, sched, ops.slot(0, 0, step, 0, 1), clamp=0.01)
        torch.cuda.synchronize()
        res.append(bool(torch.equal(ga.cpu(), gsum)))
        ga = grads[rank].to(dev)
    res.append(bool(torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)))
    comm.check()
    q.put((rank, ok, res))
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


def _comm_case(world, real, one_kernel=False):
    """one_kernel: False (two launches), True (one kernel, remote reads) or "push" (one kernel, posted remote writes)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_comm_worker, args=(r, world, port, q, real)) for r in range(world)]
    # ranks sharing one device default to the two-launch exchange (dp.PeerComm); GM_DP_ONE_KERNEL=1 keeps the
    # one-kernel form, whose cross-rank protocol (arrival counter, last arriver signals) is what this case covers;
    # GM_DP_PUSH=1 the push form (two arrival counters, flag phases 4 / 5, staging areas)
    key = "GM_DP_PUSH" if one_kernel == "push" else "GM_DP_ONE_KERNEL"
    if one_kernel:
        os.environ[key] = "1"
    try:
        for p in procs:
            p.start()
    finally:
        os.environ.pop(key, None)
    out = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, res in out:
        assert ok, "communicator / selfcheck failed on rank %d: %s" % (rank, res)
        assert all(res), (rank, res)


@pytest.mark.parametrize("world", [1, 2, 4])
@pytest.mark.parametrize("one_kernel", [False, True, "push"], ids=["two_launches", "one_kernel", "push"])
def test_peer_allreduce_between_processes(world, one_kernel):
    _comm_case(world, real=False, one_kernel=one_kernel)


@needs_2_gpus
@pytest.mark.parametrize("form", [True, "push"], ids=["one_kernel", "push"])
def test_peer_allreduce_between_gpus(form):
    """One rank per GPU, up to 8: the exchange crosses the xGMI links (skipped on 1-GPU boxes)."""
    _comm_case(min(torch.cuda.device_count(), 8), real=True, one_kernel=form)


SMALL = dict(image_size=64, hidden_dim=48, z_dim=8, batch=16, n_train=160, n_val=48, n_test=48,
             image_shape=(1, 8, 8))
MODS = {"ns": ("ns_gan", "NSGAN", "NSGANTrainer"), "ls": ("ls_gan", "LSGAN", "LSGANTrainer"),
        "w": ("w_gan", "WGAN", "WGANTrainer"), "wgp": ("w_gp_gan", "WGPGAN", "WGPGANTrainer"),
        "mm": ("mm_gan", "MMGAN", "MMGANTrainer"), "f": ("f_gan", "fGAN", "fGANTrainer"),
        "ra": ("ra_gan", "RaNSGAN", "RaNSGANTrainer"), "fisher": ("fisher_gan", "FisherGAN", "FisherGANTrainer"),
        "dra": ("dra_gan", "DRAGAN", "DRAGANTrainer"), "be": ("be_gan", "BEGAN", "BEGANTrainer"),
        "info": ("info_gan", "InfoGAN", "InfoGANTrainer")}


def _train_worker(rank, world, port, variant, kw, q, real=False, cfg=None):
    dist = _init(rank, world, port, real) if world > 1 else None
    if world == 1:
        sys.path.insert(0, os.path.dirname(HERE))
        sys.path.insert(0, SRC)
        torch.cuda.set_device(0)
    import contextlib
    import importlib
    import io
    from oracle import port as oport
    mod_name, model_name, trainer_name = MODS[variant]
    mod = importlib.import_module(mod_name)
    cfg = cfg or SMALL
    loaders = oport.synthetic_loaders(cfg["batch"], n_train=cfg["n_train"], n_val=cfg["n_val"],
                                      n_test=cfg["n_test"], image_shape=tuple(cfg["image_shape"]))
    torch.manual_seed(1234)
    mkw = dict(image_size=cfg["image_size"], hidden_dim=cfg["hidden_dim"], z_dim=cfg["z_dim"])
    if variant == "info":
        mkw.update(disc_dim=10, cont_dim=10)
    model = getattr(mod, model_name)(**mkw)
    tr = getattr(mod, trainer_name)(model, *loaders, viz=False)
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(**kw)
    torch.cuda.synchronize()
    eng = tr._engine
    out = dict(rank=rank, G=list(tr.Glosses), D=list(tr.Dlosses), MI=list(getattr(tr, "MIlosses", [])),
               mode=eng.comm_mode, world=eng.world, xchg=eng.exchange_form(),
               params={k: v.cpu().numpy() for k, v in model.state_dict().items()},
               rng=torch.get_rng_state().numpy().tobytes())
    q.put(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _run_world(world, variant, kw, real=False, cfg=None, env=None):
    """env: extra environment of the rank processes (spawn copies os.environ at start())."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, variant, kw, q, real, cfg))
             for r in range(world)]
    saved = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        for p in procs:
            p.start()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    out = sorted([q.get(timeout=300) for _ in range(world)], key=lambda o: o["rank"])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


@pytest.mark.parametrize("variant,kw", [("ns", dict(num_epochs=2)), ("ls", dict(num_epochs=1)),
                                        ("w", dict(num_epochs=1, D_steps=2)),
                                        ("wgp", dict(num_epochs=1, D_steps=1)),
                                        ("mm", dict(num_epochs=1, G_init=2)),
                                        ("f", dict(num_epochs=1, method="pearson")),
                                        # not a mean of per-sample terms: scalar pre-reductions over the
                                        # global batch inside the step
                                        ("ra", dict(num_epochs=1)), ("fisher", dict(num_epochs=1)),
                                        ("dra", dict(num_epochs=1, D_steps=1)), ("be", dict(num_epochs=2)),
                                        ("info", dict(num_epochs=1))],       # three optimizers, MI bucket
                         ids=["ns", "ls", "w", "wgp", "mm", "f", "ra", "fisher", "dra", "be", "info"])
def test_two_rank_engine_equals_one_rank(variant, kw):
    """The ENGINE's N > 1 path (row shards, 1/B_global scaling, in-graph peer all-reduce + Adam) on
    two ranks == the single-rank fused engine: losses 1e-5, parameters 2e-5 (fp32 summation order),
    identical replicas, identical RNG stream position on every rank."""
    one = _run_world(1, variant, kw)[0]
    two = _run_world(2, variant, kw)
    assert all(o["world"] == 2 and o["mode"] == "peer" for o in two), [o["mode"] for o in two]
    for o in two:
        assert o["rng"] == one["rng"]
        g, d = np.array(o["G"]), np.array(o["D"])
        assert np.max(np.abs(g - np.array(one["G"])) / np.maximum(1, np.abs(one["G"]))) <= 1e-5
        assert np.max(np.abs(d - np.array(one["D"])) / np.maximum(1, np.abs(one["D"]))) <= 1e-5
        if variant == "info":
            mi = np.array(o["MI"])
            assert np.max(np.abs(mi - np.array(one["MI"])) / np.maximum(1, np.abs(one["MI"]))) <= 1e-5
        ptol = 1.5e-4 if variant == "be" else 2e-5         # BEGAN: sign() gradients, one Adam step
        for k, v in o["params"].items():
            assert np.max(np.abs(v - one["params"][k])) <= ptol, k
    for k, v in two[0]["params"].items():                 # replicas stay bit-identical
        assert np.array_equal(v, two[1]["params"][k]), k


@needs_2_gpus
@pytest.mark.parametrize("comm", ["peer", "rccl"])
@pytest.mark.parametrize("variant,kw", [("ns", dict(num_epochs=2)), ("wgp", dict(num_epochs=1, D_steps=2))], ids=["ns", "wgp"])
def test_two_gpu_engine_equals_one_rank(variant, kw, comm):
    """The same comparison with ONE GPU PER RANK and RCCL as control plane (skipped on 1-GPU boxes).  comm = peer:
    either exchange mode is acceptable (peer mappings, or the RCCL fallback if the self-check refused them);
    comm = rccl (GM_DP_COMM=rccl): the fallback itself -- RCCL all-reduces captured inside the iteration's graph."""
    one = _run_world(1, variant, kw)[0]
    two = _run_world(2, variant, kw, real=True, env={"GM_DP_COMM": comm})
    if comm == "rccl":
        assert all(o["xchg"] in ("rccl_in_graph", "rccl") for o in two), [o["xchg"] for o in two]
    for o in two:
        assert o["world"] == 2 and o["rng"] == one["rng"]
        g, d = np.array(o["G"]), np.array(o["D"])
        assert np.max(np.abs(g - np.array(one["G"])) / np.maximum(1, np.abs(one["G"]))) <= 1e-5
        assert np.max(np.abs(d - np.array(one["D"])) / np.maximum(1, np.abs(one["D"]))) <= 1e-5
        ptol = 1.5e-4 if variant == "be" else 2e-5
        for k, v in o["params"].items():
            assert np.max(np.abs(v - one["params"][k])) <= ptol, k
    for k, v in two[0]["params"].items():
        assert np.array_equal(v, two[1]["params"][k]), k


def _vae_worker(rank, world, port, kind, q, cfg=None, n_train=150):
    dist = _init(rank, world, port) if world > 1 else None
    if world == 1:
        sys.path.insert(0, os.path.dirname(HERE))
        sys.path.insert(0, SRC)
        torch.cuda.set_device(0)
    import contextlib
    import io
    from oracle import port as oport
    cfg = cfg or SMALL
    loaders = oport.synthetic_loaders(cfg["batch"], n_train=n_train, n_val=cfg["n_val"], n_test=cfg["n_test"],
                                      image_shape=tuple(cfg["image_shape"]))      # 150: ragged last batch of 6
    torch.manual_seed(1234)
    if kind == "vae":
        import vae
        model = vae.VAE(image_size=cfg["image_size"], hidden_dim=cfg["hidden_dim"], z_dim=cfg["z_dim"])
        tr = vae.VAETrainer(model, *loaders, viz=False)
    else:
        import ae
        model = ae.Autoencoder(image_size=cfg["image_size"], hidden_dim=cfg["z_dim"])
        tr = ae.AutoencoderTrainer(model, *loaders, viz=False)
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(num_epochs=2)
    torch.cuda.synchronize()
    q.put(dict(rank=rank, recon=list(tr.recon_loss), kl=list(getattr(tr, "kl_loss", [])),
               best=float(tr.best_val_loss), world=tr._engine.world,
               params={k: v.cpu().numpy() for k, v in model.state_dict().items()},
               rng=torch.get_rng_state().numpy().tobytes()))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _run_vae_world(world, kind, cfg=None, n_train=150):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_vae_worker, args=(r, world, port, kind, q, cfg, n_train)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=300) for _ in range(world)], key=lambda o: o["rank"])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def _assert_vae_equal(one, many):
    for o in many:
        assert o["world"] == len(many) and o["rng"] == one["rng"]
        for key in ("recon", "kl"):
            a, b = np.array(o[key]), np.array(one[key])
            assert a.shape == b.shape
            if a.size:
                assert np.max(np.abs(a - b) / np.maximum(1, np.abs(b))) <= 2e-5, key
        assert abs(o["best"] - one["best"]) <= 2e-5 * abs(one["best"])
        for k, v in o["params"].items():
            assert np.max(np.abs(v - one["params"][k])) <= 2e-5, k
    for o in many[1:]:
        for k, v in many[0]["params"].items():
            assert np.array_equal(v, o["params"][k]), k


@pytest.mark.parametrize("kind", ["vae", "ae"])
def test_two_rank_vae_equals_one_rank(kind):
    """vae.py / ae.py under data parallelism: every batch's rows (incl. the ragged last one) split
    over two ranks, gradient SUM without 1/N (the losses are sums, vae.py:203,212), per-rank loss
    slots add up to the single-rank values."""
    one = _run_vae_world(1, kind)[0]
    _assert_vae_equal(one, _run_vae_world(2, kind))


# ---------------------------------------------------------------------------------------------
# The same comparisons at the REAL layer widths and batch sizes of BASELINE.json configs[2..4]
# (784-400-20; NSGAN / LSGAN global B = 1024 on 2 / 4 / 8 ranks = 512 / 256 / 128 rows per rank: other
# tile shapes than the 1-rank run, the LDS macro-tile kernel on one side of its M >= 1024 threshold
# and the split-reduction kernel on the other; WGAN-GP B = 256; VAE B = 512 incl. the ragged 336
# batch = 168 rows per rank).  Loops being sharded: ns_gan.py:122-156, ls_gan.py:95-171,
# w_gp_gan.py:96-175, vae.py:144-167.
# ---------------------------------------------------------------------------------------------
FULL = dict(image_size=784, hidden_dim=400, z_dim=20, n_val=64, n_test=64, image_shape=(1, 28, 28))
_ONE_RANK = {}


def _full_cfg(batch, steps_per_epoch=4):
    return dict(FULL, batch=batch, n_train=batch * steps_per_epoch)


def _one_rank_full(variant, batch, kw):
    key = (variant, batch)
    if key not in _ONE_RANK:
        _ONE_RANK[key] = _run_world(1, variant, kw, cfg=_full_cfg(batch))[0]
    return _ONE_RANK[key]


@pytest.mark.parametrize("variant,batch,world,kw",
                         [("ns", 1024, 2, dict(num_epochs=2)), ("ns", 1024, 4, dict(num_epochs=2)),
                          ("ns", 1024, 8, dict(num_epochs=2)),
                          ("ls", 1024, 2, dict(num_epochs=2)), ("ls", 1024, 4, dict(num_epochs=2)),
                          ("ls", 1024, 8, dict(num_epochs=2)),
                          ("wgp", 256, 2, dict(num_epochs=2, D_steps=1))],
                         ids=["ns1024-w2", "ns1024-w4", "ns1024-w8", "ls1024-w2", "ls1024-w4", "ls1024-w8",
                              "wgp256-w2"])
def test_n_rank_engine_equals_one_rank_full_size(variant, batch, world, kw):
    """BASELINE.json configs[4] in its multi-GPU form (and configs[2] on two ranks): 8 free-running
    D+G iterations at 784-400-20, global batch split over `world` ranks sharing GPU 0, against the
    1-rank fused engine: losses 1e-5, parameters 2e-5, replicas bit-identical, RNG position equal."""
    one = _one_rank_full(variant, batch, kw)
    many = _run_world(world, variant, kw, cfg=_full_cfg(batch))
    assert all(o["world"] == world and o["mode"] == "peer" for o in many), [o["mode"] for o in many]
    assert len(one["G"]) == 8
    for o in many:
        assert o["rng"] == one["rng"]
        for key in ("G", "D"):
            a, b = np.array(o[key]), np.array(one[key])
            assert a.shape == b.shape
            assert np.max(np.abs(a - b) / np.maximum(1, np.abs(b))) <= 1e-5, key
        for k, v in o["params"].items():
            assert np.max(np.abs(v - one["params"][k])) <= 2e-5, k
    for o in many[1:]:
        for k, v in many[0]["params"].items():
            assert np.array_equal(v, o["params"][k]), k


@pytest.mark.parametrize("variant,batch,world,kw", [("ns", 1024, 2, dict(num_epochs=2)), ("ls", 1024, 4, dict(num_epochs=2)),
                                                   ("wgp", 256, 2, dict(num_epochs=2, D_steps=1))],
                         ids=["ns1024-w2", "ls1024-w4", "wgp256-w2"])
def test_n_rank_training_with_the_one_kernel_exchange(variant, batch, world, kw):
    """The exchange form a node with one GPU per rank runs (`xchg_kernel`: reduce-scatter + all-gather + Adam in ONE
    launch per optimizer, csrc/gm_comm.hip) driven through whole TRAINING runs: ranks that share a device default to
    the two-launch form, GM_DP_ONE_KERNEL=1 keeps the one-kernel form.  N ranks == 1 rank as in the test above, and
    the two forms agree bit for bit with each other (same per-element summation order over ranks)."""
    one = _one_rank_full(variant, batch, kw)
    many = _run_world(world, variant, kw, cfg=_full_cfg(batch), env={"GM_DP_ONE_KERNEL": "1"})
    assert all(o["world"] == world and o["mode"] == "peer" and o["xchg"] == "one_kernel" for o in many), \
        [(o["mode"], o["xchg"]) for o in many]
    for o in many:
        assert o["rng"] == one["rng"]
        for key in ("G", "D"):
            a, b = np.array(o[key]), np.array(one[key])
            assert a.shape == b.shape and np.max(np.abs(a - b) / np.maximum(1, np.abs(b))) <= 1e-5, key
        for k, v in o["params"].items():
            assert np.max(np.abs(v - one["params"][k])) <= 2e-5, k
    for o in many[1:]:
        for k, v in many[0]["params"].items():
            assert np.array_equal(v, o["params"][k]), k
    two_launch = _run_world(world, variant, kw, cfg=_full_cfg(batch))
    assert all(o["xchg"] == "two_kernels" for o in two_launch)
    for k, v in many[0]["params"].items():
        assert np.array_equal(v, two_launch[0]["params"][k]), "one-kernel vs two-launch exchange: %s" % k
    # ... and the PUSH form (posted remote writes only: round 5) gives the same bits again
    pushed = _run_world(world, variant, kw, cfg=_full_cfg(batch), env={"GM_DP_PUSH": "1"})
    assert all(o["xchg"] == "one_kernel_push" for o in pushed), [o["xchg"] for o in pushed]
    for o in pushed:
        assert o["rng"] == one["rng"] and o["G"] == many[0]["G"] and o["D"] == many[0]["D"]
        for k, v in o["params"].items():
            assert np.array_equal(v, many[0]["params"][k]), "push vs pull exchange: %s" % k


def test_two_rank_vae_equals_one_rank_full_size():
    """BASELINE.json configs[3] on two ranks: VAE 784-400-20, B = 512, n = 1360 so that every epoch
    ends with the ragged 336 batch of the real 50 000-image run (168 rows per rank)."""
    cfg = dict(FULL, batch=512)
    one = _run_vae_world(1, "vae", cfg, 1360)[0]
    assert len(one["recon"]) == 6                     # 2 epochs x (512, 512, 336)
    _assert_vae_equal(one, _run_vae_world(2, "vae", cfg, 1360))
