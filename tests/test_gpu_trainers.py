"""Trainer-level parity on the MI355X: the drop-in <Name>Trainer.train() (fused hipGraph engine)
against (a) the CPU oracle (oracle/port.py) run here on identical seeds and (b) the committed
golden fixtures generated from the unmodified reference.  Checked: per-step loss lists, final
parameters, and that the global CPU generator ends in the same state (bit-exact draw protocol).

Tolerance: 1e-5 (north_star) on losses relative to max(1,|loss|) over the free-running horizon
used here (<= 24 steps; SURVEY.md section 4 explains why longer free-running horizons diverge
chaotically even between two CPU summation orders); parameters 1e-5 absolute on O(0.1) weights.
"""
import glob
import importlib
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(os.path.dirname(HERE), "generative_models_amd", "src")
GOLDEN = os.path.join(HERE, "golden")
sys.path.insert(0, SRC)

from oracle import port  # noqa: E402

MODS = {"ns": ("ns_gan", "NSGAN", "NSGANTrainer"), "mm": ("mm_gan", "MMGAN", "MMGANTrainer"),
        "w": ("w_gan", "WGAN", "WGANTrainer"), "wgp": ("w_gp_gan", "WGPGAN", "WGPGANTrainer"),
        "ls": ("ls_gan", "LSGAN", "LSGANTrainer"), "ra": ("ra_gan", "RaNSGAN", "RaNSGANTrainer"),
        "fisher": ("fisher_gan", "FisherGAN", "FisherGANTrainer"),
        "f": ("f_gan", "fGAN", "fGANTrainer"), "dra": ("dra_gan", "DRAGAN", "DRAGANTrainer"),
        "be": ("be_gan", "BEGAN", "BEGANTrainer"), "info": ("info_gan", "InfoGAN", "InfoGANTrainer")}
TOL = 1e-5


def lclose(got, ref, what, tol=TOL):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() <= tol, "%s: max err %.3e at step %d\n got %s\n ref %s" % (
        what, err.max(), int(err.argmax()), got[:6], ref[:6])


def build_product(variant, cfg, batch, loaders=None, use_graph=True):
    mod_name, model_name, trainer_name = MODS[variant]
    mod = importlib.import_module(mod_name)
    if loaders is None:
        loaders = port.synthetic_loaders(batch, n_train=cfg["n_train"], n_val=cfg["n_val"],
                                         n_test=cfg["n_test"],
                                         image_shape=tuple(cfg["image_shape"]))
    torch.manual_seed(1234)
    kw = dict(image_size=cfg["image_size"], hidden_dim=cfg["hidden_dim"], z_dim=cfg["z_dim"])
    if variant == "info":
        kw.update(disc_dim=10, cont_dim=10)
    model = getattr(mod, model_name)(**kw)
    tr = getattr(mod, trainer_name)(model, *loaders, viz=False)
    tr.use_graph = use_graph
    return tr, model


def run_product(variant, cfg, batch, train_kw, use_graph=True, capped=None, viz_dir=None):
    tr, model = build_product(variant, cfg, batch, use_graph=use_graph)
    if viz_dir is not None:                                # ns_gan.py:166-170 with viz=True
        tr.viz, tr.viz_dir = True, viz_dir
    if capped is not None:
        class Capped(torch.utils.data.DataLoader):
            def __len__(self):
                return capped
        tr.train_iter = Capped(tr.train_iter.dataset, batch_size=batch, shuffle=True)
    kw = dict(train_kw)
    if variant == "f":
        kw["method"] = kw.get("method", "jensen_shannon")
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(**kw)
    torch.cuda.synchronize()
    return tr, model, torch.get_rng_state()


def run_oracle(variant, cfg, batch, train_kw, max_steps=None):
    loaders = port.synthetic_loaders(batch, n_train=cfg["n_train"], n_val=cfg["n_val"],
                                     n_test=cfg["n_test"], image_shape=tuple(cfg["image_shape"]))
    model = port.build(variant, cfg["image_size"], cfg["hidden_dim"], cfg["z_dim"])
    kw = dict(train_kw)
    method = kw.pop("method", "jensen_shannon")
    tr = port.GANPort(variant, model, loaders[0], method=method)
    tr.train(max_steps=max_steps, **kw)
    return tr, model, torch.get_rng_state()


SMALL = dict(image_size=64, hidden_dim=48, z_dim=8, batch=16, n_train=160, n_val=48, n_test=48,
             image_shape=(1, 8, 8))
RAGGED = dict(image_size=100, hidden_dim=70, z_dim=10, batch=24, n_train=200, n_val=48, n_test=48,
              image_shape=(1, 10, 10))

CASES = [("ns", dict(num_epochs=2)), ("mm", dict(num_epochs=1, G_init=3)),
         ("w", dict(num_epochs=1, D_steps=2)), ("ls", dict(num_epochs=2)),
         ("ra", dict(num_epochs=1)), ("fisher", dict(num_epochs=1)),
         ("wgp", dict(num_epochs=1, D_steps=1)), ("wgp", dict(num_epochs=1, D_steps=3)),
         ("dra", dict(num_epochs=1, D_steps=1)), ("be", dict(num_epochs=2)),
         ("info", dict(num_epochs=1))] + \
        [("f", dict(num_epochs=1, method=m)) for m in port.F_METHODS]


@pytest.mark.parametrize("cfg", [SMALL, RAGGED], ids=["small", "ragged"])
@pytest.mark.parametrize("variant,kw", CASES, ids=["%s%d" % (v, i) for i, (v, _) in enumerate(CASES)])
def test_engine_vs_oracle(variant, kw, cfg):
    o_tr, o_model, o_rng = run_oracle(variant, cfg, cfg["batch"], kw)
    p_tr, p_model, p_rng = run_product(variant, cfg, cfg["batch"], kw)
    lclose(p_tr.Dlosses, o_tr.Dlosses, "%s Dlosses" % variant)
    lclose(p_tr.Glosses, o_tr.Glosses, "%s Glosses" % variant)
    assert torch.equal(o_rng, p_rng), "global CPU generator must end in the reference's state"
    osd, psd = o_model.state_dict(), p_model.state_dict()
    assert list(osd.keys()) == list(psd.keys())
    # BEGAN's L1 loss has sign() gradients: a sign flip near |D(x)-x| = 0 moves a parameter by a
    # full Adam step (lr = 1e-4), so its parameters get a one-step allowance
    ptol = 1.5e-4 if variant == "be" else 2e-5
    for k in osd:
        err = (psd[k].cpu() - osd[k]).abs().max().item()
        assert err <= ptol, (k, err)
    assert p_tr.num_epochs == kw["num_epochs"]
    assert p_tr._engine is not None and p_tr._stock(), "the fused hipGraph engine was not used"
    if variant == "be":
        assert abs(p_tr.K - o_tr.K) <= 1e-6
    if variant == "info":
        lclose(p_tr.MIlosses, o_tr.MIlosses, "info MIlosses")


TF_CASES = [("ns", {}), ("mm", dict(G_init=0)), ("w", {}), ("ls", {}), ("ra", {}), ("fisher", {}),
            ("wgp", {}), ("dra", {}), ("be", {}), ("info", {})] + [("f", dict(method=m)) for m in port.F_METHODS]


@pytest.mark.parametrize("variant,kw", TF_CASES, ids=["%s%d" % (v, i) for i, (v, _) in enumerate(TF_CASES)])
def test_teacher_forced_single_step_gradients(variant, kw):
    """ONE D+G iteration from identical parameters, images and noise: the gradient buffers the fused
    engine leaves behind (dL_D/dD-params of the critic step, dL_G/dG-params of the generator step on
    the updated critic; InfoGAN: the MI step's G and Q gradients) against the oracle's autograd
    gradients, tensor by tensor.  Catches a compensating error pair that a loss curve would hide."""
    cfg = SMALL
    grads = {}

    def tap(kind, tr, info):
        m = tr.model
        if kind == "D":
            grads["D"] = [p.grad.detach().clone() for p in m.D.parameters()]
        elif kind == "G" and variant != "info":
            grads["G"] = [p.grad.detach().clone() for p in m.G.parameters()]
        elif kind == "Q":
            grads["G"] = [p.grad.detach().clone() for p in m.G.parameters()]
            grads["Q"] = [p.grad.detach().clone() for p in m.Q.parameters()]
    loaders = port.synthetic_loaders(cfg["batch"], n_train=cfg["n_train"], n_val=cfg["n_val"],
                                     n_test=cfg["n_test"], image_shape=tuple(cfg["image_shape"]))
    o_model = port.build(variant, cfg["image_size"], cfg["hidden_dim"], cfg["z_dim"])
    okw = dict(kw)
    o = port.GANPort(variant, o_model, loaders[0], method=okw.pop("method", "jensen_shannon"), tap=tap)
    o.train(num_epochs=1, D_steps=1, max_steps=1, **okw)
    p_tr, p_model, _ = run_product(variant, cfg, cfg["batch"], dict(num_epochs=1, D_steps=1, **kw), capped=1)
    eng = p_tr._engine
    assert eng is not None and len(p_tr.Glosses) == 1
    torch.cuda.synchronize()
    for net, fp in (("D", eng.fD), ("G", eng.fG)) + ((("Q", eng.fQ),) if variant == "info" else ()):
        assert len(fp.params) == len(grads[net])
        for p_, off, ref in zip(fp.params, fp.offsets, grads[net]):
            got = fp.grad[off:off + p_.numel()].view(p_.shape).cpu()
            # relative to the tensor's scale, plus 2e-7 absolute: the critic's output-bias gradient
            # is a sum of two nearly cancelling halves (e.g. WGAN: -mean(s'x) + mean(s'g))
            aerr = float((got - ref).abs().max())
            tol = (2e-3 if variant == "be" else 2e-5) * float(ref.abs().max()) + 2e-7
            # BEGAN: sign() gradients flip where |D(x)-x| crosses 0 within fp32 rounding
            assert aerr <= tol, (variant, net, tuple(p_.shape), aerr, tol)


def test_eager_equals_graph():
    """The hipGraph replay and the eager launch sequence are the same kernels: bitwise equal."""
    a = run_product("ns", SMALL, 16, dict(num_epochs=2), use_graph=True)
    b = run_product("ns", SMALL, 16, dict(num_epochs=2), use_graph=False)
    assert a[0].Glosses == b[0].Glosses and a[0].Dlosses == b[0].Dlosses
    for (k, x), (_, y) in zip(a[1].state_dict().items(), b[1].state_dict().items()):
        assert torch.equal(x, y), k


def test_fp32_resident_dataset_equals_bit_packed(monkeypatch):
    """GM_PACKED=0 keeps the dataset as fp32 rows (3136 B/row); the default packs a binary dataset to
    1 bit/pixel.  Same batches either way -> bitwise the same run."""
    a = run_product("ns", SMALL, 16, dict(num_epochs=2))
    from generative_models_amd import ops
    assert isinstance(a[0]._engine.data, ops.PackedData)
    monkeypatch.setenv("GM_PACKED", "0")
    b = run_product("ns", SMALL, 16, dict(num_epochs=2))
    assert torch.is_tensor(b[0]._engine.data)
    assert a[0].Glosses == b[0].Glosses and a[0].Dlosses == b[0].Dlosses
    for (k, x), (_, y) in zip(a[1].state_dict().items(), b[1].state_dict().items()):
        assert torch.equal(x, y), k


PK_CFG = dict(image_size=784, hidden_dim=400, z_dim=20, n_train=2048, n_val=256, n_test=256, image_shape=(1, 28, 28))
PK_CASES = [("ns", PK_CFG, 256, {}), ("ns", PK_CFG, 64, {}), ("ls", PK_CFG, 256, {}), ("w", PK_CFG, 256, dict(D_steps=2)),
            ("mm", PK_CFG, 128, dict(G_init=1)), ("info", PK_CFG, 256, {}), ("f", PK_CFG, 256, dict(method="pearson")),
            ("ns", SMALL, 32, {}), ("ls", SMALL, 64, {})]


@pytest.mark.parametrize("variant,cfg,batch,kw", PK_CASES,
                         ids=["%s_b%d_%d" % (v, b, c["image_size"]) for v, c, b, _ in PK_CASES])
def test_packed_operand_rows_equal_fp32_rows(variant, cfg, batch, kw, monkeypatch):
    """GM_PACKED_OPERAND=1: the critic step reads its real rows as bits -- the gather writes
    100 B per row instead of 3136, the fp32 rows of X2 are never written -- and the run is bitwise the run on fp32 rows."""
    a = run_product(variant, cfg, batch, dict(num_epochs=2, **kw), capped=6)
    assert not a[0]._engine._packed_operand()
    assert float(a[0]._engine.X2[:batch].abs().max()) > 0.0
    monkeypatch.setenv("GM_PACKED_OPERAND", "1")
    b = run_product(variant, cfg, batch, dict(num_epochs=2, **kw), capped=6)
    eng = b[0]._engine
    assert eng._packed_operand() and bool(eng.Xbits.any())
    assert float(eng.X2[:batch].abs().max()) == 0.0          # the fp32 real rows were never written
    assert a[0].Glosses == b[0].Glosses and a[0].Dlosses == b[0].Dlosses
    for (k, x), (_, y) in zip(a[1].state_dict().items(), b[1].state_dict().items()):
        assert torch.equal(x, y), k


def test_packed_operand_is_not_taken_where_the_step_cannot_carry_it(monkeypatch):
    """Ragged batches (24 rows: not whole 32-row tiles) and the penalty variants keep the fp32 rows."""
    monkeypatch.setenv("GM_PACKED_OPERAND", "1")
    a = run_product("ns", RAGGED, 24, dict(num_epochs=1))
    assert not a[0]._engine._packed_operand() and float(a[0]._engine.X2[:24].abs().max()) > 0.0
    b = run_product("wgp", SMALL, 16, dict(num_epochs=1, D_steps=1))
    assert not b[0]._engine._packed_operand()


@pytest.mark.parametrize("variant,env", [("wgp", "GM_WGP_PEN_IN_HEAD"), ("wgp", "GM_WGP_STACK"), ("ns", "GM_FOLD_HEAD"),
                                         ("wgp", "GM_FOLD_HEAD"), ("dra", "GM_DRA_STACK"), ("ra", "GM_FOLD_HEAD_TP"),
                                         ("fisher", "GM_FOLD_HEAD_TP")])
def test_summation_order_switches_agree_within_rounding(variant, env, monkeypatch):
    """Three switches change the ORDER in which the same products are summed, not what is computed: the gradient
    penalty's share of gw2 added per row group inside the head workgroups or after the column sum
    (GM_WGP_PEN_IN_HEAD), the penalty's layer-1 gradient stacked into the critic's own launch or accumulated by
    separate launches (GM_WGP_STACK; GM_DRA_STACK: DRAGAN's three layer-1 gradients as one 4B-row reduction, the
    sigma'' path's share of the head's gradient added inside the head's backward), a score as 13 tile partials or as
    one wave's dot product (GM_FOLD_HEAD); GM_FOLD_HEAD_TP: RaGAN's / Fisher's critic step with the batch means formed
    in every consumer workgroup's prologue (fp64 block sums over 1024 threads) or by the one-workgroup loss kernel
    between separate head launches (fp64 sums over 256 threads) -- Fisher's lambda included through the losses.  The
    two settings of each must agree to fp32 rounding over a short run (results are bit-reproducible per setting, not
    across settings -- and not across the rounds in which a default changed)."""
    kw = dict(num_epochs=2) if variant in ("ns", "ra", "fisher") else dict(num_epochs=2, D_steps=2)
    a = run_product(variant, SMALL, 16, kw)
    monkeypatch.setenv(env, "0")
    b = run_product(variant, SMALL, 16, kw)
    lclose(np.array(a[0].Glosses), np.array(b[0].Glosses), env + " G losses", tol=2e-5)
    lclose(np.array(a[0].Dlosses), np.array(b[0].Dlosses), env + " D losses", tol=2e-5)
    for (k, x), (_, y) in zip(a[1].state_dict().items(), b[1].state_dict().items()):
        # Adam turns a last-bit difference of a tiny gradient into a fraction of a step: bound by one step
        assert float((x - y).abs().max()) <= 2.5e-4, (env, k)
        assert float((x - y).abs().mean()) <= 2e-5, (env, k)


@pytest.mark.parametrize("variant,kw", [("ns", dict(num_epochs=4)), ("w", dict(num_epochs=3, D_steps=2)),
                                        ("info", dict(num_epochs=3)), ("fisher", dict(num_epochs=3))])
def test_epochs_enqueued_ahead_of_the_loss_read_back_change_nothing(variant, kw, monkeypatch, capsys):
    """trainers._train enqueues epoch e+1 before it reads epoch e's losses back (side stream, behind an event at the
    epoch's end): loss histories, progress lines, parameters and the generator state must equal the run that reads
    back before enqueueing (GM_PIPELINE_EPOCHS=0)."""
    def go():
        tr, model = build_product(variant, SMALL, 16)
        tr.train(**kw)
        torch.cuda.synchronize()
        return tr, model, torch.get_rng_state(), capsys.readouterr().out
    a = go()
    monkeypatch.setenv("GM_PIPELINE_EPOCHS", "0")
    b = go()
    assert a[0].Glosses == b[0].Glosses and a[0].Dlosses == b[0].Dlosses and a[0].num_epochs == b[0].num_epochs
    assert len(a[0].Glosses) == kw["num_epochs"] * int(np.ceil(10 / kw.get("D_steps", 1)))
    if variant == "info":
        assert a[0].MIlosses == b[0].MIlosses
    assert a[3] == b[3] and a[3].count("Epoch[") == kw["num_epochs"]
    assert torch.equal(a[2], b[2])
    for (k, x), (_, y) in zip(a[1].state_dict().items(), b[1].state_dict().items()):
        assert torch.equal(x, y), k


def test_run_to_run_determinism():
    a = run_product("ls", SMALL, 16, dict(num_epochs=1))
    b = run_product("ls", SMALL, 16, dict(num_epochs=1))
    assert a[0].Glosses == b[0].Glosses and a[0].Dlosses == b[0].Dlosses


def test_failed_host_draws_raise_and_never_leave_the_gpu_waiting():
    """A draw job that fails on the fill worker (here: an op outside the restated paths injected into
    the programme of a later ring slot) must surface as GMError from run() within moments -- the fill
    gate of every already-enqueued graph is opened by the error path, nothing waits for its 20 s
    time-out -- and the engine must train normally again afterwards."""
    import time
    from generative_models_amd import _lib, engine
    tr, model = build_product("ns", SMALL, SMALL["batch"])
    eng = tr._get_engine()
    eng.configure(40, 2e-4, 2e-4, 1)
    eng.run(8, it_start=0)
    torch.cuda.synchronize()
    if not eng._native_fill:
        pytest.skip("native fill worker not in use")
    # corrupt the draw programme of ring slot 12: normal_ on 8 elements is unsupported by the replay
    bad_dst = torch.empty(8)
    v = eng._host_views(12 % eng.R)
    good = v["program"]
    ops_list = [engine.HostReplay.op(_lib.DRAW_NORMAL, 8, bad_dst, 0)]
    v["program"] = (_lib.DrawOp * 1)(*ops_list)
    # (a cold run's sub-chunks are 1, 1, 2, 4, ... iterations: the fourth one starts at iteration 12)
    t0 = time.perf_counter()
    with pytest.raises(_lib.GMError):
        eng.run(32, it_start=8)
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 5.0, "the error path waited for a gate time-out"
    v["program"] = good
    assert torch.equal(torch.ones(3, device="cuda") * 2, torch.full((3,), 2.0, device="cuda"))
    # a fresh configure() + train on the same engine works and matches the oracle again
    p, p_model, _ = run_product("ns", SMALL, SMALL["batch"], dict(num_epochs=1))
    o, _, _ = run_oracle("ns", SMALL, SMALL["batch"], dict(num_epochs=1))
    lclose(p.Glosses, o.Glosses, "G after a failed run")


@pytest.mark.parametrize("ring", ["8", "5"])
def test_small_ring_wraps_many_times_and_changes_nothing(ring, monkeypatch):
    """GM_RING=8 / 5: the host / device rings wrap every few iterations (slot reuse gated on the launches
    that last staged them in, pieces never crossing the ring end) -- losses and parameters identical
    to the default 128-slot ring, bit for bit."""
    ref, ref_model, ref_rng = run_product("ns", SMALL, SMALL["batch"], dict(num_epochs=3))
    monkeypatch.setenv("GM_RING", ring)
    got, got_model, got_rng = run_product("ns", SMALL, SMALL["batch"], dict(num_epochs=3))
    assert got._engine.R == int(ring)
    assert got.Glosses == ref.Glosses and got.Dlosses == ref.Dlosses and torch.equal(ref_rng, got_rng)
    for (k, a), (_, b) in zip(got_model.state_dict().items(), ref_model.state_dict().items()):
        assert torch.equal(a, b), k


@pytest.mark.parametrize("iters", ["1", "4", "64"])
def test_graph_size_changes_nothing(iters, monkeypatch):
    """GM_GRAPH_ITERS: iterations per captured hipGraph (1 = a launch per iteration, 64 = whole epochs of the small
    configuration in one graph) -- the same kernels on the same slots: bitwise the default run."""
    ref, ref_model, ref_rng = run_product("ns", SMALL, SMALL["batch"], dict(num_epochs=3))
    monkeypatch.setenv("GM_GRAPH_ITERS", iters)
    got, got_model, got_rng = run_product("ns", SMALL, SMALL["batch"], dict(num_epochs=3))
    assert got._engine.graph_iters == int(iters)
    assert got.Glosses == ref.Glosses and got.Dlosses == ref.Dlosses and torch.equal(ref_rng, got_rng)
    for (k, a), (_, b) in zip(got_model.state_dict().items(), ref_model.state_dict().items()):
        assert torch.equal(a, b), k


def test_overridden_epoch_end_hook_sees_its_own_epochs_parameters(monkeypatch):
    """with the epochs enqueued ahead of the loss read-back, a subclass whose _end_epoch reads the model
    would see parameters of an epoch already under way -- such a subclass is not pipelined: what it records per epoch
    equals what it records with GM_PIPELINE_EPOCHS=0."""
    import ns_gan

    class Snap(ns_gan.NSGANTrainer):
        def _end_epoch(self, epoch, num_epochs, G_losses, D_losses, quiet=False):
            torch.cuda.synchronize()
            self.snaps = getattr(self, "snaps", []) + [float(self.model.D.linear.weight.double().sum())]
            super()._end_epoch(epoch, num_epochs, G_losses, D_losses, quiet)

    def run():
        loaders = port.synthetic_loaders(16, n_train=SMALL["n_train"], n_val=48, n_test=48, image_shape=SMALL["image_shape"])
        torch.manual_seed(1234)
        tr = Snap(ns_gan.NSGAN(SMALL["image_size"], SMALL["hidden_dim"], SMALL["z_dim"]), *loaders)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            tr.train(4)
        torch.cuda.synchronize()
        return tr
    a = run()
    assert a._stock() and a._engine is not None and len(a.snaps) == 4
    monkeypatch.setenv("GM_PIPELINE_EPOCHS", "0")
    b = run()
    assert a.snaps == b.snaps and a.Glosses == b.Glosses
    assert len(set(a.snaps)) == 4                      # (the parameters do move from epoch to epoch)


def test_two_train_calls_reset_adam():
    """Optimizers are locals of the reference's train(): Adam state resets per call."""
    cfg = SMALL
    loaders = port.synthetic_loaders(16, n_train=cfg["n_train"], n_val=48, n_test=48,
                                     image_shape=cfg["image_shape"])
    model = port.build("ns", cfg["image_size"], cfg["hidden_dim"], cfg["z_dim"])
    o = port.GANPort("ns", model, loaders[0])
    o.train(1); o.train(1)
    tr, pm = build_product("ns", cfg, 16)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(1); tr.train(1)
    lclose(tr.Glosses, o.Glosses, "two-call Glosses")
    lclose(tr.Dlosses, o.Dlosses, "two-call Dlosses")
    assert tr.num_epochs == 2


# largest relative loss deviation over all 50 free-running steps, measured on the MI355X (profiles/r06_parity_50step.json)
LONG_MEASURED = {"ns_full_b256_50steps": 1.9e-7, "ls_full_b1024_50steps": 3.6e-7, "ns_full_b1024_50steps": 5.2e-6}


def _oracle_min_hidden_preact(variant, meta):
    """Per D+G step of the ORACLE'S free run of a fixture's configuration: the smallest |hidden pre-activation| its
    critic forms in that step (D on x, G(z), x_hat in the critic step; D on G(z) in the generator step)."""
    cfg, steps, batch = meta["cfg"], meta["steps"], meta["batch"]
    rng = torch.get_rng_state()

    class Capped(torch.utils.data.DataLoader):
        def __len__(self):
            return steps
    ld = port.synthetic_loaders(batch, n_train=cfg["n_train"], n_val=cfg["n_val"], n_test=cfg["n_test"],
                                image_shape=tuple(cfg["image_shape"]))
    o_model = port.build(variant, cfg["image_size"], cfg["hidden_dim"], cfg["z_dim"])
    cur, per_step = [], []
    o_model.D.linear.register_forward_hook(lambda m_, i_, out: cur.append(float(out.detach().abs().min())))

    def tap(kind, tr_, info):
        if kind == "G":
            per_step.append(min(cur))
            cur.clear()
    o = port.GANPort(variant, o_model, Capped(ld[0].dataset, batch_size=batch, shuffle=True), tap=tap)
    o.train(**meta["train_kw"])
    torch.set_rng_state(rng)
    return per_step


@pytest.mark.parametrize("name", sorted(
    os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz"))
    if os.path.basename(p).split("_")[0] in ("ns", "mm", "w", "wgp", "ls", "ra", "fisher", "f",
                                             "dra", "be", "info")))
def test_engine_vs_reference_golden(name):
    """HIP path vs fixtures produced by the UNMODIFIED reference (oracle/gen_golden.py)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    cfg, variant = meta["cfg"], meta["variant"]
    full = "steps" in meta
    batch = meta["batch"] if full else cfg["batch"]
    viz_dir = None
    if meta.get("viz"):                                    # fixture written with viz=True
        import tempfile
        viz_dir = tempfile.mkdtemp()
    p_tr, p_model, p_rng = run_product(variant, cfg, batch, meta["train_kw"],
                                       capped=meta["steps"] if full else None, viz_dir=viz_dir)
    if viz_dir is not None:
        # one sample grid per epoch, written off the step stream: 6x6 images of
        # shape x shape with torchvision's 2-pixel padding
        import struct
        side = int(cfg["image_size"] ** 0.5)
        for e in range(1, meta["train_kw"]["num_epochs"] + 1):
            png = open(os.path.join(viz_dir, p_tr.name, "reconst_%d.png" % e), "rb").read()
            assert png[:8] == b"\x89PNG\r\n\x1a\n"
            w, h = struct.unpack(">II", png[16:24])
            assert (w, h) == (6 * (side + 2) + 2, 6 * (side + 2) + 2)
    # the global CPU generator ends where the reference left it (digest recorded by gen_golden)
    import hashlib
    assert hashlib.sha256(p_rng.numpy().tobytes()).hexdigest() == meta["rng"], "RNG stream position"
    if name.endswith("50steps"):
        # 50-step FREE-RUNNING curves (NSGAN B=256; round 6: WGAN-GP B=256, LSGAN / NSGAN B=1024).  Two
        # fp32 summation orders of the same math drift apart chaotically on this horizon (survey probe: 1.6e-5 between 1
        # and 8 CPU threads of the reference itself), so: north_star's 1e-5 over the first 24 steps, and over all 50 the
        # MEASURED deviation of each fixture with a 10x margin (LONG_MEASURED; floor 2e-6) -- the measured values are
        # written to gpurun_out/parity_50step.json and quoted in DESIGN.md.
        g, d = np.asarray(p_tr.Glosses), np.asarray(p_tr.Dlosses)
        eg = np.abs(g - z["Glosses"]) / np.maximum(1, np.abs(z["Glosses"]))
        ed = np.abs(d - z["Dlosses"]) / np.maximum(1, np.abs(z["Dlosses"]))
        out = os.path.join(os.path.dirname(HERE), "gpurun_out")
        if os.path.isdir(out):
            path = os.path.join(out, "parity_50step.json")
            rec = json.load(open(path)) if os.path.isfile(path) else {}
            rec[name] = {"max_rel_err_G_50": float(eg.max()), "max_rel_err_D_50": float(ed.max()),
                         "max_rel_err_G_24": float(eg[:24].max()), "max_rel_err_D_24": float(ed[:24].max()),
                         "max_rel_err_G_40": float(eg[:40].max()), "max_rel_err_D_40": float(ed[:40].max()),
                         "rel_err_G": [float("%.3g" % x) for x in eg], "rel_err_D": [float("%.3g" % x) for x in ed]}
            json.dump(rec, open(path, "w"), indent=1)
        if variant == "wgp":
            # WGAN-GP's critic loss is a DISCONTINUOUS function of the hidden pre-activations: the penalty's gradient norm
            # ||sum_u [pre_u > 0] w2_u W1[u, :]|| (w_gp_gan.py:207-214) jumps when a unit of ONE sample changes sides, and
            # the loss with it by 10 / B * delta((n - 1)^2) ~ 1e-4 -- in THAT step only; the trajectory is unaffected to
            # first order (the generator's losses stay within 4e-6 over all 50 steps).  Measured: isolated single-step
            # spikes at steps 22 / 30 / 37 (2.2e-5 / 1.5e-5 / 1.9e-5), every other step of the 50 below 1e-5.  Asserted:
            # every spike is isolated (the next step is back under 1e-5), below 1e-4, happens where the ORACLE'S OWN
            # critic has a hidden pre-activation within 2e-6 of the kink in that step (the two trajectories are ~1e-6
            # apart by then), and there are at most 5 of them; everything else holds north_star's 1e-5 over all 50 steps.
            kink = _oracle_min_hidden_preact("wgp", meta)
            spikes = [i for i in range(len(ed)) if ed[i] > TOL]
            assert eg.max() <= TOL, eg.max()
            assert len(spikes) <= 5, spikes
            for i in spikes:
                assert ed[i] <= 1e-4 and kink[i] <= 2e-6, (i, ed[i], kink[i])
                assert i + 1 >= len(ed) or ed[i + 1] <= TOL, (i, ed[i], ed[i + 1])
        else:
            assert max(eg[:24].max(), ed[:24].max()) <= TOL, (eg[:24].max(), ed[:24].max())
            assert max(eg.max(), ed.max()) <= max(10 * LONG_MEASURED[name], 2e-6), (eg.max(), ed.max())
    else:
        lclose(p_tr.Glosses, z["Glosses"], name + " Glosses")
        lclose(p_tr.Dlosses, z["Dlosses"], name + " Dlosses")
    sd = p_model.state_dict()
    if full:
        from oracle.gen_golden import digest
        for k, v in sd.items():
            np.testing.assert_allclose(digest(v), z["digest:" + k], rtol=2e-4, atol=2e-4)
    else:
        for k, v in sd.items():
            np.testing.assert_allclose(v.cpu().numpy(), z["param:" + k], rtol=0, atol=2e-5)


def test_sampling_indices_bit_exact():
    """process_batch replacement: gathered rows == the reference DataLoader's batch, bit for bit."""
    from generative_models_amd import engine, ops
    loaders = port.synthetic_loaders(32, n_train=500, n_val=48, n_test=48, image_shape=(1, 8, 8))
    ds = loaders[0].dataset.tensors[0]
    data = ds.reshape(500, -1).cuda()
    torch.manual_seed(77)
    ref_batches = [next(iter(loaders[0]))[0].view(32, -1) for _ in range(5)]
    torch.manual_seed(77)
    out = torch.empty(32, 64, device="cuda")
    idx = np.empty(32, dtype=np.int64)
    for rb in ref_batches:
        engine.draw_sampler_indices(500, 32, idx)
        ops.gather_rows(data, torch.from_numpy(idx).cuda(), out)
        assert torch.equal(out.cpu(), rb)


def test_user_override_takes_general_path():
    """README.md:33-65: NSGAN -> LSGAN by overriding train_D/train_G only."""
    import ns_gan
    from generative_models_amd.trainers import to_cuda

    class MyLS(ns_gan.NSGANTrainer):
        def train_D(self, images):
            noise = self.compute_noise(images.shape[0], self.model.z_dim)
            G_output = self.model.G(noise)
            DX, DG = self.model.D(images), self.model.D(G_output)
            return 0.5 * torch.mean((DX - 1) ** 2) + 0.5 * torch.mean((DG - 0) ** 2)

        def train_G(self, images):
            noise = self.compute_noise(images.shape[0], self.model.z_dim)
            DG = self.model.D(self.model.G(noise))
            return 0.5 * torch.mean((DG - 1) ** 2)

    cfg = SMALL
    o_tr, o_model, o_rng = run_oracle("ls", cfg, 16, dict(num_epochs=1, G_lr=2e-4, D_lr=2e-4))
    loaders = port.synthetic_loaders(16, n_train=cfg["n_train"], n_val=48, n_test=48,
                                     image_shape=cfg["image_shape"])
    torch.manual_seed(1234)
    model = ns_gan.NSGAN(cfg["image_size"], cfg["hidden_dim"], cfg["z_dim"])
    tr = MyLS(model, *loaders)
    assert not tr._stock()
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(1)
    lclose(tr.Glosses, o_tr.Glosses, "override Glosses")
    lclose(tr.Dlosses, o_tr.Dlosses, "override Dlosses")
    assert torch.equal(o_rng, torch.get_rng_state())
    # round 6: only train_D / train_G are overridden -> the hooks were replayed as a captured graph on the device
    # data path (captured.py), not run through the host DataLoader loop
    assert tr._captured is not None and tr._captured.mode == "graph" and tr._captured.done == len(tr.Glosses)
    for (k, a), (_, b) in zip(model.state_dict().items(), o_model.state_dict().items()):
        assert float((a.cpu() - b).abs().max()) <= 2e-5, k


def _readme_ls_trainer(base):
    """The README's own example (/root/reference/README.md:33-65): NSGAN -> LSGAN by overriding train_D / train_G."""
    class MyLS(base):
        def train_D(self, images):
            noise = self.compute_noise(images.shape[0], self.model.z_dim)
            G_output = self.model.G(noise)
            DX_score, DG_score = self.model.D(images), self.model.D(G_output)
            return (0.50 * torch.mean((DX_score - 1.) ** 2)) + (0.50 * torch.mean((DG_score - 0.) ** 2))

        def train_G(self, images):
            noise = self.compute_noise(images.shape[0], self.model.z_dim)
            DG_score = self.model.D(self.model.G(noise))
            return 0.50 * torch.mean((DG_score - 1.) ** 2)
    return MyLS


@pytest.mark.parametrize("batch,D_steps", [(256, 1), (100, 2)])
def test_readme_override_full_size_captured_vs_oracle(batch, D_steps):
    """README.md:29-65 at 784-400-20: a user subclass that overrides ONLY train_D / train_G (the README's LSGAN edit of
    NSGANTrainer) runs as a captured graph on the device data path and must reproduce the oracle's LSGAN run (same
    learning rates as the NSGAN trainer it subclasses): losses 1e-5, parameters 1e-5, generator state bit-exact."""
    import ns_gan
    steps = 12
    kw = dict(num_epochs=2, G_lr=2e-4, D_lr=2e-4, D_steps=D_steps)
    ld = _capped_loaders(batch, steps // 2 * D_steps)
    o_model = port.build("ls", 784, 400, 20)
    o = port.GANPort("ls", o_model, ld[0])
    o.train(**kw)
    o_rng = torch.get_rng_state()
    ld = _capped_loaders(batch, steps // 2 * D_steps)
    torch.manual_seed(1234)
    model = ns_gan.NSGAN(784, 400, 20)
    tr = _readme_ls_trainer(ns_gan.NSGANTrainer)(model, *ld)
    assert not tr._stock() and tr._captured_general_ok()
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(2, D_steps=D_steps)
    torch.cuda.synchronize()
    assert tr._captured.mode == "graph" and len(tr.Glosses) == steps and tr.num_epochs == 2
    lclose(tr.Glosses, o.Glosses, "README override Glosses")
    lclose(tr.Dlosses, o.Dlosses, "README override Dlosses")
    assert torch.equal(o_rng, torch.get_rng_state())
    dev = _param_dev(model, o_model, 2e-4)
    _record("readme_override_captured[b%d_d%d]" % (batch, D_steps), steps=steps, Dloss_err=_loss_err(tr.Dlosses, o.Dlosses),
            Gloss_err=_loss_err(tr.Glosses, o.Glosses), params=dev)
    for k, v in dev.items():
        assert v["max"] <= 1e-5, (k, v)


def test_captured_general_path_equals_the_host_loop(monkeypatch):
    """The captured loop against the fully general loop (GM_CAPTURED_GENERAL=0: host DataLoader, CPU randn, .item() per
    step) on the same user subclass: same draws, same kernels for the GEMMs -> the same losses and parameters to
    rounding (the host loop also back-propagates the gradients the reference throws away; they do not reach a result)."""
    import ns_gan

    def run():
        loaders = port.synthetic_loaders(32, n_train=SMALL["n_train"], n_val=48, n_test=48, image_shape=SMALL["image_shape"])
        torch.manual_seed(1234)
        model = ns_gan.NSGAN(SMALL["image_size"], SMALL["hidden_dim"], SMALL["z_dim"])
        tr = _readme_ls_trainer(ns_gan.NSGANTrainer)(model, *loaders)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            tr.train(3, D_steps=2)
        torch.cuda.synchronize()
        return tr, model, torch.get_rng_state()
    a = run()
    assert a[0]._captured.mode == "graph"
    monkeypatch.setenv("GM_CAPTURED_GENERAL", "0")
    b = run()
    assert getattr(b[0], "_captured", None) is None
    lclose(a[0].Glosses, b[0].Glosses, "captured vs host loop G", tol=2e-6)
    lclose(a[0].Dlosses, b[0].Dlosses, "captured vs host loop D", tol=2e-6)
    assert torch.equal(a[2], b[2])
    for (k, x), (_, y) in zip(a[1].state_dict().items(), b[1].state_dict().items()):
        assert float((x - y).abs().max()) <= 2e-5, k


def test_captured_general_path_refuses_hooks_that_draw_themselves(monkeypatch):
    """A train_D that consumes the global CPU generator itself (WGAN-GP style torch.rand) cannot be pre-drawn by the host
    replay: the first iteration's transaction detects it, restores parameters / optimizer / generator state, and the
    run takes the host loop -- bitwise the run with the captured path switched off."""
    import ns_gan
    from generative_models_amd.trainers import to_cuda

    class Noisy(_readme_ls_trainer(ns_gan.NSGANTrainer)):
        def train_D(self, images):
            jitter = to_cuda(torch.rand(images.shape[0], 1))             # an extra draw between the stock ones
            noise = self.compute_noise(images.shape[0], self.model.z_dim)
            DX, DG = self.model.D(images), self.model.D(self.model.G(noise))
            return 0.5 * torch.mean((DX - 1 + 0.01 * jitter) ** 2) + 0.5 * torch.mean(DG ** 2)

    def run():
        loaders = port.synthetic_loaders(16, n_train=SMALL["n_train"], n_val=48, n_test=48, image_shape=SMALL["image_shape"])
        torch.manual_seed(1234)
        model = ns_gan.NSGAN(SMALL["image_size"], SMALL["hidden_dim"], SMALL["z_dim"])
        tr = Noisy(model, *loaders)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            tr.train(2)
        torch.cuda.synchronize()
        return tr, model, torch.get_rng_state()
    a = run()
    assert a[0]._captured.done == 0 and "generator" in a[0]._captured_refused
    monkeypatch.setenv("GM_CAPTURED_GENERAL", "0")
    b = run()
    assert a[0].Glosses == b[0].Glosses and a[0].Dlosses == b[0].Dlosses and torch.equal(a[2], b[2])
    for (k, x), (_, y) in zip(a[1].state_dict().items(), b[1].state_dict().items()):
        assert torch.equal(x, y), k


def test_captured_general_path_hooks_that_cannot_be_captured_run_eagerly_on_the_device_path():
    """A hook that reads a value back (.item()) cannot be captured: the iterations then run eagerly, still on the
    device data path (prefetched rings, device gather, flat Adam) -- same results as the captured replay of the same
    arithmetic."""
    import ns_gan
    LS = _readme_ls_trainer(ns_gan.NSGANTrainer)

    class Peeking(LS):
        def train_G(self, images):
            loss = LS.train_G(self, images)
            self.last_G = loss.item()                                   # a host read: not capturable
            return loss

    def run(cls):
        loaders = port.synthetic_loaders(16, n_train=SMALL["n_train"], n_val=48, n_test=48, image_shape=SMALL["image_shape"])
        torch.manual_seed(1234)
        model = ns_gan.NSGAN(SMALL["image_size"], SMALL["hidden_dim"], SMALL["z_dim"])
        tr = cls(model, *loaders)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            tr.train(2)
        torch.cuda.synchronize()
        return tr, model, torch.get_rng_state()
    a, b = run(Peeking), run(LS)
    assert a[0]._captured.mode == "eager" and a[0]._captured.done == len(a[0].Glosses) and b[0]._captured.mode == "graph"
    assert a[0].Glosses == b[0].Glosses and a[0].Dlosses == b[0].Dlosses and torch.equal(a[2], b[2])
    assert a[0].last_G == a[0].Glosses[-1]
    for (k, x), (_, y) in zip(a[1].state_dict().items(), b[1].state_dict().items()):
        assert torch.equal(x, y), k


# ---------------------------------------------------------------------------------------------
# VAE (config 4): fused engine vs oracle and vs reference fixtures, incl. the ragged last batch
# ---------------------------------------------------------------------------------------------
def run_vae_product(cfg, batch, n_train, epochs, use_graph=True):
    import vae
    loaders = port.synthetic_loaders(batch, n_train=n_train, n_val=cfg["n_val"],
                                     n_test=cfg["n_test"], image_shape=tuple(cfg["image_shape"]))
    torch.manual_seed(1234)
    model = vae.VAE(image_size=cfg["image_size"], hidden_dim=cfg["hidden_dim"], z_dim=cfg["z_dim"])
    tr = vae.VAETrainer(model, *loaders, viz=False)
    tr.use_graph = use_graph
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(num_epochs=epochs)
    torch.cuda.synchronize()
    return tr, model, torch.get_rng_state()


@pytest.mark.parametrize("cfg,n_train", [(SMALL, 160), (RAGGED, 200), (SMALL, 150)],
                         ids=["small", "ragged", "partial-batch"])
def test_vae_engine_vs_oracle(cfg, n_train):
    loaders = port.synthetic_loaders(cfg["batch"], n_train=n_train, n_val=cfg["n_val"],
                                     n_test=cfg["n_test"], image_shape=tuple(cfg["image_shape"]))
    o_model = port.build("vae", cfg["image_size"], cfg["hidden_dim"], cfg["z_dim"])
    o = port.VAEPort(o_model, *loaders)
    o.train(2)
    o_rng = torch.get_rng_state()
    p, p_model, p_rng = run_vae_product(cfg, cfg["batch"], n_train, 2)
    lclose(np.array(p.recon_loss) / 100, np.array(o.recon_loss) / 100, "vae recon")   # sums ~1e3
    lclose(p.kl_loss, o.kl_loss, "vae kl", tol=1e-5)
    assert abs(p.best_val_loss - o.best_val_loss) <= 1e-5 * max(1, abs(o.best_val_loss))
    assert torch.equal(o_rng, p_rng)
    for (k, a), (_, b) in zip(p_model.state_dict().items(), o_model.state_dict().items()):
        assert (a.cpu() - b).abs().max().item() <= 5e-5, k


@pytest.mark.parametrize("name", ["vae_small", "vae_full_b512", "vae_full_b512_ragged", "vae_full_b512_epoch98"])
def test_vae_engine_vs_reference_golden(name):
    """vae_full_b512_ragged: 3 full batches of 512 + the ragged 336 (= 50 000 mod 512), 2 epochs.
    vae_full_b512_epoch98 (round 6): ONE WHOLE EPOCH of the unmodified reference over 50 000 images -- 97 batches of 512,
    the ragged 336 and the validation pass (vae.py:127-191): the epoch bench.py times, every batch's losses at 1e-5."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    cfg = meta["cfg"]
    batch = meta.get("batch", cfg.get("batch"))
    p, p_model, p_rng = run_vae_product(cfg, batch, meta.get("n_train", cfg["n_train"]),
                                        meta["train_kw"]["num_epochs"])
    import hashlib
    assert hashlib.sha256(p_rng.numpy().tobytes()).hexdigest() == meta["rng"], "RNG stream position"
    ref = z["recon_loss"]
    got = np.array(p.recon_loss)
    assert np.max(np.abs(got - ref) / np.maximum(1, np.abs(ref))) <= 1e-5, (got[:4], ref[:4])
    ref, got = z["kl_loss"], np.array(p.kl_loss)
    assert np.max(np.abs(got - ref) / np.maximum(1, np.abs(ref))) <= 1e-5, (got[:4], ref[:4])
    assert abs(p.best_val_loss - float(z["best_val_loss"])) <= 1e-5 * abs(float(z["best_val_loss"]))
    if name.endswith("epoch98"):
        from oracle.gen_golden import digest
        _record("vae_epoch98_vs_reference", recon_err=_loss_err(p.recon_loss, z["recon_loss"]),
                kl_err=_loss_err(p.kl_loss, z["kl_loss"]),
                best_val_rel_err=abs(p.best_val_loss - float(z["best_val_loss"])) / abs(float(z["best_val_loss"])))
        for k, v in p_model.state_dict().items():
            np.testing.assert_allclose(digest(v), z["digest:" + k], rtol=2e-4, atol=2e-4)


# ---------------------------------------------------------------------------------------------
# BIR-VAE (bir_vae.py; SURVEY.md 8f item 2, second half): numpy + torch global generators
# ---------------------------------------------------------------------------------------------
def run_bir_product(cfg, batch, n_train, epochs, use_graph=True, np_seed=77):
    import bir_vae
    loaders = port.synthetic_loaders(batch, n_train=n_train, n_val=cfg["n_val"],
                                     n_test=cfg["n_test"], image_shape=tuple(cfg["image_shape"]))
    torch.manual_seed(1234)
    np.random.seed(np_seed)
    model = bir_vae.BIRVAE(image_size=cfg["image_size"], hidden_dim=cfg["hidden_dim"],
                           z_dim=cfg["z_dim"])
    tr = bir_vae.BIRVAETrainer(model, *loaders, viz=False)
    tr.use_graph = use_graph
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(num_epochs=epochs)
    torch.cuda.synchronize()
    assert tr._engine is not None, "fused BIR-VAE engine was not used"
    return tr, model, torch.get_rng_state()


@pytest.mark.parametrize("name", ["bir_small", "bir_full_b256"])
def test_bir_vae_engine_vs_reference_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    cfg = meta["cfg"]
    p, p_model, p_rng = run_bir_product(cfg, meta["batch"], meta["n_train"],
                                        meta["train_kw"]["num_epochs"], np_seed=meta["np_seed"])
    import hashlib
    assert hashlib.sha256(p_rng.numpy().tobytes()).hexdigest() == meta["rng"], "RNG stream position"
    ref, got = z["recon_loss"], np.array(p.recon_loss)
    assert np.max(np.abs(got - ref) / np.maximum(1, np.abs(ref))) <= 1e-5, (got[:4], ref[:4])
    # 1000 * (three O(B^2) sums that nearly cancel): the reference's own fp32 pairwise sums carry
    # ~1e-7 * B^2 * 1000 of rounding; same bound as the oracle's test in test_golden.py
    ref, got = z["mmd_loss"], np.array(p.mmd_loss)
    assert np.max(np.abs(got - ref) / np.maximum(10, np.abs(ref))) <= 1e-4, (got[:4], ref[:4])
    # (the validation loss CONTAINS the 1000 * MMD term bounded just above: 2e-5, not the VAE's 1e-5)
    assert abs(p.best_val_loss - float(z["best_val_loss"])) <= 2e-5 * abs(float(z["best_val_loss"]))
    for k, v in p_model.state_dict().items():
        if "param:" + k in z:
            assert np.abs(v.cpu().numpy() - z["param:" + k]).max() <= 5e-5, k


def test_bir_vae_engine_vs_oracle_eager_and_graph():
    cfg = SMALL
    loaders = port.synthetic_loaders(cfg["batch"], n_train=150, n_val=cfg["n_val"],
                                     n_test=cfg["n_test"], image_shape=tuple(cfg["image_shape"]))
    np.random.seed(5)
    o_model = port.build("bir", cfg["image_size"], cfg["hidden_dim"], cfg["z_dim"])
    o = port.BIRVAEPort(o_model, *loaders)
    o.train(2)
    o_rng, o_np = torch.get_rng_state(), np.random.get_state()[1].copy()
    for use_graph in (True, False):
        p, p_model, p_rng = run_bir_product(cfg, cfg["batch"], 150, 2, use_graph=use_graph, np_seed=5)
        assert torch.equal(o_rng, p_rng)
        assert np.array_equal(o_np, np.random.get_state()[1]), "numpy generator position"
        lclose(np.array(p.recon_loss) / 100, np.array(o.recon_loss) / 100, "bir recon")
        ref, got = np.array(o.mmd_loss), np.array(p.mmd_loss)
        assert np.max(np.abs(got - ref) / np.maximum(10, np.abs(ref))) <= 1e-4
        for (k, a), (_, b) in zip(p_model.state_dict().items(), o_model.state_dict().items()):
            assert (a.cpu() - b).abs().max().item() <= 5e-5, k


def test_bir_vae_checkpoint_resume_is_bitwise_uninterrupted(tmp_path):
    """train(1) + save + load into a fresh trainer + train(1) == train(2), bit for bit: weights, Adam
    moments and schedule position, torch's AND numpy's generator cursors all travel."""
    import bir_vae
    import contextlib, io
    cfg = SMALL

    def build(seed, np_seed):
        loaders = port.synthetic_loaders(cfg["batch"], n_train=150, n_val=cfg["n_val"], n_test=cfg["n_test"],
                                         image_shape=tuple(cfg["image_shape"]))
        torch.manual_seed(seed)
        np.random.seed(np_seed)
        model = bir_vae.BIRVAE(image_size=cfg["image_size"], hidden_dim=cfg["hidden_dim"], z_dim=cfg["z_dim"])
        return bir_vae.BIRVAETrainer(model, *loaders, viz=False), model

    def train(tr, n):
        with contextlib.redirect_stdout(io.StringIO()):
            tr.train(num_epochs=n)
        torch.cuda.synchronize()

    full, full_model = build(1234, 9)
    train(full, 2)
    full_rng, full_np = torch.get_rng_state(), np.random.get_state()[1].copy()
    tr1, _ = build(1234, 9)
    train(tr1, 1)
    path = str(tmp_path / "bir.pt")
    tr1.save_checkpoint(path)
    tr2, model2 = build(999, 31337)                    # different init and generators: all overwritten
    torch.manual_seed(4242)
    np.random.seed(4242)
    tr2.load_checkpoint(path)
    train(tr2, 1)
    assert torch.equal(torch.get_rng_state(), full_rng)
    assert np.array_equal(np.random.get_state()[1], full_np)
    assert tr2.recon_loss == full.recon_loss and tr2.mmd_loss == full.mmd_loss
    for (k, a), (_, b) in zip(model2.state_dict().items(), full_model.state_dict().items()):
        assert torch.equal(a, b), k


@pytest.mark.parametrize("name", ["vae_small_viz", "bir_small_viz"])
def test_vae_family_viz_stream_position_vs_reference_golden(name, tmp_path):
    """viz=True (vae.py:189-191, bir_vae.py:176-178): sample_images(epoch) draws torch.randn(36, z) at
    every epoch end; losses, parameters and the final generator digest against the UNMODIFIED reference
    run with viz on; the PNG grids are written without torchvision / PIL."""
    import contextlib, hashlib, io
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    cfg = meta["cfg"]
    loaders = port.synthetic_loaders(cfg["batch"], n_train=meta["n_train"], n_val=cfg["n_val"],
                                     n_test=cfg["n_test"], image_shape=tuple(cfg["image_shape"]))
    torch.manual_seed(1234)
    np.random.seed(meta["np_seed"])
    if meta["variant"] == "vae":
        import vae
        model = vae.VAE(image_size=cfg["image_size"], hidden_dim=cfg["hidden_dim"], z_dim=cfg["z_dim"])
        tr = vae.VAETrainer(model, *loaders, viz=True)
        second = "kl_loss"
    else:
        import bir_vae
        model = bir_vae.BIRVAE(image_size=cfg["image_size"], hidden_dim=cfg["hidden_dim"], z_dim=cfg["z_dim"])
        tr = bir_vae.BIRVAETrainer(model, *loaders, viz=True)
        second = "mmd_loss"
    tr.viz_dir = str(tmp_path)
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(**meta["train_kw"])
    torch.cuda.synchronize()
    assert tr._engine is not None
    assert hashlib.sha256(torch.get_rng_state().numpy().tobytes()).hexdigest() == meta["rng"]
    ref, got = z["recon_loss"], np.array(tr.recon_loss)
    assert np.max(np.abs(got - ref) / np.maximum(1, np.abs(ref))) <= 1e-5
    ref, got = z[second], np.array(getattr(tr, second))
    tol = 1e-5 if second == "kl_loss" else 1e-4
    assert np.max(np.abs(got - ref) / np.maximum(10 if second == "mmd_loss" else 1, np.abs(ref))) <= tol
    for k, v in model.state_dict().items():
        assert np.abs(v.cpu().numpy() - z["param:" + k]).max() <= 5e-5, k
    for e in (1, 2, 3):
        p = os.path.join(str(tmp_path), tr.name, "sample_%d.png" % e)
        assert os.path.getsize(p) > 100 and open(p, "rb").read(8) == b"\x89PNG\r\n\x1a\n"
    rec = tr.reconstruct_images(tr.debugging_image, 7)
    assert rec.shape[0] == tr.debugging_image.shape[0]
    assert os.path.isfile(os.path.join(str(tmp_path), tr.name, "reconst_7.png"))
    assert len(tr.sample_interpolated_images()) == cfg["z_dim"]


def test_infogan_viz_epoch_and_latent_exploration(tmp_path):
    """info_gan.py:306-365: generate_images builds its noise with the 4-argument compute_noise; c fixes
    the categorical code of every sample.  viz=True must run through an epoch."""
    import contextlib, io
    import info_gan
    cfg = SMALL
    loaders = port.synthetic_loaders(cfg["batch"], n_train=64, n_val=cfg["n_val"], n_test=cfg["n_test"],
                                     image_shape=tuple(cfg["image_shape"]))
    torch.manual_seed(5)
    model = info_gan.InfoGAN(image_size=cfg["image_size"], hidden_dim=cfg["hidden_dim"], z_dim=cfg["z_dim"],
                             disc_dim=10, cont_dim=10)
    tr = info_gan.InfoGANTrainer(model, *loaders, viz=True)
    tr.viz_dir = str(tmp_path)
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(num_epochs=1)
    assert os.path.isfile(os.path.join(str(tmp_path), tr.name, "reconst_1.png"))
    s0 = torch.get_rng_state()
    a = tr.generate_images(9, num_outputs=16, save=False, c=3)
    torch.set_rng_state(s0)
    b = tr.generate_images(9, num_outputs=16, save=False, c=3)
    assert a.shape == (16, model.shape, model.shape) and np.array_equal(a, b)
    torch.set_rng_state(s0)
    c = tr.generate_images(9, num_outputs=16, save=False, c=7)         # same z / c2 draws, other category
    assert not np.array_equal(a, c)


def test_bir_vae_user_hook_takes_general_path():
    import bir_vae
    cfg = SMALL
    loaders = port.synthetic_loaders(cfg["batch"], n_train=96, n_val=cfg["n_val"],
                                     n_test=cfg["n_test"], image_shape=tuple(cfg["image_shape"]))

    class Mine(bir_vae.BIRVAETrainer):
        def compute_batch(self, batch, LAMBDA=10.):
            return super().compute_batch(batch, LAMBDA=LAMBDA)

    torch.manual_seed(3)
    np.random.seed(3)
    model = bir_vae.BIRVAE(image_size=cfg["image_size"], hidden_dim=cfg["hidden_dim"], z_dim=cfg["z_dim"])
    tr = Mine(model, *loaders)
    tr.train(1, quiet=True)
    assert tr._engine is None and len(tr.recon_loss) == len(loaders[0])
    assert np.isfinite(tr.best_val_loss)


# ---------------------------------------------------------------------------------------------
# Autoencoder (ae.py) on the VAE engine's machinery
# ---------------------------------------------------------------------------------------------
def run_ae_product(cfg, hidden, batch, n_train, epochs, use_graph=True):
    import ae
    loaders = port.synthetic_loaders(batch, n_train=n_train, n_val=cfg["n_val"],
                                     n_test=cfg["n_test"], image_shape=tuple(cfg["image_shape"]))
    torch.manual_seed(1234)
    model = ae.Autoencoder(image_size=cfg["image_size"], hidden_dim=hidden)
    tr = ae.AutoencoderTrainer(model, *loaders, viz=False)
    tr.use_graph = use_graph
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(num_epochs=epochs)
    torch.cuda.synchronize()
    return tr, model, torch.get_rng_state()


@pytest.mark.parametrize("cfg,n_train,use_graph", [(SMALL, 160, True), (RAGGED, 200, True),
                                                   (SMALL, 150, True), (SMALL, 150, False)],
                         ids=["small", "ragged", "partial-batch", "eager"])
def test_ae_engine_vs_oracle(cfg, n_train, use_graph):
    hidden = cfg["z_dim"]
    loaders = port.synthetic_loaders(cfg["batch"], n_train=n_train, n_val=cfg["n_val"],
                                     n_test=cfg["n_test"], image_shape=tuple(cfg["image_shape"]))
    o_model = port.build("ae", cfg["image_size"], hidden)
    o = port.AEPort(o_model, *loaders)
    o.train(2)
    o_rng = torch.get_rng_state()
    p, p_model, p_rng = run_ae_product(cfg, hidden, cfg["batch"], n_train, 2, use_graph)
    assert p._engine is not None                      # the fused engine ran, not the general path
    lclose(np.array(p.recon_loss) / 100, np.array(o.recon_loss) / 100, "ae recon")   # sums ~1e3
    assert abs(p.best_val_loss - o.best_val_loss) <= 1e-5 * max(1, abs(o.best_val_loss))
    assert torch.equal(o_rng, p_rng)
    for (k, a), (_, b) in zip(p_model.state_dict().items(), o_model.state_dict().items()):
        assert (a.cpu() - b).abs().max().item() <= 5e-5, k


@pytest.mark.parametrize("name", ["ae_small", "ae_full_b512"])
def test_ae_engine_vs_reference_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    cfg = meta["cfg"]
    p, p_model, _ = run_ae_product(cfg, meta["hidden"], meta["batch"], meta["n_train"],
                                   meta["train_kw"]["num_epochs"])
    ref, got = z["recon_loss"], np.array(p.recon_loss)
    assert np.max(np.abs(got - ref) / np.maximum(1, np.abs(ref))) <= 1e-5, (got[:4], ref[:4])
    assert abs(p.best_val_loss - float(z["best_val_loss"])) <= 1e-5 * abs(float(z["best_val_loss"]))
    for k, v in p_model.state_dict().items():
        if "param:" + k in z:
            assert np.abs(v.cpu().numpy() - z["param:" + k]).max() <= 5e-5, k


def test_ae_general_path_when_hook_overridden():
    """A user subclass overriding compute_batch runs the reference loop over the HIP autograd
    Functions + flat HIP Adam and still matches the oracle."""
    import ae
    cfg = SMALL
    loaders = port.synthetic_loaders(cfg["batch"], n_train=96, n_val=cfg["n_val"],
                                     n_test=cfg["n_test"], image_shape=tuple(cfg["image_shape"]))
    o_model = port.build("ae", cfg["image_size"], cfg["z_dim"])
    o = port.AEPort(o_model, *loaders)
    o.train(1)

    class Mine(ae.AutoencoderTrainer):
        def compute_batch(self, batch):
            return super().compute_batch(batch)

    loaders = port.synthetic_loaders(cfg["batch"], n_train=96, n_val=cfg["n_val"],
                                     n_test=cfg["n_test"], image_shape=tuple(cfg["image_shape"]))
    torch.manual_seed(1234)
    model = ae.Autoencoder(image_size=cfg["image_size"], hidden_dim=cfg["z_dim"])
    tr = Mine(model, *loaders, viz=False)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(num_epochs=1)
    assert tr._engine is None
    lclose(np.array(tr.recon_loss) / 100, np.array(o.recon_loss) / 100, "ae general recon")
    for (k, a), (_, b) in zip(model.state_dict().items(), o_model.state_dict().items()):
        assert (a.cpu() - b).abs().max().item() <= 5e-5, k


# ---------------------------------------------------------------------------------------------
# Checkpoint / resume: train(1) + save + load into a FRESH trainer + train(1)
# must equal one uninterrupted train(2) bit for bit -- weights, Adam moments and schedule position,
# the RNG-protocol cursor and the loss history all travel in the checkpoint.
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant,kw", [("ns", {}), ("w", dict(D_steps=2)), ("wgp", dict(D_steps=1)),
                                        ("fisher", {}), ("info", {}), ("be", {}), ("dra", dict(D_steps=1))],
                         ids=["ns", "w", "wgp", "fisher", "info", "be", "dra"])
def test_gan_checkpoint_resume_is_bitwise_uninterrupted(variant, kw, tmp_path):
    import contextlib, io
    full, full_model, full_rng = run_product(variant, SMALL, 16, dict(num_epochs=2, **kw))
    tr1, _ = build_product(variant, SMALL, 16)
    with contextlib.redirect_stdout(io.StringIO()):
        tr1.train(num_epochs=1, **kw)
    path = str(tmp_path / "ck.pt")
    tr1.save_checkpoint(path)
    torch.manual_seed(999)                                 # a different init and RNG position
    tr2, model2 = build_product(variant, SMALL, 16)
    with torch.no_grad():
        for p_ in model2.parameters():                     # nothing of the fresh init may survive
            p_.add_(0.25)
    torch.manual_seed(4242)
    tr2.load_checkpoint(path)
    with contextlib.redirect_stdout(io.StringIO()):
        tr2.train(num_epochs=1, **kw)
    torch.cuda.synchronize()
    assert tr2.num_epochs == 2
    assert tr2.Glosses == full.Glosses and tr2.Dlosses == full.Dlosses
    if variant == "info":                                  # third optimizer's moments travelled
        assert tr2.MIlosses == full.MIlosses
    if variant == "be":                                    # K controller + plateau schedulers travelled
        assert tr2.K == full.K
    assert torch.equal(torch.get_rng_state(), full_rng)
    for (k, a), (_, b) in zip(model2.state_dict().items(), full_model.state_dict().items()):
        assert torch.equal(a, b), k
    # the reference's own weights-only checkpoint still loads
    tr2.save_model(str(tmp_path / "w.pt"))
    assert list(torch.load(str(tmp_path / "w.pt")).keys()) == list(full_model.state_dict().keys())


def test_checkpoint_refuses_a_run_with_other_settings(tmp_path):
    """nothing checked D_steps / lr / batch size against the saved run."""
    import contextlib, io
    from generative_models_amd._lib import GMError
    tr1, _ = build_product("w", SMALL, 16)
    with contextlib.redirect_stdout(io.StringIO()):
        tr1.train(num_epochs=1, D_steps=2)
    path = str(tmp_path / "ck.pt")
    tr1.save_checkpoint(path)
    ck = torch.load(path, weights_only=True)               # plain tensors / numbers only
    assert all(isinstance(x, float) for x in ck["history"]["Glosses"])
    tr2, _ = build_product("w", SMALL, 16)
    tr2.load_checkpoint(path)
    with pytest.raises(GMError, match="different settings"):
        with contextlib.redirect_stdout(io.StringIO()):
            tr2.train(num_epochs=1, D_steps=3)
    tr3, _ = build_product("w", SMALL, 16)
    tr3.load_checkpoint(path, strict=False)
    with contextlib.redirect_stdout(io.StringIO()):
        tr3.train(num_epochs=1, D_steps=3)                 # explicit override: allowed


@pytest.mark.parametrize("kind", ["vae", "ae"])
def test_vae_ae_checkpoint_resume_is_bitwise_uninterrupted(kind, tmp_path):
    import contextlib, io
    cfg = SMALL

    def build(seed):
        loaders = port.synthetic_loaders(cfg["batch"], n_train=150, n_val=cfg["n_val"],
                                         n_test=cfg["n_test"], image_shape=tuple(cfg["image_shape"]))
        torch.manual_seed(seed)
        if kind == "vae":
            import vae
            model = vae.VAE(image_size=cfg["image_size"], hidden_dim=cfg["hidden_dim"],
                            z_dim=cfg["z_dim"])
            return vae.VAETrainer(model, *loaders, viz=False), model
        import ae
        model = ae.Autoencoder(image_size=cfg["image_size"], hidden_dim=cfg["z_dim"])
        return ae.AutoencoderTrainer(model, *loaders, viz=False), model

    def train(tr, n):
        with contextlib.redirect_stdout(io.StringIO()):
            tr.train(num_epochs=n)
        torch.cuda.synchronize()

    full, full_model = build(1234)
    train(full, 2)
    full_rng = torch.get_rng_state()
    tr1, _ = build(1234)
    train(tr1, 1)
    path = str(tmp_path / "ck.pt")
    tr1.save_checkpoint(path)
    tr2, model2 = build(999)
    torch.manual_seed(4242)
    tr2.load_checkpoint(path)
    train(tr2, 1)
    assert tr2.num_epochs == 2 and tr2.recon_loss == full.recon_loss
    if kind == "vae":
        assert tr2.kl_loss == full.kl_loss
    assert tr2.best_val_loss == full.best_val_loss
    assert torch.equal(torch.get_rng_state(), full_rng)
    for (k, a), (_, b) in zip(model2.state_dict().items(), full_model.state_dict().items()):
        assert torch.equal(a, b), k


@pytest.mark.parametrize("in_graph", [True, False], ids=["rccl_in_graph", "host_launched"])
def test_dp_launch_structure_single_rank_rccl(in_graph, monkeypatch):
    """The RCCL fallback of the data-parallel step on a 1-rank group: must equal the single-graph run (an all-reduce
    over one rank is the identity).  in_graph (round 5): the all-reduces are nodes of the iteration's ONE hipGraph
    (gm_rccl_allreduce_f32 on the library's own communicator); host_launched (GM_RCCL_IN_GRAPH=0): round 1's hipGraph
    per segment with torch.distributed all-reduces between them."""
    import socket
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port_no = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port_no)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    monkeypatch.setenv("GM_DP_COMM", "rccl")
    monkeypatch.setenv("GM_RCCL_IN_GRAPH", "1" if in_graph else "0")
    try:
        monkeypatch.delenv("GM_DP_COMM")
        ref = run_product("ls", SMALL, 16, dict(num_epochs=2))
        monkeypatch.setenv("GM_DP_COMM", "rccl")
        tr, model = build_product("ls", SMALL, 16)
        tr.force_dp = True
        eng = tr._get_engine()
        assert eng.force_segments and eng.comm_mode == "rccl"
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            tr.train(num_epochs=2)
        torch.cuda.synchronize()
        if in_graph:
            assert eng.exchange_form() == "rccl_in_graph" and eng.seg_graphs is None and eng._one_graph()
        else:
            assert eng.exchange_form() == "rccl" and eng.seg_graphs is not None and len(eng.seg_graphs) == 3
        # same kernels except Adam (separate launch here, gradient-epilogue fusion in the
        # single-graph run): identical up to fp32 contraction differences
        lclose(tr.Glosses, ref[0].Glosses, "dp-structure Glosses", tol=1e-6)
        lclose(tr.Dlosses, ref[0].Dlosses, "dp-structure Dlosses", tol=1e-6)
        for (k, a), (_, b) in zip(model.state_dict().items(), ref[1].state_dict().items()):
            assert (a - b).abs().max().item() <= 1e-6, k
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("variant,kw", [("ls", dict(num_epochs=2)), ("w", dict(num_epochs=1, D_steps=2)),
                                        ("wgp", dict(num_epochs=1, D_steps=1))], ids=["ls", "w", "wgp"])
def test_dp_peer_structure_single_rank(variant, kw):
    """The data-parallel launch structure with the in-graph peer exchange (gradient buckets inside
    the exchange region, reduce + gather(+Adam) kernels in the ONE hipGraph per iteration) on a
    single rank: must equal the single-GPU fused run (a one-rank sum is the identity; Adam runs in
    the gather kernel instead of the gradient epilogues)."""
    ref = run_product(variant, SMALL, 16, kw)
    tr, model = build_product(variant, SMALL, 16)
    tr.force_dp = True
    eng = tr._get_engine()
    assert eng._peer() and eng._comms is not None and eng.fD.grad.data_ptr() != 0
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(**kw)
    torch.cuda.synchronize()
    lclose(tr.Glosses, ref[0].Glosses, "peer-structure Glosses", tol=1e-6)
    lclose(tr.Dlosses, ref[0].Dlosses, "peer-structure Dlosses", tol=1e-6)
    for (k, a), (_, b) in zip(model.state_dict().items(), ref[1].state_dict().items()):
        assert (a - b).abs().max().item() <= 1e-6, k


# ---------------------------------------------------------------------------------------------
# Full-size (784-400-20, bs=256, N=50000) fused engine vs the CPU oracle for the variants without a
# full-size reference fixture: exercises the aligned 16-byte / quad-transpose kernel paths and the
# multi-batch chunk schedules that the small configurations do not reach.
# ---------------------------------------------------------------------------------------------
FULLCFG = dict(image_size=784, hidden_dim=400, z_dim=20, n_train=50000, n_val=256, n_test=256,
               image_shape=(1, 28, 28))
FULL_CASES = [("dra", dict(num_epochs=1, D_steps=1)), ("be", dict(num_epochs=1)),
              ("info", dict(num_epochs=1)), ("ra", dict(num_epochs=1)), ("fisher", dict(num_epochs=1)),
              ("mm", dict(num_epochs=1, G_init=2)), ("w", dict(num_epochs=1, D_steps=2)),
              ("f", dict(num_epochs=1, method="pearson")),
              # round 4: the step counts the bench reports and the trainers default to (w_gp_gan.py:96,
              # dra_gan.py:94, w_gan.py: D_steps = 5) and the other five f-divergences (f_gan.py:99-142)
              ("wgp", dict(num_epochs=1, D_steps=5)), ("dra", dict(num_epochs=1, D_steps=5)),
              ("w", dict(num_epochs=1, D_steps=5)),
              ("f", dict(num_epochs=1, method="total_variation")), ("f", dict(num_epochs=1, method="forward_kl")),
              ("f", dict(num_epochs=1, method="reverse_kl")), ("f", dict(num_epochs=1, method="hellinger")),
              ("f", dict(num_epochs=1, method="jensen_shannon"))]


def _case_id(v, kw):
    return v + ("_" + kw["method"] if "method" in kw else "") + ("_d%d" % kw["D_steps"] if kw.get("D_steps", 1) > 2 else "")


def _record(name, **vals):
    """Measured deviations of the full-size parity tests, appended to gpurun_out/parity_full_size.jsonl (copied to
    profiles/ per round: the asserted bounds say what is tolerated, this says what was seen)."""
    import json
    try:
        d = os.path.join(os.path.dirname(HERE), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_full_size.jsonl"), "a") as f:
            f.write(json.dumps(dict(test=name, **vals)) + "\n")
    except OSError:
        pass


def _loss_err(got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float((np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max())


def _param_dev(model, o_model, lr):
    out = {}
    for (k, a), (_, b) in zip(model.state_dict().items(), o_model.state_dict().items()):
        d = (a.cpu() - b).abs()
        out[k] = dict(max=float(d.max()), mean=float(d.mean()), frac_above_tenth_step=float((d > 0.1 * lr).float().mean()))
    return out


# Parameter deviations MEASURED on the MI355X (max / mean over all tensors after the free-running steps of each test;
# profiles/r05_parity_full_size.jsonl, re-measured in round 6: profiles/r06_parity_full_size.jsonl).  Round 6 (VERDICT r5
# item 1): every case is asserted at north_star's 1e-5 (or 10x its measured deviation where that is SMALLER)
# over every parameter element -- except the elements the ORACLE'S OWN RUN marks as ill-conditioned (class _Evidence):
#   (k) the weight row / bias element / next-layer column of a hidden unit whose pre-activation comes within 2e-7 of the
#       relu's kink in some training forward of the oracle: two fp32 summation orders disagree on the sign of a 784-term
#       dot product that small, the unit takes the other branch for one sample, and Adam's normalised step turns that
#       sample's missing contribution into whole steps of the unit's own weights (VAE bs=512: all 96 elements above 1e-5
#       are row 119 of encoder.linear -- |pre| = 7.5e-9 in the first batch -- everything else is within 4.5e-8;
#       f_pearson's critic: all 334 in row 238 of D.linear, the rest within 8.6e-7);
#   (z) elements whose oracle gradient passes within 5 % of the tensor's typical gradient magnitude of ZERO in some step
#       (min_t |g_t| <= 0.05 * median_elements(max_t |g_t|)): Adam's m / (sqrt(v) + 1e-8) turns a last-bit difference
#       of such a gradient into a fraction of a step (profiles/r05_adam_amplification_cpu.json: two CPU evaluations
#       of the same loop differ by the same 1.84e-4 there).  f_hellinger: the 98 elements above 1e-5 all lie in this set.
# Inside the marked set a case may deviate up to 2x what was measured there (INSIDE_MEASURED); a case without an entry
# gets no allowance at all.  A defect of 1e-3 in any element outside the set fails.
PARAM_MEASURED = {
    "dra": (1.0e-7, 8.5e-11), "be": (1.3e-7, 9.1e-11), "info": (1.5e-6, 4.7e-8), "ra": (1.5e-7, 1.8e-10),
    "fisher": (5.2e-8, 8.0e-11), "mm": (1.3e-7, 2.5e-10), "w": (4.1e-6, 5.3e-8), "f_pearson": (8.4e-6, 3.3e-7),
    "wgp_d5": (1.1e-6, 2.4e-10), "dra_d5": (3.0e-6, 1.2e-8), "w_d5": (1.5e-8, 5.0e-11),
    "f_total_variation": (1.9e-8, 1.0e-10), "f_forward_kl": (5.7e-7, 9.6e-11), "f_reverse_kl": (2.8e-8, 7.5e-11),
    "f_hellinger": (8.1e-6, 3.9e-7), "f_jensen_shannon": (2.6e-8, 8.5e-11),
    "ns_b256": (1.2e-7, 3.0e-10), "wgp_b256": (1.4e-6, 1.3e-10), "ls_b1024": (3.8e-7, 7.3e-9), "ns_b1024": (3.5e-6, 2.4e-8),
    "vae_b512_ragged": (4.5e-8, 3.1e-8),
    "ns_b100": (3.2e-7, 3.2e-10), "ns_b64": (4.8e-7, 4.3e-10), "wgp_b100": (1.5e-7, 2.0e-10), "ls_b100": (9.3e-8, 1.7e-10),
}
# largest deviation measured INSIDE the oracle-marked element set (everything else: no allowance)
INSIDE_MEASURED = {"f_hellinger": 1.85e-4, "f_pearson": 4.7e-5, "vae_b512_ragged": 4.4e-4}
# which marks a case is granted: the VAE's deviations are ALL one kink unit's row (everything else within 4.5e-8), so it
# gets (k) only -- its gradients differ by orders of magnitude between the 512- and the 336-row batches and (z) would mark
# three quarters of encoder.linear
MARKS = {"f_hellinger": "kz", "f_pearson": "kz", "vae_b512_ragged": "k"}
KINK = 2e-7            # |pre-activation| below which two fp32 summation orders may disagree on its sign
NEAR_ZERO = 0.05       # of the tensor's median gradient magnitude
GAN_LAYERS = {"D.linear": (("D.linear.weight", "D.linear.bias"), ("D.discriminate.weight",)),
              "G.linear": (("G.linear.weight", "G.linear.bias"), ("G.generate.weight",))}
VAE_LAYERS = {"encoder.linear": (("encoder.linear.weight", "encoder.linear.bias"), ("encoder.mu.weight", "encoder.log_var.weight")),
              "decoder.linear": (("decoder.linear.weight", "decoder.linear.bias"), ("decoder.recon.weight",))}


class _Evidence:
    """What the oracle's own free-running run says about WHICH parameter elements are ill-conditioned: forward hooks on
    the hidden layers (training forwards only) record every unit's smallest |pre-activation|; tap() keeps every
    parameter element's smallest and largest |gradient| over the optimizer steps."""

    def __init__(self, o_model, layers):
        self.layers, self.pre, self.gmin, self.gmax = layers, {}, {}, {}
        for name in layers:
            mod = o_model
            for part in name.split("."):
                mod = getattr(mod, part)
            mod.register_forward_hook(lambda m_, i_, out, name=name: self._pre(o_model, name, out))

    def _pre(self, o_model, name, out):
        if o_model.training:
            v = out.detach().abs().min(dim=0).values
            self.pre[name] = v if name not in self.pre else torch.minimum(self.pre[name], v)

    def tap(self, kind, tr_, info):
        if kind in ("D", "G"):
            named = [("%s.%s" % (kind, n), p_) for n, p_ in getattr(tr_.model, kind).named_parameters()]
        elif kind == "VAE":
            named = list(tr_.model.named_parameters())
        else:
            return
        for n, p_ in named:
            g = p_.grad.detach().abs()
            self.gmin[n] = g.clone() if n not in self.gmin else torch.minimum(self.gmin[n], g)
            self.gmax[n] = g.clone() if n not in self.gmax else torch.maximum(self.gmax[n], g)

    def kink_units(self):
        return {name: torch.nonzero(v <= KINK).flatten().tolist() for name, v in self.pre.items()}

    def marked(self, key, shape, marks="kz"):
        m = torch.zeros(shape, dtype=torch.bool)
        for name, (rows, cols) in (self.layers.items() if "k" in marks else ()):
            units = torch.nonzero(self.pre[name] <= KINK).flatten()
            if key in rows:
                m[units] = True
            if key in cols:
                m[:, units] = True
        if "z" in marks and key in self.gmin and self.gmin[key].numel() >= 64:           # (a median over a handful of elements says nothing)
            m |= self.gmin[key] <= NEAR_ZERO * self.gmax[key].median()
        return m


def _param_assert(key, dev, model=None, o_model=None, ev=None):
    """dev: _param_dev(...).  Every element within min(1e-5, 10x this case's measured deviation) (floor 1e-6) and every
    tensor's mean within 10x measured (floor 1e-9).  Cases with an INSIDE_MEASURED entry: the elements the oracle's own
    run marks (ev.marked) may deviate up to 2x what was measured inside the set; the bound on all others stays."""
    mx, mean = PARAM_MEASURED[key]
    bmax, bmean = min(max(10 * mx, 1e-6), 1e-5), max(10 * mean, 1e-9)
    if key not in INSIDE_MEASURED:
        for k, v in dev.items():
            assert v["max"] <= bmax and v["mean"] <= bmean, (key, k, v, bmax, bmean)
        return None
    inside_bound, rec = 2 * INSIDE_MEASURED[key], {}
    for (k, a), (_, b) in zip(model.state_dict().items(), o_model.state_dict().items()):
        d = (a.cpu() - b).abs()
        m = ev.marked(k, d.shape, MARKS[key])
        out_max = float(d[~m].max()) if bool((~m).any()) else 0.0
        in_max = float(d[m].max()) if bool(m.any()) else 0.0
        rec[k] = dict(outside_max=out_max, inside_max=in_max, marked_frac=float(m.float().mean()),
                      above_1e5=int((d > 1e-5).sum()), above_1e5_outside=int((d[~m] > 1e-5).sum()))
        assert out_max <= bmax, (key, k, "an element the oracle does not mark as ill-conditioned deviates", rec[k])
        assert in_max <= inside_bound, (key, k, rec[k], inside_bound)
        assert float(m.float().mean()) <= 0.5, (key, k, "the marked set must stay a minority", rec[k])
        assert dev[k]["mean"] <= bmean, (key, k, dev[k], bmean)
    return dict(kink_units=ev.kink_units(), tensors=rec)


@pytest.mark.parametrize("variant,kw", FULL_CASES, ids=[_case_id(v, kw) for v, kw in FULL_CASES])
def test_full_size_engine_vs_oracle(variant, kw):
    steps = 6

    class Capped(torch.utils.data.DataLoader):
        def __len__(self):
            return steps * kw.get("D_steps", 1)

    def loaders():
        ld = port.synthetic_loaders(256, n_train=FULLCFG["n_train"], n_val=256, n_test=256,
                                    image_shape=FULLCFG["image_shape"])
        return (Capped(ld[0].dataset, batch_size=256, shuffle=True),) + ld[1:]

    ld = loaders()
    o_model = port.build(variant, 784, 400, 20)
    okw = dict(kw)
    method = okw.pop("method", "jensen_shannon")
    ev = _Evidence(o_model, GAN_LAYERS) if _case_id(variant, kw) in INSIDE_MEASURED else None
    o = port.GANPort(variant, o_model, ld[0], method=method, tap=ev.tap if ev else None)
    o.train(**okw)
    o_rng = torch.get_rng_state()

    tr, model = build_product(variant, FULLCFG, 256, loaders=loaders())
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(**kw)
    torch.cuda.synchronize()
    assert tr._engine is not None
    lclose(tr.Dlosses, o.Dlosses, "%s full-size Dlosses" % variant)
    lclose(tr.Glosses, o.Glosses, "%s full-size Glosses" % variant)
    if variant == "info":
        lclose(tr.MIlosses, o.MIlosses, "info full-size MIlosses")
    assert torch.equal(o_rng, torch.get_rng_state())
    dev = _param_dev(model, o_model, 2e-4)
    attributed = None
    try:
        attributed = _param_assert(_case_id(variant, kw), dev, model, o_model, ev)
    finally:
        _record("full_size_engine_vs_oracle[%s]" % _case_id(variant, kw), steps=steps, attributed=attributed,
                Dloss_err=_loss_err(tr.Dlosses, o.Dlosses), Gloss_err=_loss_err(tr.Glosses, o.Glosses), params=dev)


# ---------------------------------------------------------------------------------------------
# The four BASELINE.json GPU configurations, TENSOR BY TENSOR against the CPU oracle (oracle/port.py,
# pinned bit for bit to the unmodified reference) after >= 12 free-running steps: NSGAN bs=256,
# WGAN-GP bs=256, NSGAN / LSGAN bs=1024 (ns_gan.py:94-170, w_gp_gan.py:96-175, ls_gan.py:95-171) and
# the VAE at bs=512 over epochs that end with the ragged 336 batch (vae.py:127-191).  The golden
# fixtures of these configurations only carry per-tensor digests.
# ---------------------------------------------------------------------------------------------
BASELINE_GAN_CASES = [("ns", 256, dict(num_epochs=1)), ("wgp", 256, dict(num_epochs=1, D_steps=1)),
                      ("ls", 1024, dict(num_epochs=1)), ("ns", 1024, dict(num_epochs=1))]


@pytest.mark.parametrize("variant,batch,kw", BASELINE_GAN_CASES,
                         ids=["%s_b%d" % (v, b) for v, b, _ in BASELINE_GAN_CASES])
def test_baseline_configs_parameters_tensor_by_tensor(variant, batch, kw):
    steps = 12

    class Capped(torch.utils.data.DataLoader):
        def __len__(self):
            return steps * kw.get("D_steps", 1)

    def loaders():
        ld = port.synthetic_loaders(batch, n_train=FULLCFG["n_train"], n_val=256, n_test=256,
                                    image_shape=FULLCFG["image_shape"])
        return (Capped(ld[0].dataset, batch_size=batch, shuffle=True),) + ld[1:]

    ld = loaders()                                  # (seeds the dataset generator: BEFORE the model's seed)
    o_model = port.build(variant, 784, 400, 20)
    o = port.GANPort(variant, o_model, ld[0])
    o.train(**kw)
    o_rng = torch.get_rng_state()
    tr, model = build_product(variant, FULLCFG, batch, loaders=loaders())
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(**kw)
    torch.cuda.synchronize()
    assert tr._engine is not None and len(tr.Glosses) == steps
    lclose(tr.Dlosses, o.Dlosses, "%s bs=%d Dlosses" % (variant, batch))
    lclose(tr.Glosses, o.Glosses, "%s bs=%d Glosses" % (variant, batch))
    assert torch.equal(o_rng, torch.get_rng_state())
    dev = _param_dev(model, o_model, 2e-4)
    _record("baseline_configs_tensor_by_tensor[%s_b%d]" % (variant, batch), steps=steps,
            Dloss_err=_loss_err(tr.Dlosses, o.Dlosses), Gloss_err=_loss_err(tr.Glosses, o.Glosses), params=dev)
    _param_assert("%s_b%d" % (variant, batch), dev)


def test_vae_b512_ragged_parameters_tensor_by_tensor():
    """VAE 784-400-20, B = 512, n = 3 * 512 + 336: three epochs = 12 training batches, every epoch
    ending with the ragged 336 batch of the real 50 000-image run, + the per-epoch validation pass."""
    import vae
    n_train = 512 * 3 + 336
    mk = lambda: port.synthetic_loaders(512, n_train=n_train, n_val=512, n_test=512, image_shape=(1, 28, 28))
    ld0 = mk()                                      # (seeds the dataset generator: BEFORE the model's seed)
    o_model = port.build("vae", 784, 400, 20)
    ev = _Evidence(o_model, VAE_LAYERS)
    o = port.VAEPort(o_model, *ld0, tap=ev.tap)
    o.train(3)
    o_rng = torch.get_rng_state()
    ld = mk()
    torch.manual_seed(1234)
    model = vae.VAE(image_size=784, hidden_dim=400, z_dim=20)
    tr = vae.VAETrainer(model, *ld, viz=False)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(num_epochs=3)
    torch.cuda.synchronize()
    assert len(tr.recon_loss) == 12
    dev = _param_dev(model, o_model, 1e-3)
    attributed = None
    try:
        attributed = _param_assert("vae_b512_ragged", dev, model, o_model, ev)
    finally:
        _record("vae_b512_ragged_tensor_by_tensor", recon_err=_loss_err(tr.recon_loss, o.recon_loss),
                kl_err=_loss_err(tr.kl_loss, o.kl_loss), attributed=attributed,
                best_val_rel_err=abs(tr.best_val_loss - o.best_val_loss) / abs(o.best_val_loss), params=dev)
    # measured (profiles/r04_parity_full_size.jsonl): recon 1.0e-7, kl 1.4e-7, best_val 7.6e-8 -- asserted with a
    # 15-fold margin, an order below north_star's 1e-5
    lclose(tr.recon_loss, o.recon_loss, "VAE recon", tol=2e-6)
    lclose(tr.kl_loss, o.kl_loss, "VAE kl", tol=2e-6)
    assert abs(tr.best_val_loss - o.best_val_loss) <= 2e-6 * abs(o.best_val_loss)
    assert torch.equal(o_rng, torch.get_rng_state())


# ---------------------------------------------------------------------------------------------
# Round 5 (VERDICT r4 "what's weak" 1-3).  (a) The reference's own default batch, get_data(BATCH_SIZE=100)
# (/root/reference/src/utils.py:16), and BASELINE.json configs[0]'s B = 64 at the real widths: neither is a
# multiple of the 16 / 32-row tiles, so they reach the row-tail paths of every kernel.  (b) LOCKSTEP
# teacher-forced gradients at full size: before every step the product's parameters are overwritten with the
# oracle's and both start from the same generator state, so every gradient tensor can be compared without an
# optimizer in between -- this separates "a kernel computes a wrong gradient" from "Adam's 1 / (sqrt(v) + 1e-8)
# amplifies a last-bit difference of an almost-zero gradient" (tools/adam_amplification_cpu.py shows two CPU
# evaluations of the same loop differ by what the GPU path differs from the CPU by).
# ---------------------------------------------------------------------------------------------
def _capped_loaders(batch, n_batches, n_train=None):
    class Capped(torch.utils.data.DataLoader):
        def __len__(self):
            return n_batches
    ld = port.synthetic_loaders(batch, n_train=n_train or FULLCFG["n_train"], n_val=256, n_test=256,
                                image_shape=FULLCFG["image_shape"])
    return (Capped(ld[0].dataset, batch_size=batch, shuffle=True),) + ld[1:]


DEFAULT_BATCH_CASES = [("ns", 100, dict(num_epochs=1), 12), ("ns", 64, dict(num_epochs=1), 12),
                       ("wgp", 100, dict(num_epochs=1, D_steps=5), 6), ("ls", 100, dict(num_epochs=1), 12)]


@pytest.mark.parametrize("variant,batch,kw,steps", DEFAULT_BATCH_CASES,
                         ids=["%s_b%d" % (v, b) for v, b, _, _ in DEFAULT_BATCH_CASES])
def test_reference_default_batch_full_size(variant, batch, kw, steps):
    """784-400-20 at B = 100 (utils.py:16: every number in BASELINE.md was taken at it) and B = 64: free-running
    iterations against the CPU oracle; losses 1e-5, sampling protocol bit-exact, parameters tensor by tensor."""
    nb = steps * kw.get("D_steps", 1)
    ld = _capped_loaders(batch, nb)
    o_model = port.build(variant, 784, 400, 20)
    o = port.GANPort(variant, o_model, ld[0])
    o.train(**kw)
    o_rng = torch.get_rng_state()
    tr, model = build_product(variant, FULLCFG, batch, loaders=_capped_loaders(batch, nb))
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(**kw)
    torch.cuda.synchronize()
    assert tr._engine is not None and tr._stock() and len(tr.Glosses) == steps
    lclose(tr.Dlosses, o.Dlosses, "%s bs=%d Dlosses" % (variant, batch))
    lclose(tr.Glosses, o.Glosses, "%s bs=%d Glosses" % (variant, batch))
    assert torch.equal(o_rng, torch.get_rng_state())
    dev = _param_dev(model, o_model, 2e-4)
    _record("reference_default_batch_full_size[%s_b%d]" % (variant, batch), steps=steps,
            Dloss_err=_loss_err(tr.Dlosses, o.Dlosses), Gloss_err=_loss_err(tr.Glosses, o.Glosses), params=dev)
    _param_assert("%s_b%d" % (variant, batch), dev)


def test_vae_reference_default_batch_full_epoch():
    """vae.py with the reference's default loaders: B = 100, 50 000 images = 500 batches per epoch, no ragged tail,
    + the validation pass; one whole epoch free-running against the CPU oracle -- and against the same oracle with exp()
    rounded correctly (port.rounded_exp): by batch ~40 the posterior has collapsed (KL ~ 1) and exp(lv) - lv - 1 cancels
    five digits, so the last bit of exp() steers the run; torch's CPU exp and the device's expf differ in that bit now
    and then.  Two CPU runs that differ ONLY in it end 6e-2 apart in the parameters (profiles/r05_vae_exp_rounding.json),
    as far as the HIP path ends from either of them."""
    import vae
    mk = lambda: port.synthetic_loaders(100, n_train=50000, n_val=1000, n_test=200, image_shape=(1, 28, 28))
    ld0 = mk()
    o_model = port.build("vae", 784, 400, 20)
    o = port.VAEPort(o_model, *ld0)
    o.train(1)
    o_rng = torch.get_rng_state()
    with port.rounded_exp():
        ld1 = mk()
        x_model = port.build("vae", 784, 400, 20)
        x = port.VAEPort(x_model, *ld1)
        x.train(1)
    ld = mk()
    torch.manual_seed(1234)
    model = vae.VAE(image_size=784, hidden_dim=400, z_dim=20)
    tr = vae.VAETrainer(model, *ld, viz=False)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(num_epochs=1)
    torch.cuda.synchronize()
    assert len(tr.recon_loss) == 500 and tr._engine is not None
    assert torch.equal(o_rng, torch.get_rng_state())
    rel = lambda a, b: np.abs(np.asarray(a) - np.asarray(b)) / np.maximum(1, np.abs(np.asarray(b)))
    er, ek = rel(tr.recon_loss, o.recon_loss), rel(tr.kl_loss, o.kl_loss)
    xr, xk = rel(tr.recon_loss, x.recon_loss), rel(tr.kl_loss, x.kl_loss)
    dev_o, dev_x = _param_dev(model, o_model, 1e-3), _param_dev(model, x_model, 1e-3)
    cpu_pair = _param_dev(o_model, x_model, 1e-3)
    pmax = lambda d: max(v["max"] for v in d.values())
    _record("vae_b100_full_epoch", recon_err_first30=float(er[:30].max()), kl_err_first30=float(ek[:30].max()),
            first_batch_the_oracle_moves_1e6_under_exp_rounding=int(np.argmax(np.maximum(rel(o.recon_loss, x.recon_loss), rel(o.kl_loss, x.kl_loss)) > 1e-6)),
            recon_err_500=float(er.max()), kl_err_500=float(ek.max()),
            vs_rounded_exp=dict(recon_err_500=float(xr.max()), kl_err_500=float(xk.max()), param_max=pmax(dev_x)),
            cpu_stock_vs_cpu_rounded_exp_param_max=pmax(cpu_pair),
            best_val_rel_err=abs(tr.best_val_loss - o.best_val_loss) / abs(o.best_val_loss), params=dev_o)
    # against the stock oracle: north_star's bound while the problem is well conditioned (measured 4e-7 up to batch 33;
    # the KL term falls from 1 300 to ~20 over these batches and the posterior collapses right behind them).  ADVICE r5:
    # the strict check over this horizon extends to every parameter's gradient in
    # test_vae_full_size_teacher_forced_gradients_lockstep[100] (30 batches, measured <= 7.6e-7 of each tensor's scale);
    # the drift-relative criterion below is what remains once exp(lv) - lv - 1 cancels five digits.  (Where the ORACLE
    # ITSELF first moves by 1e-6 under a different rounding of exp() depends on the host's libm / vector width: batch
    # 31 - 35 in the build container, 80 on the GPU box -- recorded, not asserted.)
    assert max(er[:30].max(), ek[:30].max()) <= TOL, (er[:30].max(), ek[:30].max())
    # over the whole epoch the HIP path must not be further from the stock oracle than the oracle's own exp() rounding
    # moves it (factor 3 of slack).  (Measured: the three evaluations -- torch's exp, the rounded exp, the device's expf
    # -- end pairwise 6.0e-2 / 6.0e-2 / 6.1e-2 apart in the parameters and 5e-2 / 5e-2 / 6e-2 in the KL term: once the
    # last bit of exp() matters, every choice of it is its own trajectory.)
    assert max(er.max(), ek.max()) <= 3 * max(rel(o.recon_loss, x.recon_loss).max(), rel(o.kl_loss, x.kl_loss).max()) + TOL
    assert pmax(dev_o) <= 3 * pmax(cpu_pair) + 1e-5
    assert abs(tr.best_val_loss - o.best_val_loss) <= 1e-4 * abs(o.best_val_loss)


def _grad_errs(pairs):
    """[(name, got, ref)] -> {name: (abs err, scale)}"""
    return {n: (float((g.cpu() - r).abs().max()), float(r.abs().max())) for n, g, r in pairs}


LOCKSTEP_CASES = [("ns", 256, {}), ("wgp", 256, {}), ("ls", 1024, {}), ("ns", 1024, {}), ("ns", 100, {}),
                  ("f", 256, dict(method="hellinger")), ("f", 256, dict(method="pearson")),
                  # the other loss families at the real widths (mm_gan.py:138-170, w_gan.py:140-175, ra_gan.py:197-214,
                  # fisher_gan.py:199-229, dra_gan.py:192-231)
                  ("mm", 256, dict(G_init=0)), ("w", 256, {}), ("ra", 256, {}), ("fisher", 256, {}), ("dra", 256, {})]
GRAD_TOL = 1e-5            # relative to the tensor's largest gradient element
GRAD_ABS = 1e-7            # + an absolute floor: the critic's output-bias gradient is two nearly cancelling half-sums


@pytest.mark.parametrize("variant,batch,kw", LOCKSTEP_CASES,
                         ids=["%s_b%d%s" % (v, b, "_" + k["method"] if "method" in k else "") for v, b, k in LOCKSTEP_CASES])
def test_full_size_teacher_forced_gradients_lockstep(variant, batch, kw):
    """784-400-20, three D+G iterations in lockstep: same parameters, same generator state, same batch and noise on
    both sides before every iteration; dL_D / dD-parameters of the critic step and dL_G / dG-parameters of the
    generator step (on the critic as updated by the step before it) against the oracle's autograd, tensor by
    tensor (ns_gan.py:122-156, w_gp_gan.py:177-220, ls_gan.py:95-171, f_gan.py:99-142)."""
    steps = 3
    grads = {}

    def tap(kind, tr_, info):
        if kind in ("D", "G"):
            grads[kind] = [p.grad.detach().clone() for p in getattr(tr_.model, kind).parameters()]
    ld_o = _capped_loaders(batch, 1)
    o_model = port.build(variant, 784, 400, 20)
    okw = dict(kw)
    o = port.GANPort(variant, o_model, ld_o[0], method=okw.pop("method", "jensen_shannon"), tap=tap)
    # every hidden pre-activation the oracle forms in a step (D on x and G(z) in the critic step, D on G(z) in the
    # generator step; G's hidden layer): how close does the REFERENCE come to the kink of a relu?
    kink = {"D": [], "G": []}
    o_model.D.linear.register_forward_hook(lambda m_, i_, out: kink["D"].append(float(out.detach().abs().min())))
    o_model.G.linear.register_forward_hook(lambda m_, i_, out: kink["G"].append(float(out.detach().abs().min())))
    tr, model = build_product(variant, FULLCFG, batch, loaders=_capped_loaders(batch, 1))
    import contextlib, io
    worst, flips = {}, []
    for step in range(steps):
        model.load_state_dict(o_model.state_dict())               # teacher forcing: the oracle's parameters
        kink["D"].clear(); kink["G"].clear()
        s0 = torch.get_rng_state()
        o.train(num_epochs=1, D_steps=1, max_steps=1, **okw)
        s1 = torch.get_rng_state()
        torch.set_rng_state(s0)
        with contextlib.redirect_stdout(io.StringIO()):
            tr.train(num_epochs=1, D_steps=1, **kw)
        torch.cuda.synchronize()
        assert torch.equal(s1, torch.get_rng_state()), "same draws on both sides (step %d)" % step
        eng = tr._engine
        assert eng is not None and tr._stock()
        lclose(tr.Dlosses[-1:], o.Dlosses[-1:], "lockstep D loss")
        lclose(tr.Glosses[-1:], o.Glosses[-1:], "lockstep G loss")
        step_bad = []
        for net, fp in (("D", eng.fD), ("G", eng.fG)):
            names = [n for n, _ in getattr(model, net).named_parameters()]
            pairs = [("%s.%s" % (net, n), fp.grad[off:off + p_.numel()].view(p_.shape), ref)
                     for n, p_, off, ref in zip(names, fp.params, fp.offsets, grads[net])]
            for n, (aerr, scale) in _grad_errs(pairs).items():
                rel = aerr / max(scale, 1e-30)
                if aerr > GRAD_TOL * scale + GRAD_ABS:
                    step_bad.append((n, rel, aerr, scale))
                elif rel > worst.get(n, (0, 0, 0))[0]:
                    worst[n] = (rel, aerr, scale)
        if step_bad:
            # The only deviation a correct fp32 evaluation may show beyond rounding: a hidden unit whose pre-activation
            # is within rounding of 0 in the REFERENCE takes the other branch of the relu (its row's whole contribution
            # appears / disappears in the gradients behind it).  Accepted only with that evidence and only at the size
            # of one row's contribution; the strict bound stays on every other tensor and step.
            k_d, k_g = min(kink["D"]), min(kink["G"])
            flips.append(dict(step=step, tensors={n: rel for n, rel, _, _ in step_bad}, min_abs_preact_D=k_d, min_abs_preact_G=k_g))
            # (fp32 rounding of a 784-term dot product of O(1) size is ~1e-7: two summation orders can only disagree on
            # the sign of a pre-activation that small)
            assert min(k_d, k_g) <= 2e-7, ("gradient off without a relu at its kink", step_bad, k_d, k_g)
            for n, rel, aerr, scale in step_bad:
                assert rel <= 1e-2 and aerr <= 1e-4, (variant, batch, n, rel, aerr, scale)
    tag = "%s_b%d%s" % (variant, batch, "_" + kw["method"] if "method" in kw else "")
    _record("full_size_teacher_forced_lockstep[%s]" % tag, steps=steps, relu_kink_steps=flips,
            grad_rel_err={n: v[0] for n, v in worst.items()}, grad_abs_err={n: v[1] for n, v in worst.items()},
            grad_scale={n: v[2] for n, v in worst.items()})
    assert len(flips) <= 1, flips                                 # (measured: 0 or 1 per case)


@pytest.mark.parametrize("batch", [512, 100])
def test_vae_full_size_teacher_forced_gradients_lockstep(batch):
    """vae.py:193-212 at 784-400-20: batches in lockstep (one training batch + the validation pass per call), every
    parameter's gradient of recon + kl against the oracle's autograd.  B = 512: four batches; B = 100 (the reference's
    default loaders): thirty -- the horizon over which test_vae_reference_default_batch_full_epoch holds the losses to
    1e-5 (the strict check there extended to the parameters' gradients)."""
    import vae
    steps = 30 if batch == 100 else 4
    mk = lambda: port.synthetic_loaders(batch, n_train=batch, n_val=batch, n_test=batch, image_shape=(1, 28, 28))
    ld0 = mk()
    o_model = port.build("vae", 784, 400, 20)
    o = port.VAEPort(o_model, *ld0)
    ld = mk()
    torch.manual_seed(1234)
    model = vae.VAE(image_size=784, hidden_dim=400, z_dim=20)
    tr = vae.VAETrainer(model, *ld, viz=False)
    import contextlib, io
    worst = {}
    for step in range(steps):
        model.load_state_dict(o_model.state_dict())
        s0 = torch.get_rng_state()
        o.train(1)
        s1 = torch.get_rng_state()
        torch.set_rng_state(s0)
        with contextlib.redirect_stdout(io.StringIO()):
            tr.train(num_epochs=1)
        torch.cuda.synchronize()
        assert torch.equal(s1, torch.get_rng_state())
        lclose(tr.recon_loss[-1:], o.recon_loss[-1:], "lockstep recon", tol=2e-6)
        lclose(tr.kl_loss[-1:], o.kl_loss[-1:], "lockstep kl", tol=2e-6)
        fp = tr._engine.fp
        ref = dict(o_model.named_parameters())
        name_of = {id(p_): n for n, p_ in model.named_parameters()}
        pairs = [(name_of[id(p_)], fp.grad[off:off + p_.numel()].view(p_.shape), ref[name_of[id(p_)]].grad)
                 for p_, off in zip(fp.params, fp.offsets)]
        assert len(pairs) == len(ref)
        for n, (aerr, scale) in _grad_errs(pairs).items():
            rel = aerr / max(scale, 1e-30)
            if rel > worst.get(n, (0, 0, 0))[0]:
                worst[n] = (rel, aerr, scale)
    _record("vae_full_size_teacher_forced_lockstep[b%d]" % batch, steps=steps,
            grad_rel_err={n: v[0] for n, v in worst.items()}, grad_abs_err={n: v[1] for n, v in worst.items()},
            grad_scale={n: v[2] for n, v in worst.items()})
    for n, (rel, aerr, scale) in worst.items():
        assert aerr <= GRAD_TOL * scale + GRAD_ABS, (batch, n, rel, aerr, scale)


@pytest.mark.parametrize("hidden", [528, 800])
def test_vae_hidden_wider_than_the_fused_backward(hidden):
    """gm_vae_bwd_mid keeps one hidden width <= 512 per workgroup (gm_fused.hip); VAE(hidden_dim=800) must take the
    two generic dX launches instead of raising."""
    cfg = dict(SMALL, hidden_dim=hidden)
    loaders = port.synthetic_loaders(cfg["batch"], n_train=160, n_val=cfg["n_val"], n_test=cfg["n_test"],
                                     image_shape=tuple(cfg["image_shape"]))
    o_model = port.build("vae", cfg["image_size"], hidden, cfg["z_dim"])
    o = port.VAEPort(o_model, *loaders)
    o.train(1)
    o_rng = torch.get_rng_state()
    p, p_model, p_rng = run_vae_product(cfg, cfg["batch"], 160, 1)
    lclose(np.array(p.recon_loss) / 100, np.array(o.recon_loss) / 100, "vae recon")
    lclose(p.kl_loss, o.kl_loss, "vae kl", tol=1e-5)
    assert torch.equal(o_rng, p_rng)
    for (k, a), (_, b) in zip(p_model.state_dict().items(), o_model.state_dict().items()):
        assert (a.cpu() - b).abs().max().item() <= 5e-5, k


def test_bench_contract_line_end_to_end():
    """`python bench.py` with the driver's flags (short: no CPU legs, no configs section) runs through and prints the
    contract's JSON line last on stdout, with the roofline object -- the one command of the repo that no other test
    executes, and the one the round is judged by (round 4: an engine clean-up removed an attribute only bench.py read)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                        "--reps", "2", "--no-cpu-baseline", "--no-configs", "--sustained", "0"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().split("\n")[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in line, k
    assert line["steps"] == 20 and line["warmup"] == 5 and line["n_gpus"] == 1 and line["value"] > 1e6
    rf = line["roofline"]
    assert rf["bound"] == "mfma" and 0.05 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert "workload" in line["config"]
    # the evidence lives inside the objects the driver's record keeps (config / roofline), not in extra top-level keys
    assert 0.02 < rf["step_frac"] < rf["frac"] and "profile_kernel_us" in rf and isinstance(rf["stale_profile"], bool)
    assert line["config"]["steady_us_per_step"] > 10 and line["config"]["run_fixed_cost_us"] is not None
