"""Pins oracle/port.py bit-for-bit against the UNMODIFIED reference executed on CPU.

Runs only where /root/reference is mounted (the build container); on the GPU box these tests
skip and the committed fixtures (tests/test_golden.py) carry the pin instead."""
import numpy as np
import pytest
import torch

from oracle import port, ref_harness

needs_ref = pytest.mark.skipif(not ref_harness.available(), reason="reference not mounted")

IMG, HID, Z, B, N = 64, 48, 8, 16, 160       # small dims: same code path, seconds to run


SMALL_DIMS = dict(img=IMG, hid=HID, z=Z, batch=B, n_train=N, n_val=48, shape=(1, 8, 8))


def _loaders(batch, dims=SMALL_DIMS):
    return ref_harness.synthetic_loaders(batch, n_train=dims["n_train"], n_val=dims["n_val"], n_test=dims["n_val"],
                                         image_shape=dims["shape"])


def _run_reference(variant, steps_kw, dims=SMALL_DIMS):
    mod_name, model_name, trainer_name = port.REFERENCE_NAMES[variant]
    mod = ref_harness.load(mod_name)
    train_iter, val_iter, test_iter = _loaders(dims["batch"], dims)
    torch.manual_seed(1234)
    IMG, HID, Z = dims["img"], dims["hid"], dims["z"]
    if variant == "info":
        model = getattr(mod, model_name)(image_size=IMG, hidden_dim=HID, z_dim=Z, disc_dim=10,
                                         cont_dim=10)
    else:
        model = getattr(mod, model_name)(image_size=IMG, hidden_dim=HID, z_dim=Z)
    trainer = getattr(mod, trainer_name)(model, train_iter, val_iter, test_iter, viz=False)
    with ref_harness.quiet():
        trainer.train(**steps_kw)
    return trainer, model


def _run_port(variant, steps_kw, dims=SMALL_DIMS):
    train_iter, val_iter, test_iter = _loaders(dims["batch"], dims)
    model = port.build(variant, dims["img"], dims["hid"], dims["z"])
    kw = dict(steps_kw)
    method = kw.pop("method", "jensen_shannon")
    trainer = port.GANPort(variant, model, train_iter, method=method)
    trainer.train(**kw)
    return trainer, model


GAN_CASES = [
    ("ns", dict(num_epochs=2)),
    ("mm", dict(num_epochs=1, G_init=3)),
    ("w", dict(num_epochs=1, D_steps=2)),
    ("wgp", dict(num_epochs=1, D_steps=1)),
    ("wgp", dict(num_epochs=1, D_steps=3)),
    ("ls", dict(num_epochs=2)),
    ("dra", dict(num_epochs=1, D_steps=1)),
    ("be", dict(num_epochs=2)),
    ("ra", dict(num_epochs=1)),
    ("fisher", dict(num_epochs=1)),
    ("info", dict(num_epochs=1)),
] + [("f", dict(num_epochs=1, method=m)) for m in port.F_METHODS]


@needs_ref
@pytest.mark.parametrize("variant,kw", GAN_CASES, ids=[f"{v}-{i}" for i, (v, _) in enumerate(GAN_CASES)])
def test_gan_port_bit_exact(variant, kw):
    ref_tr, ref_model = _run_reference(variant, kw)
    ref_rng = torch.get_rng_state()
    my_tr, my_model = _run_port(variant, kw)
    assert len(ref_tr.Glosses) == len(my_tr.Glosses) > 0
    np.testing.assert_array_equal(np.array(ref_tr.Glosses), np.array(my_tr.Glosses))
    np.testing.assert_array_equal(np.array(ref_tr.Dlosses), np.array(my_tr.Dlosses))
    if variant == "info":
        np.testing.assert_array_equal(np.array(ref_tr.MIlosses), np.array(my_tr.MIlosses))
    ref_sd, my_sd = ref_model.state_dict(), my_model.state_dict()
    assert list(ref_sd.keys()) == list(my_sd.keys())          # drop-in key names
    for k in ref_sd:
        assert torch.equal(ref_sd[k], my_sd[k]), k
    # both consumed the global generator identically
    assert torch.equal(ref_rng, torch.get_rng_state())


def _assert_gan_bit_exact(variant, kw, dims):
    ref_tr, ref_model = _run_reference(variant, kw, dims)
    ref_rng = torch.get_rng_state()
    my_tr, my_model = _run_port(variant, kw, dims)
    assert len(ref_tr.Glosses) == len(my_tr.Glosses) > 0
    np.testing.assert_array_equal(np.array(ref_tr.Glosses), np.array(my_tr.Glosses))
    np.testing.assert_array_equal(np.array(ref_tr.Dlosses), np.array(my_tr.Dlosses))
    ref_sd, my_sd = ref_model.state_dict(), my_model.state_dict()
    assert list(ref_sd.keys()) == list(my_sd.keys())
    for k in ref_sd:
        assert torch.equal(ref_sd[k], my_sd[k]), k
    assert torch.equal(ref_rng, torch.get_rng_state())
    return len(my_tr.Glosses)


# The same pin at the REAL layer widths (784-400-20: other GEMM blockings and thread splits inside torch's CPU kernels
# than 64-48-8) and the batch sizes that matter: the BASELINE.json configurations (256 / 512 / 1024) and the
# reference's own default, get_data(BATCH_SIZE=100) (/root/reference/src/utils.py:16).  n_train = 4 batches per
# epoch: >= 3 free-running optimizer steps per network, parameters and generator state compared bit for bit.
FULL_PIN_CASES = [("ns", 256, dict(num_epochs=1)), ("ns", 100, dict(num_epochs=1)), ("ns", 64, dict(num_epochs=1)),
                  ("wgp", 256, dict(num_epochs=1, D_steps=1)), ("wgp", 100, dict(num_epochs=3, D_steps=5)),
                  ("ls", 1024, dict(num_epochs=1)), ("f", 256, dict(num_epochs=1, method="hellinger")),
                  ("f", 256, dict(num_epochs=1, method="pearson")),
                  # round 5: every other variant at the real widths too (the GPU parity tests compare against the port at
                  # these shapes: tests/test_gpu_trainers.py FULL_CASES)
                  ("dra", 256, dict(num_epochs=1, D_steps=1)), ("be", 256, dict(num_epochs=1)),
                  ("info", 256, dict(num_epochs=1)), ("ra", 256, dict(num_epochs=1)),
                  ("fisher", 256, dict(num_epochs=1)), ("mm", 256, dict(num_epochs=1, G_init=2)),
                  ("w", 256, dict(num_epochs=2, D_steps=2))]


@needs_ref
@pytest.mark.parametrize("variant,batch,kw", FULL_PIN_CASES,
                         ids=["%s_b%d%s" % (v, b, "_" + k["method"] if "method" in k else "") for v, b, k in FULL_PIN_CASES])
def test_gan_port_bit_exact_full_size(variant, batch, kw):
    dims = dict(img=784, hid=400, z=20, batch=batch, n_train=4 * batch, n_val=batch, shape=(1, 28, 28))
    steps = _assert_gan_bit_exact(variant, kw, dims)
    assert steps >= 3


@needs_ref
@pytest.mark.parametrize("batch,n_train", [(512, 512 * 3 + 336), (100, 400)], ids=["b512_ragged", "b100"])
def test_vae_port_bit_exact_full_size(batch, n_train):
    """vae.py at 784-400-20: B = 512 with the ragged 336 batch of the real 50 000-image epoch, and the reference's
    default B = 100; two epochs incl. the per-epoch validation pass."""
    dims = dict(img=784, hid=400, z=20, batch=batch, n_train=n_train, n_val=batch, shape=(1, 28, 28))
    mod = ref_harness.load("vae")
    tr_i, va_i, te_i = _loaders(batch, dims)
    torch.manual_seed(1234)
    ref_model = mod.VAE(image_size=784, hidden_dim=400, z_dim=20)
    ref_tr = mod.VAETrainer(ref_model, tr_i, va_i, te_i, viz=False)
    with ref_harness.quiet():
        ref_tr.train(num_epochs=2)
    ref_state = torch.get_rng_state()
    tr_i, va_i, te_i = _loaders(batch, dims)
    my_model = port.build("vae", 784, 400, 20)
    my_tr = port.VAEPort(my_model, tr_i, va_i, te_i)
    my_tr.train(num_epochs=2)
    assert len(my_tr.recon_loss) >= 8
    np.testing.assert_array_equal(np.array(ref_tr.recon_loss), np.array(my_tr.recon_loss))
    np.testing.assert_array_equal(np.array(ref_tr.kl_loss), np.array(my_tr.kl_loss))
    assert ref_tr.best_val_loss == my_tr.best_val_loss
    for (k, a), (_, b) in zip(ref_model.state_dict().items(), my_model.state_dict().items()):
        assert torch.equal(a, b), k
    assert torch.equal(ref_state, torch.get_rng_state())


@needs_ref
def test_bir_vae_port_bit_exact():
    """bir_vae.py (second half): the oracle for the next row is pinned ahead of
    the product.  Two generators are in play: torch's (sampling, MMD prior draws) and numpy's
    (reparameterisation noise) -- both must end in the same state."""
    mod = ref_harness.load("bir_vae")
    tr_i, va_i, te_i = _loaders(B)
    torch.manual_seed(1234); np.random.seed(77)
    ref_model = mod.BIRVAE(image_size=IMG, hidden_dim=HID, z_dim=Z)
    ref_tr = mod.BIRVAETrainer(ref_model, tr_i, va_i, te_i, viz=False)
    with ref_harness.quiet():
        ref_tr.train(num_epochs=2)
    ref_state, ref_np = torch.get_rng_state(), np.random.get_state()[1].copy()

    tr_i, va_i, te_i = _loaders(B)
    np.random.seed(77)
    my_model = port.build("bir", IMG, HID, Z)
    my_tr = port.BIRVAEPort(my_model, tr_i, va_i, te_i)
    my_tr.train(num_epochs=2)
    np.testing.assert_array_equal(np.array(ref_tr.recon_loss), np.array(my_tr.recon_loss))
    np.testing.assert_array_equal(np.array(ref_tr.mmd_loss), np.array(my_tr.mmd_loss))
    assert ref_tr.best_val_loss == my_tr.best_val_loss and ref_tr.num_epochs == my_tr.num_epochs == 0
    ref_sd, my_sd = ref_model.state_dict(), my_model.state_dict()
    assert list(ref_sd.keys()) == list(my_sd.keys())
    for k in ref_sd:
        assert torch.equal(ref_sd[k], my_sd[k]), k
    assert torch.equal(ref_state, torch.get_rng_state())
    np.testing.assert_array_equal(ref_np, np.random.get_state()[1])


@needs_ref
def test_ae_port_bit_exact():
    """ae.py: same protocol as the VAE pin, one loss list."""
    mod = ref_harness.load("ae")
    tr_i, va_i, te_i = _loaders(B)
    torch.manual_seed(1234)
    ref_model = mod.Autoencoder(image_size=IMG, hidden_dim=Z)
    ref_tr = mod.AutoencoderTrainer(ref_model, tr_i, va_i, te_i, viz=False)
    with ref_harness.quiet():
        ref_tr.train(num_epochs=2)
    ref_state = torch.get_rng_state()

    tr_i, va_i, te_i = _loaders(B)
    my_model = port.build("ae", IMG, Z)
    my_tr = port.AEPort(my_model, tr_i, va_i, te_i)
    my_tr.train(num_epochs=2)
    np.testing.assert_array_equal(np.array(ref_tr.recon_loss), np.array(my_tr.recon_loss))
    assert ref_tr.best_val_loss == my_tr.best_val_loss
    ref_sd, my_sd = ref_model.state_dict(), my_model.state_dict()
    assert list(ref_sd.keys()) == list(my_sd.keys())
    for k in ref_sd:
        assert torch.equal(ref_sd[k], my_sd[k]), k
    assert torch.equal(ref_state, torch.get_rng_state())


@needs_ref
def test_vae_port_bit_exact():
    mod = ref_harness.load("vae")
    tr_i, va_i, te_i = _loaders(B)
    torch.manual_seed(1234)
    ref_model = mod.VAE(image_size=IMG, hidden_dim=HID, z_dim=Z)
    ref_tr = mod.VAETrainer(ref_model, tr_i, va_i, te_i, viz=False)
    with ref_harness.quiet():
        ref_tr.train(num_epochs=2)
    ref_state = torch.get_rng_state()

    tr_i, va_i, te_i = _loaders(B)
    my_model = port.build("vae", IMG, HID, Z)
    my_tr = port.VAEPort(my_model, tr_i, va_i, te_i)
    my_tr.train(num_epochs=2)
    np.testing.assert_array_equal(np.array(ref_tr.recon_loss), np.array(my_tr.recon_loss))
    np.testing.assert_array_equal(np.array(ref_tr.kl_loss), np.array(my_tr.kl_loss))
    assert ref_tr.best_val_loss == my_tr.best_val_loss
    ref_sd, my_sd = ref_model.state_dict(), my_model.state_dict()
    assert list(ref_sd.keys()) == list(my_sd.keys())
    for k in ref_sd:
        assert torch.equal(ref_sd[k], my_sd[k]), k
    assert torch.equal(ref_state, torch.get_rng_state())


@needs_ref
def test_dropin_modules_expose_the_reference_surface():
    """Every class, function and public method of the reference's src/*.py exists in the drop-in
    module of the same name with the same leading parameters and defaults (extra trailing keyword
    parameters are allowed) -- what a script written against the reference relies on."""
    import importlib
    import inspect
    import os
    import sys
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "generative_models_amd", "src")
    if src not in sys.path:
        sys.path.insert(0, src)

    def sig(f):
        return [(p.name, p.default if p.default is not inspect._empty else "<required>")
                for p in inspect.signature(f).parameters.values()]
    problems = []
    for m in ["ns_gan", "mm_gan", "w_gan", "w_gp_gan", "ls_gan", "dra_gan", "be_gan", "ra_gan", "f_gan",
              "fisher_gan", "info_gan", "vae", "ae", "bir_vae"]:
        ref, mine = ref_harness.load(m), importlib.import_module(m)
        for name, obj in vars(ref).items():
            if getattr(obj, "__module__", None) != ref.__name__:
                continue
            if not hasattr(mine, name):
                problems.append("%s.%s missing" % (m, name))
                continue
            if inspect.isclass(obj):
                for meth, v in vars(obj).items():
                    if callable(v) and (meth in ("__init__", "train") or not meth.startswith("_")):
                        if not hasattr(getattr(mine, name), meth):
                            problems.append("%s.%s.%s missing" % (m, name, meth))
                            continue
                        a, b = sig(v), sig(getattr(getattr(mine, name), meth))
                        if a != b[:len(a)]:
                            problems.append("%s.%s.%s: %s vs %s" % (m, name, meth, a, b))
    assert not problems, problems
