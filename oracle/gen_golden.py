"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz by executing the UNMODIFIED reference
(/root/reference/src/*.py, via oracle/ref_harness.py) on CPU.  Run in the build container:

    python -m oracle.gen_golden

The fixtures travel to the GPU box (where /root/reference does not exist) and pin both
oracle/port.py (tests/test_golden.py, CPU) and the HIP path (tests/test_gpu_trainers.py).

Every fixture is reproducible from seeds: dataset = ref_harness.synthetic_loaders(seed 3435),
model = torch.manual_seed(1234) then the reference constructor, all later draws come from the
global CPU generator in the reference's own order.  Stored: per-step loss lists, final
parameters (small configs) or per-tensor digests (full-size configs), and the final RNG state
digest.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

from oracle import port, ref_harness

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

SMALL = dict(image_size=64, hidden_dim=48, z_dim=8, batch=16, n_train=160, n_val=48, n_test=48,
             image_shape=(1, 8, 8))
FULL = dict(image_size=784, hidden_dim=400, z_dim=20, n_train=50000, n_val=512, n_test=512,
            image_shape=(1, 28, 28))


def digest(t):
    a = t.detach().cpu().numpy().astype(np.float64).ravel()
    return np.array([a.sum(), np.abs(a).sum(), (a * a).sum(), a[0], a[-1], a[a.size // 2]])


def rng_digest():
    return hashlib.sha256(torch.get_rng_state().numpy().tobytes()).hexdigest()


def run_reference(variant, cfg, batch, train_kw, steps_cap=None):
    mod_name, model_name, trainer_name = port.REFERENCE_NAMES[variant]
    mod = ref_harness.load(mod_name)
    loaders = ref_harness.synthetic_loaders(batch, n_train=cfg["n_train"], n_val=cfg["n_val"],
                                            n_test=cfg["n_test"], image_shape=cfg["image_shape"])
    torch.manual_seed(1234)
    kw = dict(image_size=cfg["image_size"], hidden_dim=cfg["hidden_dim"], z_dim=cfg["z_dim"])
    if variant == "info":
        kw.update(disc_dim=10, cont_dim=10)
    model = getattr(mod, model_name)(**kw)
    trainer = getattr(mod, trainer_name)(model, *loaders, viz=False)
    if steps_cap is not None:
        # Cap the number of steps WITHOUT touching the reference: len(train_iter) drives
        # epoch_steps (ns_gan.py:114); a DataLoader subclass reporting a shorter length keeps
        # sampling identical (fresh full permutation per step, first B rows).
        class Capped(torch.utils.data.DataLoader):
            def __len__(self):
                return steps_cap
        ds = loaders[0].dataset
        trainer.train_iter = Capped(ds, batch_size=batch, shuffle=True)
    with ref_harness.quiet():
        trainer.train(**train_kw)
    return trainer, model


def save(name, arrays, meta):
    arrays = dict(arrays)
    arrays["meta"] = np.array(json.dumps(meta))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrays.items() if k != "meta"})


SMALL_CASES = {
    "ns": dict(num_epochs=2), "mm": dict(num_epochs=1, G_init=3), "w": dict(num_epochs=1, D_steps=2),
    "wgp": dict(num_epochs=1, D_steps=1), "ls": dict(num_epochs=2),
    "dra": dict(num_epochs=1, D_steps=1), "be": dict(num_epochs=2), "ra": dict(num_epochs=1),
    "fisher": dict(num_epochs=1), "info": dict(num_epochs=1),
}

FULL_CASES = {      # variant -> (batch, steps, train kwargs); BASELINE.json configs 2,3,5
    "ns": (256, 24, dict(num_epochs=1)),
    "ls": (1024, 12, dict(num_epochs=1)),
    "wgp": (256, 16, dict(num_epochs=1, D_steps=1)),
}


def main():
    if not ref_harness.available():
        sys.exit("reference not mounted; fixtures can only be generated in the build container")
    torch.set_num_threads(1)            # one thread: the most reproducible summation order
    for variant, kw in SMALL_CASES.items():
        tr, model = run_reference(variant, SMALL, SMALL["batch"], kw)
        arrays = {"Glosses": np.array(tr.Glosses), "Dlosses": np.array(tr.Dlosses)}
        if variant == "info":
            arrays["MIlosses"] = np.array(tr.MIlosses)
        for k, v in model.state_dict().items():
            arrays["param:" + k] = v.numpy()
        save(variant + "_small", arrays, dict(variant=variant, cfg=SMALL, train_kw=kw,
                                              rng=rng_digest(), torch=torch.__version__))
    for method in port.F_METHODS:
        kw = dict(num_epochs=1, method=method)
        tr, model = run_reference("f", SMALL, SMALL["batch"], kw)
        arrays = {"Glosses": np.array(tr.Glosses), "Dlosses": np.array(tr.Dlosses)}
        for k, v in model.state_dict().items():
            arrays["param:" + k] = v.numpy()
        save("f_%s_small" % method, arrays, dict(variant="f", cfg=SMALL, train_kw=kw,
                                                 rng=rng_digest(), torch=torch.__version__))
    # VAE small
    mod = ref_harness.load("vae")
    loaders = ref_harness.synthetic_loaders(SMALL["batch"], n_train=SMALL["n_train"],
                                            n_val=SMALL["n_val"], n_test=SMALL["n_test"],
                                            image_shape=SMALL["image_shape"])
    torch.manual_seed(1234)
    model = mod.VAE(image_size=SMALL["image_size"], hidden_dim=SMALL["hidden_dim"],
                    z_dim=SMALL["z_dim"])
    tr = mod.VAETrainer(model, *loaders, viz=False)
    with ref_harness.quiet():
        tr.train(num_epochs=2)
    arrays = {"recon_loss": np.array(tr.recon_loss), "kl_loss": np.array(tr.kl_loss),
              "best_val_loss": np.array(tr.best_val_loss)}
    for k, v in model.state_dict().items():
        arrays["param:" + k] = v.numpy()
    save("vae_small", arrays, dict(variant="vae", cfg=SMALL, train_kw=dict(num_epochs=2),
                                   rng=rng_digest(), torch=torch.__version__))
    # full-size loss curves + digests
    for variant, (batch, steps, kw) in FULL_CASES.items():
        tr, model = run_reference(variant, FULL, batch, kw, steps_cap=steps)
        arrays = {"Glosses": np.array(tr.Glosses), "Dlosses": np.array(tr.Dlosses)}
        for k, v in model.state_dict().items():
            arrays["digest:" + k] = digest(v)
        save("%s_full_b%d" % (variant, batch), arrays,
             dict(variant=variant, cfg=FULL, batch=batch, steps=steps, train_kw=kw,
                  rng=rng_digest(), torch=torch.__version__))
    # VAE full size, B=512: one capped epoch (first `steps` batches of the real epoch order)
    mod = ref_harness.load("vae")
    steps = 12
    loaders = ref_harness.synthetic_loaders(512, n_train=512 * steps, n_val=FULL["n_val"],
                                            n_test=FULL["n_test"], image_shape=FULL["image_shape"])
    torch.manual_seed(1234)
    model = mod.VAE(image_size=784, hidden_dim=400, z_dim=20)
    tr = mod.VAETrainer(model, *loaders, viz=False)
    with ref_harness.quiet():
        tr.train(num_epochs=1)
    arrays = {"recon_loss": np.array(tr.recon_loss), "kl_loss": np.array(tr.kl_loss),
              "best_val_loss": np.array(tr.best_val_loss)}
    for k, v in model.state_dict().items():
        arrays["digest:" + k] = digest(v)
    save("vae_full_b512", arrays, dict(variant="vae", cfg=FULL, batch=512, steps=steps,
                                       n_train=512 * steps, train_kw=dict(num_epochs=1),
                                       rng=rng_digest(), torch=torch.__version__))


def gen_ae():
    """ae.py fixtures: small dims with full parameters (2 epochs, ragged last
    batch of the validation pass) and 784-32 at B = 512 with a ragged last training batch."""
    mod = ref_harness.load("ae")
    for name, cfg, hidden, batch, n_train, epochs, full in (
            ("ae_small", SMALL, 8, SMALL["batch"], 150, 2, False),
            ("ae_full_b512", FULL, 32, 512, 512 * 6 + 336, 1, True)):
        loaders = ref_harness.synthetic_loaders(batch, n_train=n_train, n_val=cfg["n_val"],
                                                n_test=cfg["n_test"],
                                                image_shape=cfg["image_shape"])
        torch.manual_seed(1234)
        model = mod.Autoencoder(image_size=cfg["image_size"], hidden_dim=hidden)
        tr = mod.AutoencoderTrainer(model, *loaders, viz=False)
        with ref_harness.quiet():
            tr.train(num_epochs=epochs)
        arrays = {"recon_loss": np.array(tr.recon_loss), "best_val_loss": np.array(tr.best_val_loss)}
        for k, v in model.state_dict().items():
            arrays[("digest:" if full else "param:") + k] = digest(v) if full else v.numpy()
        save(name, arrays, dict(variant="ae", cfg=cfg, hidden=hidden, batch=batch, n_train=n_train,
                                train_kw=dict(num_epochs=epochs), rng=rng_digest(),
                                torch=torch.__version__))


def gen_bir():
    """bir_vae.py fixtures (second half): the oracle of the NEXT row, produced
    ahead of the product.  numpy's global generator (reparameterisation noise) is seeded with 77
    right before the model is built; small dims with full parameters, and 784-400-20 at B = 256."""
    mod = ref_harness.load("bir_vae")
    for name, cfg, batch, n_train, epochs, full in (
            ("bir_small", SMALL, SMALL["batch"], 150, 2, False),
            ("bir_full_b256", FULL, 256, 256 * 6 + 100, 1, True)):
        loaders = ref_harness.synthetic_loaders(batch, n_train=n_train, n_val=cfg["n_val"],
                                                n_test=cfg["n_test"],
                                                image_shape=cfg["image_shape"])
        torch.manual_seed(1234)
        np.random.seed(77)
        model = mod.BIRVAE(image_size=cfg["image_size"], hidden_dim=cfg["hidden_dim"],
                           z_dim=cfg["z_dim"])
        tr = mod.BIRVAETrainer(model, *loaders, viz=False)
        with ref_harness.quiet():
            tr.train(num_epochs=epochs)
        arrays = {"recon_loss": np.array(tr.recon_loss), "mmd_loss": np.array(tr.mmd_loss),
                  "best_val_loss": np.array(tr.best_val_loss)}
        for k, v in model.state_dict().items():
            arrays[("digest:" if full else "param:") + k] = digest(v) if full else v.numpy()
        save(name, arrays, dict(variant="bir", cfg=cfg, batch=batch, n_train=n_train, np_seed=77,
                                train_kw=dict(num_epochs=epochs), rng=rng_digest(),
                                torch=torch.__version__))


def gen_round2():
    """Round-2 fixtures: the 50-step free-running NSGAN B=256 curve of SURVEY.md
    8(d), and a full-size VAE B=512 epoch whose last training batch is the ragged 336
    (50 000 mod 512) -- n_train = 512*3 + 336."""
    tr, model = run_reference("ns", FULL, 256, dict(num_epochs=1), steps_cap=50)
    arrays = {"Glosses": np.array(tr.Glosses), "Dlosses": np.array(tr.Dlosses)}
    for k, v in model.state_dict().items():
        arrays["digest:" + k] = digest(v)
    save("ns_full_b256_50steps", arrays, dict(variant="ns", cfg=FULL, batch=256, steps=50,
                                              train_kw=dict(num_epochs=1), rng=rng_digest(),
                                              torch=torch.__version__))
    # viz=True (ns_gan.py:166-170): one extra compute_noise(36, z) draw per epoch end moves the RNG
    # stream -- the fixture pins losses, parameters and the final generator state with viz on
    mod_name, model_name, trainer_name = port.REFERENCE_NAMES["ns"]
    mod = ref_harness.load(mod_name)
    import tempfile
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:          # the reference writes ../viz/<name>/ relative to cwd
        os.makedirs(os.path.join(tmp, "src"))
        os.chdir(os.path.join(tmp, "src"))
        try:
            loaders = ref_harness.synthetic_loaders(SMALL["batch"], n_train=SMALL["n_train"],
                                                    n_val=SMALL["n_val"], n_test=SMALL["n_test"],
                                                    image_shape=SMALL["image_shape"])
            torch.manual_seed(1234)
            model = getattr(mod, model_name)(image_size=SMALL["image_size"], hidden_dim=SMALL["hidden_dim"],
                                             z_dim=SMALL["z_dim"])
            tr = getattr(mod, trainer_name)(model, *loaders, viz=True)
            with ref_harness.quiet():
                tr.train(num_epochs=3)
        finally:
            os.chdir(cwd)
    arrays = {"Glosses": np.array(tr.Glosses), "Dlosses": np.array(tr.Dlosses)}
    for k, v in model.state_dict().items():
        arrays["param:" + k] = v.numpy()
    save("ns_small_viz", arrays, dict(variant="ns", cfg=SMALL, train_kw=dict(num_epochs=3), viz=True,
                                      rng=rng_digest(), torch=torch.__version__))
    mod = ref_harness.load("vae")
    n_train = 512 * 3 + 336
    loaders = ref_harness.synthetic_loaders(512, n_train=n_train, n_val=FULL["n_val"],
                                            n_test=FULL["n_test"], image_shape=FULL["image_shape"])
    torch.manual_seed(1234)
    model = mod.VAE(image_size=784, hidden_dim=400, z_dim=20)
    tr = mod.VAETrainer(model, *loaders, viz=False)
    with ref_harness.quiet():
        tr.train(num_epochs=2)
    arrays = {"recon_loss": np.array(tr.recon_loss), "kl_loss": np.array(tr.kl_loss),
              "best_val_loss": np.array(tr.best_val_loss)}
    for k, v in model.state_dict().items():
        arrays["digest:" + k] = digest(v)
    save("vae_full_b512_ragged", arrays, dict(variant="vae", cfg=FULL, batch=512, steps=8,
                                              n_train=n_train, train_kw=dict(num_epochs=2),
                                              rng=rng_digest(), torch=torch.__version__))


def gen_vae_viz():
    """vae.py / bir_vae.py with viz=True (vae.py:189-191, bir_vae.py:176-178): sample_images(epoch)
    draws torch.randn(36, z_dim) from the global generator at every epoch end.  The reference's own
    rendering needs torchvision / PIL / IPython's display(); they are replaced by arithmetic-free
    no-ops here (the draw itself is the reference's own line)."""
    import tempfile

    class _Img:
        def save(self, *a, **k):
            pass
    for name, modname, cls, tcls, hist in (("vae_small_viz", "vae", "VAE", "VAETrainer", ("recon_loss", "kl_loss")),
                                           ("bir_small_viz", "bir_vae", "BIRVAE", "BIRVAETrainer",
                                            ("recon_loss", "mmd_loss"))):
        mod = ref_harness.load(modname)
        mod.ToPILImage = lambda *a, **k: (lambda *b, **kk: _Img())
        mod.make_grid = lambda t, *a, **k: t
        mod.display = lambda *a, **k: None
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as tmp:
            os.makedirs(os.path.join(tmp, "src"))
            os.chdir(os.path.join(tmp, "src"))
            try:
                loaders = ref_harness.synthetic_loaders(SMALL["batch"], n_train=150, n_val=SMALL["n_val"],
                                                        n_test=SMALL["n_test"], image_shape=SMALL["image_shape"])
                torch.manual_seed(1234)
                np.random.seed(77)
                model = getattr(mod, cls)(image_size=SMALL["image_size"], hidden_dim=SMALL["hidden_dim"],
                                          z_dim=SMALL["z_dim"])
                tr = getattr(mod, tcls)(model, *loaders, viz=True)
                with ref_harness.quiet():
                    tr.train(num_epochs=3)
            finally:
                os.chdir(cwd)
        arrays = {h: np.array(getattr(tr, h)) for h in hist}
        arrays["best_val_loss"] = np.array(tr.best_val_loss)
        for k, v in model.state_dict().items():
            arrays["param:" + k] = v.numpy()
        save(name, arrays, dict(variant=modname, cfg=SMALL, n_train=150, np_seed=77, viz=True,
                                train_kw=dict(num_epochs=3), rng=rng_digest(), torch=torch.__version__))


def gen_round3():
    """Round-3 fixture: BASELINE.json configs[4]'s NSGAN leg at its real batch,
    NSGAN 784-400-20 B = 1024, 12 free-running D+G steps of the unmodified reference
    (ns_gan.py:94-170) -- loss curves, per-tensor digests, final RNG position."""
    tr, model = run_reference("ns", FULL, 1024, dict(num_epochs=1), steps_cap=12)
    arrays = {"Glosses": np.array(tr.Glosses), "Dlosses": np.array(tr.Dlosses)}
    for k, v in model.state_dict().items():
        arrays["digest:" + k] = digest(v)
    save("ns_full_b1024", arrays, dict(variant="ns", cfg=FULL, batch=1024, steps=12,
                                       train_kw=dict(num_epochs=1), rng=rng_digest(),
                                       torch=torch.__version__))


def gen_round6():
    """Round-6 fixtures (VERDICT r5 "missing" 3): 50-step FREE-RUNNING curves of the unmodified reference for the
    BASELINE configs that only had 12 - 16 steps -- WGAN-GP B = 256 D_steps = 1 (w_gp_gan.py:96-175), LSGAN and NSGAN
    B = 1024 (ls_gan.py:95-171, ns_gan.py:94-170) -- and ONE FULL VAE EPOCH at B = 512 over 50 000 images: 97 full
    batches + the ragged 336 + the validation pass (vae.py:127-191), the epoch bench.py times."""
    for variant, batch, kw in (("wgp", 256, dict(num_epochs=1, D_steps=1)), ("ls", 1024, dict(num_epochs=1)),
                               ("ns", 1024, dict(num_epochs=1))):
        tr, model = run_reference(variant, FULL, batch, kw, steps_cap=50)
        arrays = {"Glosses": np.array(tr.Glosses), "Dlosses": np.array(tr.Dlosses)}
        for k, v in model.state_dict().items():
            arrays["digest:" + k] = digest(v)
        save("%s_full_b%d_50steps" % (variant, batch), arrays,
             dict(variant=variant, cfg=FULL, batch=batch, steps=50, train_kw=kw, rng=rng_digest(),
                  torch=torch.__version__))
    mod = ref_harness.load("vae")
    loaders = ref_harness.synthetic_loaders(512, n_train=50000, n_val=FULL["n_val"], n_test=FULL["n_test"],
                                            image_shape=FULL["image_shape"])
    torch.manual_seed(1234)
    model = mod.VAE(image_size=784, hidden_dim=400, z_dim=20)
    tr = mod.VAETrainer(model, *loaders, viz=False)
    with ref_harness.quiet():
        tr.train(num_epochs=1)
    assert len(tr.recon_loss) == 98
    arrays = {"recon_loss": np.array(tr.recon_loss), "kl_loss": np.array(tr.kl_loss),
              "best_val_loss": np.array(tr.best_val_loss)}
    for k, v in model.state_dict().items():
        arrays["digest:" + k] = digest(v)
    save("vae_full_b512_epoch98", arrays, dict(variant="vae", cfg=FULL, batch=512, steps=98, n_train=50000,
                                               train_kw=dict(num_epochs=1), rng=rng_digest(),
                                               torch=torch.__version__))


if __name__ == "__main__":
    if sys.argv[1:] == ["round6"]:
        torch.set_num_threads(1)
        gen_round6()
    elif sys.argv[1:] == ["round3"]:
        torch.set_num_threads(1)
        gen_round3()
    elif sys.argv[1:] == ["vae_viz"]:
        gen_vae_viz()
    elif sys.argv[1:] == ["round2"]:
        gen_round2()
    elif sys.argv[1:] == ["ae"]:
        gen_ae()                 # only the ae.py fixtures (the others are unchanged)
    elif sys.argv[1:] == ["bir"]:
        gen_bir()
    else:
        main()
        gen_ae()
        gen_bir()
        gen_round2()
        gen_round3()
        gen_round6()
