"""us per iteration of EVERY drop-in trainer at the BASELINE shapes (784-400-20; GANs bs=256, the VAE family
bs=512), through `Trainer.train` with each module's default arguments: one warm-up epoch, then timed epochs.
Finds variants whose step is out of line with NSGAN's (a slow helper kernel shows up here first).
usage: python tools/variant_times.py [comma list of variants] [epochs]"""
import contextlib
import importlib
import io
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (puts generative_models_amd/src on the path, synthetic dataset)

MODS = {"ns": ("ns_gan", "NSGAN"), "mm": ("mm_gan", "MMGAN"), "w": ("w_gan", "WGAN"), "wgp": ("w_gp_gan", "WGPGAN"),
        "ls": ("ls_gan", "LSGAN"), "ra": ("ra_gan", "RaNSGAN"), "fisher": ("fisher_gan", "FisherGAN"),
        "f": ("f_gan", "fGAN"), "dra": ("dra_gan", "DRAGAN"), "be": ("be_gan", "BEGAN"),
        "info": ("info_gan", "InfoGAN"), "vae": ("vae", "VAE"), "ae": ("ae", "Autoencoder"),
        "bir": ("bir_vae", "BIRVAE")}


def main():
    which = sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] != "all" else list(MODS)
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    ds = bench.synthetic_dataset()
    for v in which:
        mod_name, cls = MODS[v]
        mod = importlib.import_module(mod_name)
        B = 512 if v in ("vae", "ae", "bir") else 256
        mk = lambda: torch.utils.data.DataLoader(ds, batch_size=B, shuffle=True)
        torch.manual_seed(1234)
        kw = dict(image_size=bench.IMG, hidden_dim=bench.HID, z_dim=bench.Z)
        if v == "info":
            kw.update(disc_dim=10, cont_dim=10)
        if v == "ae":
            kw = dict(image_size=bench.IMG, hidden_dim=32)          # ae.py:72 default
        tkw = dict(method="jensen_shannon") if v == "f" else {}
        try:
            model = getattr(mod, cls)(**kw)
            vds = torch.utils.data.TensorDataset(ds.tensors[0][:10000], ds.tensors[1][:10000])
            vl = torch.utils.data.DataLoader(vds, batch_size=B, shuffle=True)
            tr = getattr(mod, cls + "Trainer")(model, mk(), vl, vl, viz=False)
            steps = len(tr.train_iter)
            with contextlib.redirect_stdout(io.StringIO()):
                tr.train(1, **tkw)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                tr.train(epochs, **tkw)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            print("%-7s bs=%d  %8.1f us / iteration   (%d iterations, engine %s)"
                  % (v, B, dt / (epochs * steps) * 1e6, epochs * steps,
                     type(getattr(tr, "_engine", None)).__name__), flush=True)
        except Exception as e:                       # noqa: BLE001
            print("%-7s FAILED: %r" % (v, e), flush=True)


if __name__ == "__main__":
    main()
