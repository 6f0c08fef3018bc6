#!/usr/bin/env python
"""Throughput of the BASELINE.json configs other than the headline one, on ONE MI355X, through the
drop-in trainers (`Trainer.train(num_epochs=1)` exactly as a reference user would call it):
  config 3: WGAN-GP bs=256 (D_steps=1 as in w_gp_gan.py __main__, and D_steps=5)
  config 4: VAE bs=512, one epoch incl. the ragged last batch and the validation pass
  config 5 (1-GPU leg): NSGAN and LSGAN bs=1024
Prints one JSON line per config (images/sec = real images consumed by D steps / wall time).
Also times the CPU oracle (oracle/port.py) on a short sample of the same config when --cpu is given."""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "generative_models_amd", "src"))

N_TRAIN = 50000


def loaders(batch, n_val=10000):
    torch.manual_seed(3435)
    mk = lambda n: torch.utils.data.DataLoader(
        torch.utils.data.TensorDataset(torch.bernoulli(torch.full((n, 1, 28, 28), 0.1307)),
                                       torch.zeros(n, dtype=torch.int64)),
        batch_size=batch, shuffle=True)
    return mk(N_TRAIN), mk(n_val), mk(2048)


def run(name, module, model_cls, trainer_cls, batch, train_kw, epochs=6):
    import importlib
    mod = importlib.import_module(module)
    ld = loaders(batch)
    torch.manual_seed(1234)
    if module == "vae":
        model = getattr(mod, model_cls)()
    elif module == "info_gan":
        model = getattr(mod, model_cls)(784, 400, 20, 10, 10)
    else:
        model = getattr(mod, model_cls)(784, 400, 20)
    tr = getattr(mod, trainer_cls)(model, *ld)
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(1, **train_kw)                      # warm-up epoch (graph capture, clocks)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.train(epochs, **train_kw)
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    d_steps = train_kw.get("D_steps", 5 if module == "w_gan" else 1)
    if module == "vae":
        images = epochs * N_TRAIN
        steps = epochs * len(ld[0])
    else:
        steps = epochs * -(-len(ld[0]) // d_steps)
        images = steps * d_steps * batch
    return {"config": name, "images_per_sec": images / dt, "ms_per_iteration": dt / steps * 1e3,
            "batch": batch, "iterations": steps, "wall_s": dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    cfgs = [
        ("NSGAN bs=256", "ns_gan", "NSGAN", "NSGANTrainer", 256, {}),
        ("NSGAN bs=1024", "ns_gan", "NSGAN", "NSGANTrainer", 1024, {}),
        ("LSGAN bs=1024", "ls_gan", "LSGAN", "LSGANTrainer", 1024, {}),
        ("WGAN-GP bs=256 D_steps=1", "w_gp_gan", "WGPGAN", "WGPGANTrainer", 256, {"D_steps": 1}),
        ("WGAN-GP bs=256 D_steps=5", "w_gp_gan", "WGPGAN", "WGPGANTrainer", 256, {"D_steps": 5}),
        ("VAE bs=512 (train epoch + 10k-image validation pass)", "vae", "VAE", "VAETrainer", 512, {}),
        ("DRAGAN bs=256 D_steps=1", "dra_gan", "DRAGAN", "DRAGANTrainer", 256, {"D_steps": 1}),
        ("BEGAN bs=256", "be_gan", "BEGAN", "BEGANTrainer", 256, {}),
        ("InfoGAN bs=256", "info_gan", "InfoGAN", "InfoGANTrainer", 256, {}),
        ("WGAN bs=256 D_steps=5", "w_gan", "WGAN", "WGANTrainer", 256, {}),
    ]
    for c in cfgs:
        if args.only and args.only not in c[0]:
            continue
        print(json.dumps(run(*c)), flush=True)


if __name__ == "__main__":
    main()
