#!/usr/bin/env python
"""The unmodified reference trainers (/root/reference/src/*.py through oracle/ref_harness.py) timed beside their
restatement oracle/port.py ON THE SAME HOST, same synthetic data, same thread counts: what `cpu_baseline.kind = "port"`
in a bench line from the GPU box (where the reference cannot travel) stands for.  Test infrastructure.

    python tools/cpu_reference_vs_port.py > profiles/r05_cpu_reference_vs_port.json

as-written: Trainer.train as the reference runs it (ns_gan.py:94-170: a new DataLoader iterator -- a 50 000-entry
reshuffle -- per process_batch call); compute-only: process_batch returns one pre-fetched batch.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import port, ref_harness  # noqa: E402

N, IMG, HID, Z = 50000, 784, 400, 20


def loaders(B, n_train=N):
    return ref_harness.synthetic_loaders(B, n_train=n_train, n_val=B, n_test=B)


class Capped(torch.utils.data.DataLoader):
    cap = 1

    def __len__(self):
        return self.cap


def capped(ld, B, nb):
    c = Capped(ld.dataset, batch_size=B, shuffle=True)
    c.cap = nb
    return c


def time_gan(which, variant, B, steps, compute_only, kw):
    tr_i, va_i, te_i = loaders(B)
    d_steps = kw.get("D_steps", 1)
    tr_i = capped(tr_i, B, steps * d_steps)
    if which == "reference":
        mod_name, model_name, trainer_name = port.REFERENCE_NAMES[variant]
        mod = ref_harness.load(mod_name)
        torch.manual_seed(1234)
        model = getattr(mod, model_name)(image_size=IMG, hidden_dim=HID, z_dim=Z)
        tr = getattr(mod, trainer_name)(model, tr_i, va_i, te_i, viz=False)
    else:
        model = port.build(variant, IMG, HID, Z)
        tr = port.GANPort(variant, model, tr_i)
    if compute_only:
        if which == "reference":                # ns_gan.py:222: process_batch(self, iterator)
            fixed = tr.process_batch(tr.train_iter)
            tr.process_batch = lambda iterator: fixed
        else:
            fixed = tr.process_batch()
            tr.process_batch = lambda: fixed
    run = lambda: tr.train(num_epochs=1, **kw)
    with ref_harness.quiet():
        tr_i.cap = 3 * d_steps
        run()                                   # warm
        tr_i.cap = steps * d_steps
        t0 = time.perf_counter()
        run()
        dt = time.perf_counter() - t0
    return dt / steps


def time_vae(which, B, steps):
    tr_i, va_i, te_i = loaders(B, n_train=B * steps)
    if which == "reference":
        mod = ref_harness.load("vae")
        torch.manual_seed(1234)
        model = mod.VAE(image_size=IMG, hidden_dim=HID, z_dim=Z)
        tr = mod.VAETrainer(model, tr_i, va_i, te_i, viz=False)
    else:
        model = port.build("vae", IMG, HID, Z)
        tr = port.VAEPort(model, tr_i, va_i, te_i)
    with ref_harness.quiet():
        tr.train(num_epochs=1)
        t0 = time.perf_counter()
        tr.train(num_epochs=1)
        dt = time.perf_counter() - t0
    return dt / steps                            # (incl. one B-row validation batch per epoch on both sides)


def main():
    assert ref_harness.available(), "needs the reference at %s" % ref_harness.REFERENCE_ROOT
    ncpu = os.cpu_count() or 1
    out = {"host_cores": ncpu, "torch": torch.__version__, "reference": ref_harness.REFERENCE_ROOT,
           "what": "seconds per D+G iteration (per training batch for the VAE), N = 50000 synthetic 28x28 images", "rows": []}
    cases = [("ns_b256", "ns", 256, 60, {}), ("wgp_b256_d1", "wgp", 256, 40, {"D_steps": 1}),
             ("wgp_b256_d5", "wgp", 256, 12, {"D_steps": 5}), ("ns_b1024", "ns", 1024, 20, {})]
    for threads in sorted({min(8, ncpu), min(16, ncpu), min(4, ncpu)}):
        torch.set_num_threads(threads)
        for name, variant, B, steps, kw in cases:
            for compute_only in (False, True):
                r = time_gan("reference", variant, B, steps, compute_only, kw)
                p = time_gan("port", variant, B, steps, compute_only, kw)
                out["rows"].append({"config": name, "threads": threads, "mode": "compute-only" if compute_only else "as-written",
                                    "reference_ms": round(r * 1e3, 3), "port_ms": round(p * 1e3, 3), "port_over_reference": round(p / r, 4),
                                    "reference_img_s": round(B / r, 1), "port_img_s": round(B / p, 1), "steps": steps})
                print(out["rows"][-1], file=sys.stderr, flush=True)
        r, p = time_vae("reference", 512, 30), time_vae("port", 512, 30)
        out["rows"].append({"config": "vae_b512", "threads": threads, "mode": "as-written", "reference_ms": round(r * 1e3, 3),
                            "port_ms": round(p * 1e3, 3), "port_over_reference": round(p / r, 4),
                            "reference_img_s": round(512 / r, 1), "port_img_s": round(512 / p, 1), "steps": 30})
        print(out["rows"][-1], file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
