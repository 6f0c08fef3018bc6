#!/usr/bin/env python
"""vae.py on the CPU oracle, twice: torch's fp32 exp against exp evaluated in fp64 and rounded once
(oracle/port.py rounded_exp).  Both are legitimate fp32 evaluations of the reference; how far do they drift apart?
Evidence for tests/test_gpu_trainers.py::test_vae_reference_default_batch_full_epoch.  Test infrastructure.

    python tools/vae_exp_rounding_cpu.py > profiles/r05_vae_exp_rounding.json
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import port  # noqa: E402


def run(rounded, B, n, epochs):
    ctx = port.rounded_exp() if rounded else __import__("contextlib").nullcontext()
    with ctx:
        ld = port.synthetic_loaders(B, n_train=n, n_val=1000, n_test=200, image_shape=(1, 28, 28))
        m = port.build("vae", 784, 400, 20)
        tr = port.VAEPort(m, *ld)
        tr.train(epochs)
    return np.array(tr.recon_loss), np.array(tr.kl_loss), m


def main():
    out = {"what": "oracle/port.py VAE 784-400-20, torch %s CPU: stock torch.exp vs exp in fp64 rounded once; relative loss "
                   "differences |a - b| / max(1, |b|) at a few steps, parameter differences at the end" % torch.__version__, "cases": []}
    for B, n, epochs in ((100, 50000, 1), (512, 512 * 3 + 336, 3), (512, 50000, 1)):
        r0, k0, m0 = run(False, B, n, epochs)
        r1, k1, m1 = run(True, B, n, epochs)
        er = np.abs(r0 - r1) / np.maximum(1, np.abs(r1))
        ek = np.abs(k0 - k1) / np.maximum(1, np.abs(k1))
        steps = [s for s in (0, 10, 20, 30, 40, 49, 99, 200, 499) if s < len(k0)]
        out["cases"].append({"B": B, "n_train": n, "epochs": epochs, "batches": len(k0), "steps": steps,
                             "kl": [float(k1[s]) for s in steps], "kl_rel_diff": [float(ek[s]) for s in steps],
                             "recon_rel_diff_max": float(er.max()), "kl_rel_diff_max": float(ek.max()),
                             "params": {k: {"max": float((a - b).abs().max()), "mean": float((a - b).abs().mean())}
                                        for (k, a), (_, b) in zip(m0.state_dict().items(), m1.state_dict().items())}})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
