"""How far do two CPU runs of the SAME fp32 training loop drift apart when only the rounding inside the linear layers'
GEMMs changes (torch's fp32 kernels vs the same products accumulated in fp64 and rounded once; forward and backward)?  Evidence for the parameter bounds of the full-size GPU parity
tests: the deviations the GPU path shows against the CPU oracle (profiles/r0N_parity_full_size.jsonl) are of the size
two CPU runs show against each other -- Adam divides by sqrt(v) + 1e-8, so an element whose gradient is ~1e-8 and
formed by cancellation turns a last-bit difference into a fraction of a step.  Test infrastructure (uses oracle/).

    python tools/adam_amplification_cpu.py > profiles/r05_adam_amplification_cpu.json
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import port  # noqa: E402


import contextlib

import torch.nn.functional as F


@contextlib.contextmanager
def gemm_mode(wide):
    """wide: every nn.Linear product (and, through autograd, both of its gradients) accumulated in fp64, rounded to
    fp32 once -- another legal fp32 evaluation of the same expressions, like a GPU kernel's summation order."""
    if not wide:
        yield
        return
    orig = F.linear

    def linear(x, w, b=None):
        y = x.double() @ w.double().t()
        if b is not None:
            y = y + b.double()
        return y.float()
    F.linear = linear
    torch.nn.functional.linear = linear
    try:
        yield
    finally:
        F.linear = orig
        torch.nn.functional.linear = orig


def gan(variant, batch, steps, wide, **kw):
    with gemm_mode(wide):
        return _gan(variant, batch, steps, **kw)


def vae(batch, n_train, epochs, wide):
    with gemm_mode(wide):
        return _vae(batch, n_train, epochs)


def _gan(variant, batch, steps, **kw):

    class Capped(torch.utils.data.DataLoader):
        def __len__(self):
            return steps * kw.get("D_steps", 1)
    ld = port.synthetic_loaders(batch, n_train=50000, n_val=256, n_test=256, image_shape=(1, 28, 28))
    it = Capped(ld[0].dataset, batch_size=batch, shuffle=True)
    model = port.build(variant, 784, 400, 20)
    method = kw.pop("method", "jensen_shannon")
    tr = port.GANPort(variant, model, it, method=method)
    tr.train(num_epochs=1, **kw)
    return dict(G=tr.Glosses, D=tr.Dlosses), model


def _vae(batch, n_train, epochs):
    ld = port.synthetic_loaders(batch, n_train=n_train, n_val=batch, n_test=batch, image_shape=(1, 28, 28))
    model = port.build("vae", 784, 400, 20)
    tr = port.VAEPort(model, *ld)
    tr.train(epochs)
    return dict(recon=tr.recon_loss, kl=tr.kl_loss), model


def compare(name, a, b, lr):
    (la, ma), (lb, mb) = a, b
    out = dict(case=name, lr=lr, loss_rel={}, params={})
    for k in la:
        x, y = torch.tensor(la[k], dtype=torch.float64), torch.tensor(lb[k], dtype=torch.float64)
        out["loss_rel"][k] = float(((x - y).abs() / y.abs().clamp(min=1)).max())
    for (k, p), (_, q) in zip(ma.state_dict().items(), mb.state_dict().items()):
        d = (p - q).abs()
        out["params"][k] = dict(max=float(d.max()), mean=float(d.mean()),
                                frac_above_tenth_step=float((d > 0.1 * lr).float().mean()))
    out["param_max"] = max(v["max"] for v in out["params"].values())
    return out


if __name__ == "__main__":
    res = []
    res.append(compare("vae_b512_ragged_12steps", vae(512, 512 * 3 + 336, 3, False), vae(512, 512 * 3 + 336, 3, True), 1e-3))
    for variant, batch, steps, kw, lr in [("ns", 256, 12, {}, 2e-4), ("f", 256, 6, dict(method="hellinger"), 2e-4),
                                          ("f", 256, 6, dict(method="pearson"), 2e-4), ("ns", 1024, 12, {}, 2e-4),
                                          ("wgp", 256, 12, dict(D_steps=1), 1e-4)]:
        name = "%s%s_b%d_%dsteps" % (variant, "_" + kw["method"] if "method" in kw else "", batch, steps)
        res.append(compare(name, gan(variant, batch, steps, False, **dict(kw)), gan(variant, batch, steps, True, **dict(kw)), lr))
    print(json.dumps(dict(what="oracle/port.py, torch %s CPU, fp32 GEMMs vs fp64-accumulated GEMMs rounded once, same seeds" % torch.__version__,
                          results=res), indent=1))
