"""Round 6: WHERE do the three full-size cases whose free-running parameters end more than 1e-5
from the oracle's differ?  For f-GAN hellinger / pearson (6 steps, B = 256) and the VAE at B = 512 (3 epochs of
3 x 512 + 336) this runs the oracle with hooks that record, per step, every hidden unit's smallest |pre-activation|
(a relu at its kink) and every parameter element's |gradient|, then the HIP path, and writes the per-element
deviations next to that evidence (gpurun_out/param_attribution/*.npz) so the element masks the tests assert can be
designed from data.  GPU tool; imports the oracle as the checker (test infrastructure)."""
import contextlib
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_trainers as T  # noqa: E402
from oracle import port  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out", "param_attribution")
os.makedirs(OUT, exist_ok=True)


def hook_first_layers(model, layers, store):
    for name in layers:
        mod = model
        for part in name.split("."):
            mod = getattr(mod, part)

        def fn(m_, i_, out, name=name):
            if model.training:
                store.setdefault(name, []).append(out.detach().abs().min(dim=0).values.numpy().copy())
        mod.register_forward_hook(fn)


def gan_case(method):
    variant, kw, steps = "f", dict(num_epochs=1, method=method), 6

    class Capped(torch.utils.data.DataLoader):
        def __len__(self):
            return steps

    def loaders():
        ld = port.synthetic_loaders(256, n_train=50000, n_val=256, n_test=256, image_shape=(1, 28, 28))
        return (Capped(ld[0].dataset, batch_size=256, shuffle=True),) + ld[1:]
    ld = loaders()
    o_model = port.build(variant, 784, 400, 20)
    pre, gabs = {}, {}
    hook_first_layers(o_model, ["D.linear", "G.linear"], pre)

    def tap(kind, tr_, info):
        if kind in ("D", "G"):
            for n, p in getattr(tr_.model, kind).named_parameters():
                gabs.setdefault("%s.%s" % (kind, n), []).append(p.grad.detach().abs().numpy().copy())
    o = port.GANPort(variant, o_model, ld[0], method=method, tap=tap)
    o.train(num_epochs=1)
    tr, model = T.build_product(variant, T.FULLCFG, 256, loaders=loaders())
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(**kw)
    torch.cuda.synchronize()
    save("f_" + method, model, o_model, pre, gabs)


def vae_case():
    import vae
    n_train = 512 * 3 + 336
    mk = lambda: port.synthetic_loaders(512, n_train=n_train, n_val=512, n_test=512, image_shape=(1, 28, 28))
    ld0 = mk()
    o_model = port.build("vae", 784, 400, 20)
    pre, gabs = {}, {}
    hook_first_layers(o_model, ["encoder.linear", "decoder.linear"], pre)

    def tap(kind, tr_, info):
        for n, p in tr_.model.named_parameters():
            gabs.setdefault(n, []).append(p.grad.detach().abs().numpy().copy())
    o = port.VAEPort(o_model, *ld0, tap=tap)
    o.train(3)
    ld = mk()
    torch.manual_seed(1234)
    model = vae.VAE(image_size=784, hidden_dim=400, z_dim=20)
    tr = vae.VAETrainer(model, *ld, viz=False)
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(num_epochs=3)
    torch.cuda.synchronize()
    save("vae_b512_ragged", model, o_model, pre, gabs)


def save(tag, model, o_model, pre, gabs):
    out = {}
    for (k, a), (_, b) in zip(model.state_dict().items(), o_model.state_dict().items()):
        out["dev/" + k] = (a.cpu() - b).numpy()
    for k, v in pre.items():
        out["pre/" + k] = np.stack(v)                       # [forward calls in training mode, H]: min over rows of |pre|
    for k, v in gabs.items():
        v = np.stack(v)
        out["gmin/" + k], out["gmax/" + k], out["glast/" + k] = v.min(0), v.max(0), v[-1]
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **out)
    print(tag)
    for k in out:
        if k.startswith("dev/"):
            d = np.abs(out[k])
            print("   %-28s max %.3e  n>1e-5 %d of %d" % (k[4:], d.max(), int((d > 1e-5).sum()), d.size))
    for k in out:
        if k.startswith("pre/"):
            p = out[k]
            print("   %-28s min |pre| over the run %.3e; units with min <= 2e-7: %s" % (
                k, p.min(), sorted(set(np.nonzero(p <= 2e-7)[1].tolist()))))


if __name__ == "__main__":
    which = sys.argv[1:] or ["hellinger", "pearson", "vae"]
    for w in which:
        vae_case() if w == "vae" else gan_case(w)
