#!/usr/bin/env python
"""Where does the VAE's HIP path leave the CPU oracle?  vae.py at 784-400-20, batch B, n images: one epoch of the
product under a set of environment toggles (each removes one launch fusion) against oracle/port.py, per-step loss
errors printed at a few steps.  Test infrastructure (uses oracle/); run on the GPU box.
    python tools/vae_divergence_probe.py [B] [n_train]"""
import contextlib
import io
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "generative_models_amd", "src"))


def product(B, n):
    import vae
    from oracle import port
    ld = port.synthetic_loaders(B, n_train=n, n_val=1000, n_test=200, image_shape=(1, 28, 28))
    torch.manual_seed(1234)
    model = vae.VAE(image_size=784, hidden_dim=400, z_dim=20)
    tr = vae.VAETrainer(model, *ld, viz=False)
    tr.use_graph = os.environ.get("PROBE_EAGER") != "1"
    with contextlib.redirect_stdout(io.StringIO()):
        tr.train(num_epochs=1)
    torch.cuda.synchronize()
    return np.array(tr.recon_loss), np.array(tr.kl_loss), {k: v.cpu() for k, v in model.state_dict().items()}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
    if os.environ.get("PROBE_CHILD") == "1":
        r, k, sd = product(B, n)
        torch.save({"r": r, "k": k, "sd": sd}, os.environ["PROBE_OUT"])
        return
    from oracle import port
    ld = port.synthetic_loaders(B, n_train=n, n_val=1000, n_test=200, image_shape=(1, 28, 28))
    om = port.build("vae", 784, 400, 20)
    o = port.VAEPort(om, *ld)
    o.train(1)
    orr, ok = np.array(o.recon_loss), np.array(o.kl_loss)
    osd = om.state_dict()
    # (round 5 ran this over every fusion toggle of the VAE engine -- same drift under each, profiles/r05_experiments.md
    # section 4; round 6 removed those environment switches, the rows below that name one now repeat the default)
    cases = [("default", {}), ("eager", {"PROBE_EAGER": "1"}), ("no bwd_mid", {"GM_VAE_BWD_MID": "0"}),
             ("no reparam_fwd", {"GM_VAE_FUSE_REPARAM_FWD": "0"}), ("no reparam_bwd", {"GM_VAE_FUSE_REPARAM_BWD": "0"}),
             ("no sqerr", {"GM_VAE_FUSE_SQERR": "0"}), ("no prefetch", {"GM_VAE_PREFETCH_GATHER": "0"}),
             ("no finalize_in_dw", {"GM_VAE_FINALIZE_IN_DW": "0"}), ("no pair", {"GM_PAIR_DW": "0"}),
             ("no fused adam", {"GM_FUSE_ADAM": "0"}), ("fp32 dataset", {"GM_PACKED": "0"}),
             ("graph_iters 1", {"GM_GRAPH_ITERS": "1"})]
    steps = [0, 1, 2, 5, 10, 20, 31, 32, 33, 40, 49, 63, 64, 65, 99, 200, len(orr) - 1]
    steps = [s for s in steps if s < len(orr)]
    print("VAE B=%d n=%d: %d batches; columns: |kl - oracle| / max(1, |oracle|) at steps %s; then max recon err, max param dev" % (B, n, len(orr), steps))
    for name, env in cases:
        out = "/tmp/probe_%d.pt" % os.getpid()
        e = dict(os.environ, PROBE_CHILD="1", PROBE_OUT=out, **env)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), str(B), str(n)], env=e, capture_output=True, text=True)
        if p.returncode != 0:
            print("%-18s FAILED: %s" % (name, p.stderr.strip().splitlines()[-1] if p.stderr.strip() else "?"))
            continue
        d = torch.load(out, weights_only=False)
        ek = np.abs(d["k"] - ok) / np.maximum(1, np.abs(ok))
        er = np.abs(d["r"] - orr) / np.maximum(1, np.abs(orr))
        pd = max(float((d["sd"][k] - osd[k]).abs().max()) for k in osd)
        first = int(np.argmax(ek > 1e-5)) if (ek > 1e-5).any() else -1
        print("%-18s %s | recon %.1e | params %.1e | first kl err > 1e-5 at step %d" % (name, " ".join("%.0e" % ek[s] for s in steps), er.max(), pd, first))


if __name__ == "__main__":
    main()
