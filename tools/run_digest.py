"""Digest of a short deterministic NSGAN run on the HIP engine (losses + parameters), for same-bits A/B checks between two
builds of the library (GM_LIB_PATH).  usage: run_digest.py [variant] [batch] [iters]"""
import sys, os, hashlib, importlib.util
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "src"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from generative_models_amd import engine as gm_engine  # noqa: E402
variant = sys.argv[1] if len(sys.argv) > 1 else "ns"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 64
mod, cls = {"ns": ("ns_gan", "NSGAN"), "ls": ("ls_gan", "LSGAN"), "wgp": ("w_gp_gan", "WGPGAN")}[variant]
m = __import__(mod)
dev = torch.device("cuda:0")
ds = bench.synthetic_dataset()
torch.manual_seed(1234)
model = getattr(m, cls)(image_size=bench.IMG, hidden_dim=bench.HID, z_dim=bench.Z)
tr = getattr(m, cls + "Trainer")(model, torch.utils.data.DataLoader(ds, batch_size=B, shuffle=True), None, None, viz=False)
data = ds.tensors[0].reshape(bench.N_TRAIN, -1).to(dev).contiguous()
eng = gm_engine.GANEngine(variant, tr.model, data, B, dev)
eng.configure(iters, 2e-4, 2e-4, 1)
eng.run(iters)
torch.cuda.synchronize()
h = hashlib.sha256()
for t in (eng.lossG, eng.lossD):
    h.update(t.detach().cpu().numpy().tobytes())
for p in tr.model.parameters():
    h.update(p.detach().cpu().numpy().tobytes())
print(variant, B, iters, h.hexdigest()[:24], float(eng.lossG.detach().cpu()[:iters].mean()))
