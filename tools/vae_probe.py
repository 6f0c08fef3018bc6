#!/usr/bin/env python
"""Where does a VAE bs=512 epoch spend its wall time?  Counts graph launches / captures per pass."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "generative_models_amd", "src"))
import bench
from generative_models_amd import ops, engine as E

dev = torch.device("cuda", 0)
import vae
from generative_models_amd.trainers import _epoch_order
ds = bench.synthetic_dataset()
if os.environ.get("PROBE_NORAGGED"):
    ds = torch.utils.data.TensorDataset(ds.tensors[0][:97 * 512], ds.tensors[1][:97 * 512])
tl = torch.utils.data.DataLoader(ds, batch_size=512, shuffle=True)
torch.manual_seed(1234)
tr = vae.VAETrainer(vae.VAE(784, 400, 20), tl, tl, tl)
eng = E.VAEEngine(tr.model, dev, use_graph=not os.environ.get("PROBE_EAGER"))
steps = len(tl)
eng.configure(512, int(os.environ.get("PROBE_EPOCHS", "6")) * steps, 1e-3, 1e-5)
tdata = ds.tensors[0].reshape(len(ds), -1).to(dev).contiguous()
launches = []
slow = []
orig = ops.Graph.launch
def counted(self):
    launches.append(1)
    t = time.perf_counter()
    r = orig(self)
    dt = time.perf_counter() - t
    if dt > 2e-3:
        slow.append(("graph launch #%d" % len(launches), dt))
    return r
ops.Graph.launch = counted
def timed(obj, name):
    f = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        dt = time.perf_counter() - t
        if dt > 2e-3:
            slow.append((name, dt))
        return r
    setattr(obj, name, w)
timed(eng, "_draw_chunk"); timed(eng, "_upload_chunk")
_es = torch.cuda.Event.synchronize
def es(self):
    t = time.perf_counter(); r = _es(self); dt = time.perf_counter() - t
    if dt > 2e-3:
        slow.append(("event.synchronize", dt))
    return r
torch.cuda.Event.synchronize = es
for e in range(int(os.environ.get("PROBE_EPOCHS", "6"))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    perm = _epoch_order(tl)
    t1 = time.perf_counter()
    n0 = len(launches)
    eng.run_pass(tdata, perm, True, e * steps)
    t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    if slow:
        print("   slow host calls:", [(n, round(d * 1e3, 1)) for n, d in slow]); del slow[:]
    print("epoch %d: order %.2f ms, run_pass returned after %.2f ms, synced after %.2f ms; %d graph launches, %d graphs cached, use_graph=%s graph_iters=%d"
          % (e, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t1) * 1e3, len(launches) - n0, len(eng.graphs), eng.use_graph, eng.graph_iters), flush=True)
