"""What one hipGraph launch ("piece") of the NSGAN bs=256 engine costs beyond its iterations: warm runs cut into pieces
of 4 / 8 / 16 / 32 (/ 64) iterations, GPU time of every piece from timing events behind each launch (GM_TRACE_RUN's
marks), least-squares fit  piece_us = a + s * iterations  (the first piece of a run is left out: it holds the run's start).
usage: piece_cost_probe.py [pieces, e.g. 4,8,16,32,4]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "generative_models_amd", "src"))
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location("gm_bench_pc", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
import ns_gan  # noqa: E402
from generative_models_amd import engine as gm_engine  # noqa: E402

pieces = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4,8,16,32,4").split(",")]
n = sum(pieces)
dev = torch.device("cuda:0")
ds = bench.synthetic_dataset()
torch.manual_seed(1234)
model = ns_gan.NSGAN(image_size=bench.IMG, hidden_dim=bench.HID, z_dim=bench.Z)
tr = ns_gan.NSGANTrainer(model, torch.utils.data.DataLoader(ds, batch_size=256, shuffle=True), None, None, viz=False)
data = ds.tensors[0].reshape(bench.N_TRAIN, -1).to(dev).contiguous()
eng = gm_engine.GANEngine("ns", tr.model, data, 256, dev)
reps = 8
W = 2 * eng.R if hasattr(eng, "R") else 256
total = 4096
eng.configure(total, 2e-4, 2e-4, 1)
W = 2 * eng.R
assert eng.R % n == 0, (eng.R, n)
eng.run(W, it_start=0, horizon=total)
torch.cuda.synchronize()
eng._plan = lambda it, k, cold: list(pieces)
eng._trace = []
it = W
for r in range(reps):
    eng.run(n, it_start=it, horizon=total)
    it += n
torch.cuda.synchronize()
ends = [e[1] for e in eng._trace if e[0] == "gpu_piece_ends_us"]
rows = []
for pe in ends[2:]:                                  # (two runs to settle)
    prev = 0.0
    for j, (k, t) in enumerate(zip(pieces, pe)):
        if j > 0:
            rows.append((k, t - prev))
        prev = t
ks = np.array([r[0] for r in rows], dtype=np.float64)
ts = np.array([r[1] for r in rows], dtype=np.float64)
A = np.stack([np.ones_like(ks), ks], axis=1)
(a, s), res, _, _ = np.linalg.lstsq(A, ts, rcond=None)
by = {k: round(float(ts[ks == k].mean()), 1) for k in sorted(set(pieces[1:]))}
print("env %s pieces %s: piece_us = %.1f + %.2f * iterations  (mean piece us by size: %s; max |residual| %.1f us; first-piece us: %s)"
      % ({k: v for k, v in os.environ.items() if k.startswith("GM_")}, pieces, a, s, by,
         float(np.abs(A @ np.array([a, s]) - ts).max()), [round(pe[0], 1) for pe in ends[2:]]))
