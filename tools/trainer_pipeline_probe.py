"""GPU timeline of Trainer.train(epochs) with the epochs enqueued ahead (no synchronisation inside run()): duration of every
graph launch and the idle time between consecutive launches, from timing events evaluated after the run."""
import contextlib
import importlib.util
import io
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gm_bench_tpp", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
import ns_gan  # noqa: E402

os.environ["GM_TRACE_RUN"] = "1"
os.environ["GM_TRACE_NOSYNC"] = "1"
ds = bench.synthetic_dataset()
torch.manual_seed(1234)
model = ns_gan.NSGAN(image_size=bench.IMG, hidden_dim=bench.HID, z_dim=bench.Z)
tr = ns_gan.NSGANTrainer(model, torch.utils.data.DataLoader(ds, batch_size=bench.B_PER_GPU, shuffle=True), None, None, viz=False)
with contextlib.redirect_stdout(io.StringIO()):
    tr.train(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train(6)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
eng = tr._engine
print("R %d graph_iters %d: train(6) %.1f us per step" % (eng.R, eng.graph_iters, (t1 - t0) / (6 * 196) * 1e6))
ev = eng._trace
runs = [e for e in ev if e[0] == "gpu_piece_events"]
launched = [e for e in ev if e[0] == "launched"]
base = runs[0][1][0]
li = 0
for n, r in enumerate(runs):
    tev = r[1]
    ts = [base.elapsed_time(e) * 1e3 for e in tev]       # us since the first run's entry event
    its = [launched[li + j][1] for j in range(len(tev) - 1)]     # iteration each piece ends at
    li += len(tev) - 1
    durations = [round(ts[j] - ts[j - 1], 1) for j in range(1, len(ts))]
    print("run %d: entry event at %.0f us, piece ends at %s, piece durations %s, iterations up to %s"
          % (n, ts[0], [round(x) for x in ts[1:]], durations, its))
