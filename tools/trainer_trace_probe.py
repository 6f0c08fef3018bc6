"""Host / GPU timeline of Trainer.train(epochs) through the drop-in module (GANEngine's GM_TRACE_RUN marks): where an
epoch of 196 iterations spends the microseconds it has over 196 steady steps."""
import contextlib
import importlib.util
import io
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gm_bench_tp", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
import ns_gan  # noqa: E402

ds = bench.synthetic_dataset()
torch.manual_seed(1234)
model = ns_gan.NSGAN(image_size=bench.IMG, hidden_dim=bench.HID, z_dim=bench.Z)
tr = ns_gan.NSGANTrainer(model, torch.utils.data.DataLoader(ds, batch_size=bench.B_PER_GPU, shuffle=True), None, None, viz=False)
with contextlib.redirect_stdout(io.StringIO()):
    tr.train(1)
    torch.cuda.synchronize()
    eng = tr._engine
    for trace in (False, True):
        os.environ["GM_TRACE_RUN"] = "1" if trace else "0"       # (configure() reads it)
        t0 = time.perf_counter()
        tr.train(6)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print("trace=%s: train(6) %.1f us per step" % (trace, (t1 - t0) / (6 * 196) * 1e6), file=sys.stderr)
ev = eng._trace
runs = [i for i, e in enumerate(ev) if e[0] == "run"]
for n, i in enumerate(runs):
    j = runs[n + 1] if n + 1 < len(runs) else len(ev)
    seg = ev[i:j]
    tr0 = seg[0][2]
    ends = [e for e in seg if e[0] == "gpu_piece_ends_us"]
    launched = [(e[1], round((e[2] - tr0) * 1e6)) for e in seg if e[0] == "launched"]
    got = [(e[1], round((e[2] - tr0) * 1e6)) for e in seg if e[0] == "got"]
    print("run %d entered at %+d us after train() start; pieces launched (iteration, host us): %s" % (n, (tr0 - t0) * 1e6, launched), file=sys.stderr)
    print("    draws ready (iteration, host us): %s" % got, file=sys.stderr)
    if ends:
        pe = ends[0][1]
        its = [a for a, _ in launched]
        prev_t, prev_i = 0.0, seg[0][1]
        per = []
        for t_, i_ in zip(pe, its):
            per.append((i_ - prev_i, round(t_ - prev_t, 1), round((t_ - prev_t) / max(1, i_ - prev_i), 2)))
            prev_t, prev_i = t_, i_
        print("    GPU: pieces (iterations, us, us per iteration): %s; total %.1f us = %.2f per step; returned to host at %d us"
              % (per, pe[-1], pe[-1] / 196, (ends[0][2] - tr0) * 1e6), file=sys.stderr)
