"""Per-wave timeline of the GEMM launches of ONE training iteration inside the real step.

Runs the fused engine on a build of the library whose GEMM kernels record time stamps (gm_gemm.hip under -DGM_STAMPS,
generative_models_amd/libgm_hip_stamps.so: `python -c "from generative_models_amd import _build; _build.build_stamps()"`):
after a warm-up a fresh graph of `--iters` iterations is captured (its launches get stamp slots 0, 1, ...), replayed a few
times, the stamp buffer is cleared, and ONE more replay is recorded.  Per launch: first workgroup entry / last workgroup
exit (100 MHz wall clock, all workgroups) -> launch-to-launch gaps; for the probe workgroup: every wave's stamps (shader
cycles and wall clock).  Stamps perturb the kernels a little (an s_memtime pair + a scheduling barrier per stamp): the
step under the probe is printed next to the un-instrumented one.

usage: wave_timeline.py [--variant ns] [--batch 256] [--iters 2] [--block N] [--out file.md]"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="ns")
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--iters", type=int, default=2)
ap.add_argument("--block", type=int, default=-1, help="probe tile (linear tile index of every launch; default 57)")
ap.add_argument("--out", default=None)
args = ap.parse_args()
LIB = os.path.join(ROOT, "generative_models_amd", "libgm_hip_stamps.so")
assert os.path.isfile(LIB), "build it first: _build.build_stamps()"
os.environ["GM_LIB_PATH"] = LIB
os.environ["GM_GRAPH_ITERS"] = str(args.iters)      # graphs of 1, 2, .. iters iterations: their launches fit the stamp slots
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "generative_models_amd", "src"))
import torch  # noqa: E402
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location("gm_bench_tl", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
from generative_models_amd import _lib, engine as gm_engine  # noqa: E402

lib = _lib.load()
lay = (ctypes.c_int * 6)()
lib.gm_stamps_layout(lay)
WAVES, IDS, HDR, SLOT_WORDS, SLOTS, MAXB = list(lay)
mod, cls = {"ns": ("ns_gan", "NSGAN"), "ls": ("ls_gan", "LSGAN"), "wgp": ("w_gp_gan", "WGPGAN")}[args.variant]
m = __import__(mod)
dev = torch.device("cuda:0")
ds = bench.synthetic_dataset()
torch.manual_seed(1234)
model = getattr(m, cls)(image_size=bench.IMG, hidden_dim=bench.HID, z_dim=bench.Z)
tr = getattr(m, cls + "Trainer")(model, torch.utils.data.DataLoader(ds, batch_size=args.batch, shuffle=True), None, None, viz=False)
data = ds.tensors[0].reshape(bench.N_TRAIN, -1).to(dev).contiguous()
eng = gm_engine.GANEngine(args.variant, tr.model, data, args.batch, dev)
total = 2048
buf = torch.zeros(SLOTS * SLOT_WORDS, dtype=torch.int64, device=dev)
block = 57 if args.block < 0 else args.block
# BEFORE the graphs are captured: the host numbers the GEMM launches from 0 on in issue order (graph of 1 iteration
# first, then 2, ...: (2 * iters - 1) x launches-per-iteration slots in all)
assert lib.gm_stamps_set(ctypes.c_void_p(buf.data_ptr()), block) == 0
eng.configure(total, 2e-4, 2e-4, 1)
k = args.iters
eng._plan = lambda it, kk, cold: [k] * (kk // k)
eng.run(32 * k, it_start=0, horizon=total)
torch.cuda.synchronize()
buf.zero_()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
eng.run(k, it_start=32 * k, horizon=total)
e1.record()
torch.cuda.synchronize()
raw, us_iter = buf.cpu().numpy().astype(np.uint64).reshape(SLOTS, SLOT_WORDS), e0.elapsed_time(e1) * 1e3 / k
MODE = {0: "fwd", 1: "dx", 2: "dw", 10: "fwd (LDS tiles)", 11: "dx (LDS tiles)", 22: "dw (LDS-DMA chunks)"}
launches = []
for s in range(SLOTS):
    h = raw[s, :HDR]
    if h[0] == 0:
        continue
    nb = min(int(h[2]), MAXB)
    ed = raw[s, HDR:HDR + 2 * nb].reshape(nb, 2).astype(np.int64)
    ed = ed[(ed[:, 0] > 0) & (ed[:, 1] > 0)]
    launches.append(dict(slot=s, t0=int(ed[:, 0].min()), t1=int(ed[:, 1].max()), grid=int(h[2]), M=int(h[3]), N=int(h[4]), K=int(h[5]),
                         mode=MODE.get(int(h[6]), str(int(h[6]))), threads=int(h[7]),
                         entry_spread=(int(ed[:, 0].max()) - int(ed[:, 0].min())) / 100.0,
                         wg_mean=float((ed[:, 1] - ed[:, 0]).mean()) / 100.0))
launches.sort(key=lambda d: d["t0"])
out = []
P = out.append
P("# Per-wave timeline: %s bs=%d, one %d-iteration graph replay inside the running engine (tools/wave_timeline.py)" % (args.variant, args.batch, args.iters))
P("")
P("Wall clock = s_memrealtime (100 MHz, 10 ns steps) folded over ALL workgroups of a launch (first entry / last exit); "
  "cycles = s_memtime of the probe workgroup's waves.  Iteration under the probe: %.1f us." % us_iter)
P("")
P("| # | launch (GEMM M x N x K as launched) | workgroups | gap before (us) | first entry -> last exit (us) | last workgroup entered after (us) | mean workgroup residence (us) |")
P("|---|---|---|---|---|---|---|")
base = launches[0]["t0"] if launches else 0
for i, L in enumerate(launches):
    gap = (L["t0"] - launches[i - 1]["t1"]) / 100.0 if i else 0.0
    P("| %d | %s %d x %d x %d | %d | %.2f | %.2f | %.2f | %.2f |" % (i, L["mode"], L["M"], L["N"], L["K"], L["grid"], gap, (L["t1"] - L["t0"]) / 100.0,
                                                        L["entry_spread"], L["wg_mean"]))
P("")
NAMES = {0: "entry", 1: "chunk 0 operands in, MFMAs start", 2: "chunk 0 MFMAs retired", 3: "chunk 1 operands in", 4: "chunk 1 MFMAs retired",
         5: "chunk 2 operands in", 6: "chunk 2 MFMAs retired", 7: "chunk 3 operands in", 8: "chunk 3 MFMAs retired",
         9: "reduction loop done", 10: "all waves' partial tiles in LDS (barrier passed)", 11: "images summed", 12: "epilogue stores issued",
         13: "folded head: dS rebuilt in LDS (barrier passed)", 14: "folded head: this wave's partial dots landed",
         15: "folded head: this wave's rows done, barrier next", 16: "operand bases resolved (arguments, slot counters)",
         17: "first chunk's operand loads issued"}
LDS_NAMES = {0: "entry", 1: "2 stages stored to LDS, 4 more requested", 2: "first fragments in registers", 36: "WK partial tiles in LDS (barrier passed)",
             37: "epilogue stores issued"}
for i, L in enumerate(launches):
    st = raw[L["slot"], HDR + 2 * MAXB:].reshape(WAVES, IDS, 2)
    if not st[:, 0, 1].any():
        continue
    lds = "LDS" in L["mode"]
    P("## launch %d: %s %d x %d x %d, probe workgroup's waves (us after the launch's first workgroup entry; [cycles since the wave's entry])" % (i, L["mode"], L["M"], L["N"], L["K"]))
    P("")
    ids = [j for j in range(IDS) if st[:, j, 1].any()]
    waves = [w for w in range(WAVES) if st[w, 0, 1]]
    # effective clock from the longest span of wave 0
    w0 = waves[0]
    last = max(j for j in ids if st[w0, j, 1])
    dw = (int(st[w0, last, 1]) - int(st[w0, 0, 1])) / 100.0
    dc = int(st[w0, last, 0]) - int(st[w0, 0, 0])
    mhz = dc / dw if dw > 0 else 0.0
    P("shader clock over wave %d's span: %.0f MHz.  Workgroup entered %.2f us after the launch's first workgroup." % (w0, mhz, (int(st[w0, 0, 1]) - L["t0"]) / 100.0))
    P("")
    P("| stamp | " + " | ".join("w%d" % w for w in waves) + " |")
    P("|---|" + "---|" * len(waves))
    for j in ids:
        if "DMA" in L["mode"] and 20 <= j < 36:
            nm = "chunk %d %s" % ((j - 20) // 2, "pieces landed in LDS" if j % 2 == 0 else "MFMAs retired")
        else:
            nm = (("stage %d done (barrier passed)" % (j - 3)) if (lds and 3 <= j < 36) else (LDS_NAMES if lds else NAMES).get(j, str(j)))
        row = []
        for w in waves:
            if not st[w, j, 1]:
                row.append("")
                continue
            row.append("%.2f [%d]" % ((int(st[w, j, 1]) - L["t0"]) / 100.0, int(st[w, j, 0]) - int(st[w, 0, 0])))
        P("| %s | " % nm + " | ".join(row) + " |")
    P("")
text = "\n".join(out) + "\n"
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    open(args.out, "w").write(text)
print(text)
