"""How long after the last kernel ends does torch.cuda.synchronize() return?  Default HIP scheduling flags against
hipDeviceScheduleSpin (hipSetDeviceFlags), and the ROC_ACTIVE_WAIT_TIMEOUT environment knob (set by the caller).
    python tools/sync_latency_probe.py [spin]"""
import ctypes, sys, time
import numpy as np
import torch

hip = ctypes.CDLL("libamdhip64.so")
torch.cuda.init()
x = torch.zeros(1 << 20, device="cuda")
torch.cuda.synchronize()
if "spin" in sys.argv:
    print("hipSetDeviceFlags(hipDeviceScheduleSpin) ->", hip.hipSetDeviceFlags(1))


def sample(work_us, n=200):
    cyc = int(work_us * 100)          # torch.cuda._sleep counts in ~10 ns ticks on this stack (calibrated below)
    out = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if cyc:
            torch.cuda._sleep(cyc)
        else:
            x[:1].add_(1.0)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6))
    return np.median([a for a, b in out]), np.median([b for a, b in out]), np.percentile([b for a, b in out], 90)


e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for work in (0, 100, 1500):
    torch.cuda.synchronize()
    e0.record()
    if work:
        torch.cuda._sleep(int(work * 100))
    else:
        x[:1].add_(1.0)
    e1.record()
    torch.cuda.synchronize()
    gpu_us = e0.elapsed_time(e1) * 1e3
    l, t, p90 = sample(work)
    print("work ~%6.1f us on the GPU: launch call %.1f us, launch + synchronize %.1f us (p90 %.1f)  => %.1f us around the work"
          % (gpu_us, l, t, p90, t - gpu_us))
