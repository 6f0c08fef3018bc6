"""One-off (round 6): gemm16_k32_fwd_kernel against the 16-wave forward on the same inputs, bit for bit.
Run twice -- with and without GM_TMP_K32_OFF=1 (the temporary experiment knob of that round) -- then `compare`."""
import sys, hashlib, json
import torch
from generative_models_amd import ops

def run(tag):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    out = {}
    for (M, K, N) in [(512, 20, 400), (256, 20, 400), (37, 20, 50), (2048, 20, 400), (100, 16, 400), (336, 32, 404), (64, 4, 32), (1, 20, 400), (512, 28, 784)]:
        for act in ("id", "relu", "sigmoid"):
            for bias in (True, False):
                x = torch.randn(M, K, generator=g).to(dev)
                W = (torch.randn(N, K, generator=g) * 0.3).to(dev)
                b = (torch.randn(N, generator=g) * 0.1).to(dev) if bias else None
                y = torch.empty(M, N, device=dev)
                ops.linear_fwd(x, W, b, y, act)
                torch.cuda.synchronize()
                out["%d_%d_%d_%s_%d" % (M, K, N, act, bias)] = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()
    json.dump(out, open("gpurun_out/r06_o/bits_%s.json" % tag, "w"))
    print(tag, len(out), "cases")

if sys.argv[1] == "compare":
    a = json.load(open("gpurun_out/r06_o/bits_k32.json")); b = json.load(open("gpurun_out/r06_o/bits_w16.json"))
    bad = [k for k in a if a[k] != b[k]]
    print("cases", len(a), "differing", bad)
    sys.exit(1 if bad else 0)
else:
    run(sys.argv[1])
