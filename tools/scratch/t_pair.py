import torch, sys
sys.path.insert(0,'/root/repo')
import torch.nn as nn
from generative_models_amd import ops
from generative_models_amd.engine import FlatParams, _Linear
DEV='cuda'
B,H,I,Z=256,400,784,20
def run(paired):
    torch.manual_seed(11)
    net = nn.Sequential(nn.Linear(Z, H), nn.Linear(H, I))
    fp = FlatParams(net.parameters(), DEV)
    fp.m.normal_().mul_(1e-3); fp.v.uniform_(0.0, 1e-4)
    L1, L2 = _Linear(fp, net[0]), _Linear(fp, net[1])
    dX, Hg = torch.randn(B, I).to(DEV), torch.relu(torch.randn(B, H)).to(DEV)
    dH, z = torch.randn(B, H).to(DEV), torch.randn(B, Z).to(DEV)
    sched = torch.from_numpy(ops.adam_schedule(2e-4, 4)).to(DEV)
    adam = dict(sched=sched, sched_slot=ops.slot(0, 0, 1, 0, 1), clamp=0.0)
    if paired:
        ops.linear_bwd_dw_adam_pair(dict(dA=dX, X=Hg, lin=L2, adam=adam), dict(dA=dH, X=z, lin=L1, adam=adam))
    else:
        ops.linear_bwd_dw_adam(dX, Hg, L2, adam); ops.linear_bwd_dw_adam(dH, z, L1, adam)
    torch.cuda.synchronize()
    segs = [(n, p.data_ptr()) for n, p in net.named_parameters()]
    return [t.clone() for t in (fp.flat, fp.grad, fp.m, fp.v)], [(n, p.numel()) for n,p in net.named_parameters()]
(a, names), (b, _) = run(True), run(False)
for x, y, nm in zip(a, b, ("params","grads","m","v")):
    d = (x != y).nonzero().flatten()
    print(nm, 'ndiff', d.numel(), 'maxabs', float((x-y).abs().max()), 'first idx', d[:8].tolist())
print(names)
