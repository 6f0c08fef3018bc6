import torch, sys
sys.path.insert(0,'/root/repo')
from generative_models_amd import ops, ops_fused as of
M,K,N,rows=2048,400,784,1024
torch.manual_seed(M + N)
dev = "cuda"
x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev) * 0.1
real = torch.bernoulli(torch.full((rows, N), 0.3, device=dev)); eps = torch.rand(rows, device=dev)
y0, y1 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
xh0, xh1 = torch.zeros(rows, N, device=dev), torch.full((rows, N), -3.0, device=dev)
ops.linear_fwd(x, W, b, y0, "sigmoid"); of.interp(eps, ops.NO_SLOT, real, y0[:rows], xh0)
ops.linear_fwd_interp(x, W, b, y1, "sigmoid", eps, ops.NO_SLOT, real, xh1, rows)
torch.cuda.synchronize()
d = (xh0 != xh1)
print('ndiff', int(d.sum()), 'of', d.numel(), 'max', float((xh0-xh1).abs().max()))
idx = d.nonzero()[:10]; print(idx.tolist())
print('rows with diff', d.any(1).sum().item(), 'cols with diff', d.any(0).sum().item())
print('untouched (-3)', int((xh1 == -3).sum()))
