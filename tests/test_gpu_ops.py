"""Per-op parity of the HIP kernels (through the C-ABI) against torch CPU fp32 / the oracle's
loss formulas.  Tolerances: GEMM-backed ops 1e-5 relative to the output scale (fp32 MFMA is an
exact fmaf chain, only the summation order differs from MKL); elementwise ops 1e-6."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from generative_models_amd import ops  # noqa: E402
from oracle import port  # noqa: E402

DEV = "cuda:0"


def close(got, ref, tol=1e-5, what="", atol=0.0):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    scale = max(ref.abs().max().item(), 1e-30)
    err = max((got - ref).abs().max().item() - atol, 0.0) / scale
    assert err <= tol, "%s: max err %.3e (scaled) > %.1e; ref scale %.3e" % (what, err, tol, scale)


def act_cpu(y, act):
    return F.relu(y) if act == "relu" else torch.sigmoid(y) if act == "sigmoid" else y


SHAPES = [  # (M, K, N)  -- layer shapes of the path + ragged edges
    (256, 784, 400), (512, 784, 400), (256, 400, 784), (256, 20, 400), (256, 400, 1),
    (64, 784, 400), (1024, 784, 400), (336, 784, 400), (256, 400, 40), (256, 40, 400),
    (7, 13, 5), (33, 65, 31), (1, 784, 400), (256, 400, 20), (100, 64, 48),
]


@pytest.mark.parametrize("M,K,N", SHAPES)
@pytest.mark.parametrize("act", ["id", "relu", "sigmoid"])
def test_linear_fwd(M, K, N, act):
    torch.manual_seed(M * 1000 + K + N)
    x, W, b = torch.randn(M, K), torch.randn(N, K) / K ** 0.5, torch.randn(N)
    ref = act_cpu(F.linear(x, W, b), act)
    y = torch.full((M, N), float("nan"), device=DEV)
    ops.linear_fwd(x.to(DEV), W.to(DEV), b.to(DEV), y, act)
    close(y, ref, 2e-6 * max(1, K ** 0.5 / 4), "fwd %s" % act)
    # no bias
    y2 = torch.empty(M, N, device=DEV)
    ops.linear_fwd(x.to(DEV), W.to(DEV), None, y2, act)
    close(y2, act_cpu(F.linear(x, W), act), 2e-6 * max(1, K ** 0.5 / 4), "fwd nobias")


def test_linear_fwd_transpose_detect():
    """A = I check with an asymmetric B (cdna guide section 3)."""
    K = N = 32
    W = torch.arange(N * K, dtype=torch.float32).reshape(N, K)
    x = torch.eye(32)
    y = torch.empty(32, N, device=DEV)
    ops.linear_fwd(x.to(DEV), W.to(DEV), None, y, "id")
    assert torch.equal(y.cpu(), W.t())


@pytest.mark.parametrize("M,K,N", SHAPES)
@pytest.mark.parametrize("epi", ["id", "relu", "sigmoid"])
def test_linear_bwd_dx(M, K, N, epi):
    torch.manual_seed(M + K * 7 + N)
    dA, W = torch.randn(M, N), torch.randn(N, K) / N ** 0.5
    below = torch.rand(M, K) - (0.5 if epi == "relu" else 0.0)
    ref = dA @ W
    if epi == "relu":
        ref = ref * (below > 0)
    elif epi == "sigmoid":
        ref = ref * (below * (1 - below))
    dX = torch.full((M, K), float("nan"), device=DEV)
    ops.linear_bwd_dx(dA.to(DEV), W.to(DEV), dX, below.to(DEV) if epi != "id" else None, epi)
    close(dX, ref, 2e-6 * max(1, N ** 0.5 / 4), "dx %s" % epi)


@pytest.mark.parametrize("M,K,N", SHAPES)
def test_linear_bwd_dw(M, K, N):
    torch.manual_seed(M + K + N * 13)
    dA, X = torch.randn(M, N), torch.randn(M, K)
    dW = torch.full((N, K), float("nan"), device=DEV)
    db = torch.full((N,), float("nan"), device=DEV)
    ops.linear_bwd_dw(dA.to(DEV), X.to(DEV), dW, db)
    tol = 2e-6 * max(1, M ** 0.5 / 4)
    close(dW, dA.t() @ X, tol, "dW")
    close(db, dA.sum(0), tol, "db")
    # accumulate + no db
    base = torch.randn(N, K)
    dW2 = base.to(DEV).clone()
    ops.linear_bwd_dw(dA.to(DEV), X.to(DEV), dW2, None, accumulate=True)
    close(dW2, base + dA.t() @ X, tol, "dW acc")
    db2 = torch.ones(N, device=DEV)
    dW3 = base.to(DEV).clone()
    ops.linear_bwd_dw(dA.to(DEV), X.to(DEV), dW3, db2, accumulate=True)
    close(db2, 1 + dA.sum(0), tol, "db acc")
    close(dW3, base + dA.t() @ X, tol, "dW acc2")


def test_views_and_slots():
    """Row-offset views (ld > K), ring slots driven by a device counter."""
    torch.manual_seed(3)
    B, K, N, R = 48, 20, 33, 5
    ring = torch.randn(R, B, K)
    W, b = torch.randn(N, K), torch.randn(N)
    ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
    rd = ring.to(DEV)
    big = torch.zeros(2 * B, N + 7, device=DEV)
    for it in range(7):
        y = big[B:, :N]          # strided view: ld = N+7, row offset B
        ops.linear_fwd(rd[0], W.to(DEV), b.to(DEV), y, "relu", M=B,
                       x_slot=ops.slot(ctr.data_ptr(), 1, 2, R, B * K))
        close(y, F.relu(F.linear(ring[(it + 2) % R], W, b)), 1e-5, "slot fwd it=%d" % it)
        assert big[:B].abs().max().item() == 0 and big[:, N:].abs().max().item() == 0
        ops.tick(ctr)
    assert ctr.item() == 7


def test_gather():
    torch.manual_seed(0)
    data = torch.randn(1000, 784)
    idx = torch.randint(0, 1000, (3, 256))
    out = torch.empty(256, 784, device=DEV)
    ctr = torch.full((1,), 4, dtype=torch.int64, device=DEV)
    ops.gather_rows(data.to(DEV), idx.to(DEV), out, idx_slot=ops.slot(ctr.data_ptr(), 1, 0, 3, 256))
    assert torch.equal(out.cpu(), data[idx[1]])
    # ragged row length (scalar path) into a wider destination
    data2 = torch.randn(50, 61)
    out2 = torch.zeros(10, 64, device=DEV)
    i2 = torch.randint(0, 50, (10,))
    ops.gather_rows(data2.to(DEV), i2.to(DEV), out2[:, :61])
    assert torch.equal(out2.cpu()[:, :61], data2[i2])


LOSS_CASES = [("ns", "ns"), ("mm", "mm"), ("w", "w"), ("ls", "ls"), ("ra", "ra"),
              ("fisher", "fisher")] + [("f", "f_" + m) for m in port.F_METHODS]


@pytest.mark.parametrize("variant,key", LOSS_CASES)
@pytest.mark.parametrize("B", [256, 100, 1024])
@pytest.mark.parametrize("out_act", ["sigmoid", "relu"])
def test_gan_loss(variant, key, B, out_act):
    """Loss value and d loss / d pre-activation vs autograd through the oracle's loss code."""
    torch.manual_seed(B + len(key))
    ax = torch.randn(B, 1, requires_grad=True)
    ag = torch.randn(B, 1, requires_grad=True)
    f = torch.sigmoid if out_act == "sigmoid" else F.relu
    if out_act == "relu" and variant not in ("w", "ls", "fisher"):
        pytest.skip("log-losses are only used with sigmoid critics")

    class FakeNet:
        def __init__(self, pre):
            self.pre = pre
        def __call__(self, _):
            return f(self.pre)

    # drive the oracle's own loss code with fixed scores
    tr = port.GANPort.__new__(port.GANPort)
    tr.variant, tr.method = variant, (key[2:] if variant == "f" else None)
    tr.LAMBDA = torch.full((1,), 0.3, requires_grad=True)
    tr.RHO = torch.tensor(0.05)

    class M:
        z_dim = 2
    tr.model = M()
    images = torch.zeros(B, 4)
    calls = iter([f(ax), f(ag)])
    tr.model.D = lambda _x: next(calls)
    tr.model.G = lambda z: z
    tr.noise = lambda b: torch.zeros(b, 2)
    if variant in ("ns", "w", "ls"):          # G first, then D(x), D(G)
        pass
    d_loss = tr.d_loss(images)
    d_loss.sum().backward()
    loss_dev = torch.zeros(4, device=DEV)
    dax, dag = torch.empty(B, device=DEV), torch.empty(B, device=DEV)
    aux = torch.tensor([0.3, 0, 0, 0, 0, 0, 0, 0], device=DEV)
    hyper = (0.0, 1.0, 1.0) if variant == "ls" else (0.05,) if variant == "fisher" else ()
    sx, sg = f(ax).detach().reshape(-1).to(DEV), f(ag).detach().reshape(-1).to(DEV)
    ops.gan_loss(key, False, sx, sg, B, out_act, loss_dev, dax, dag, hyper=hyper,
                 loss_slot=ops.slot(0, 0, 2, 0, 1), aux=aux)
    close(loss_dev[2], d_loss.detach().reshape(()), 2e-6, "D loss %s" % key, atol=5e-7)
    close(dax, ax.grad.reshape(-1), 1e-5, "dax %s" % key)
    close(dag, ag.grad.reshape(-1), 1e-5, "dag %s" % key)
    if variant == "fisher":
        lam_new = 0.3 + 0.05 * tr.LAMBDA.grad.item()
        assert abs(aux[0].item() - lam_new) < 1e-6
    # generator mode
    ag2 = torch.randn(B, 1, requires_grad=True)
    tr.model.D = lambda _x: f(ag2)
    g_loss = tr.g_loss(images)
    g_loss.backward()
    dag2 = torch.empty(B, device=DEV)
    ops.gan_loss(key, True, None, f(ag2).detach().reshape(-1).to(DEV), B, out_act, loss_dev, None,
                 dag2, hyper=hyper)
    close(loss_dev[0], g_loss.detach(), 2e-6, "G loss %s" % key, atol=5e-7)
    close(dag2, ag2.grad.reshape(-1), 1e-5, "G dag %s" % key)


@pytest.mark.parametrize("wd,clamp", [(0.0, 0.0), (1e-5, 0.0), (0.0, 0.01)])
def test_adam_vs_torch(wd, clamp):
    torch.manual_seed(9)
    n, steps, lr = 10007, 6, 2e-4
    p = torch.randn(n)
    ref_p = p.clone().requires_grad_()
    opt = torch.optim.Adam([ref_p], lr=lr, weight_decay=wd)
    dp, dm, dv = p.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    sched = torch.from_numpy(ops.adam_schedule(lr, steps)).to(DEV)
    ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
    for s in range(steps):
        g = torch.randn(n) * (10.0 ** (s - 3))
        ref_p.grad = g.clone()
        opt.step()
        if clamp:
            ref_p.data.clamp_(-clamp, clamp)
        ops.adam(dp, g.to(DEV), dm, dv, sched, ops.slot(ctr.data_ptr(), 1, 0, 0, 1),
                 weight_decay=wd, clamp=clamp)
        ops.tick(ctr)
        err = (dp.cpu() - ref_p.data).abs().max().item()
        assert err <= 2e-7 * max(1.0, ref_p.data.abs().max().item()), (s, err)
    st = opt.state[ref_p]
    close(dm, st["exp_avg"], 1e-6, "exp_avg")
    close(dv, st["exp_avg_sq"], 1e-6, "exp_avg_sq")


def test_act_bwd_and_autograd_functions():
    """General autograd path (user-overridden train_D): first and second order vs torch CPU."""
    torch.manual_seed(5)
    B, K, H = 64, 48, 40
    x = torch.randn(B, K)
    W1, b1 = torch.randn(H, K) / 7, torch.randn(H) / 7
    w2, b2 = torch.randn(1, H) / 6, torch.randn(1)
    def net(x, W1, b1, w2, b2, lin):
        return lin(lin(x, W1, b1, "relu"), w2, b2, "sigmoid")
    cpu_lin = lambda x, W, b, a: act_cpu(F.linear(x, W, b), a)
    ps = [t.clone().requires_grad_() for t in (W1, b1, w2, b2)]
    xr = x.clone().requires_grad_()
    out = net(xr, *ps, cpu_lin)
    g = torch.autograd.grad(out.sum(), xr, create_graph=True)[0]
    pen = ((g.norm(2, dim=1) - 1) ** 2).mean() + out.mean()
    pen.backward()
    dps = [t.clone().to(DEV).requires_grad_() for t in (W1, b1, w2, b2)]
    xd = x.clone().to(DEV).requires_grad_()
    out_d = net(xd, *dps, ops.fused_linear)
    close(out_d, out, 1e-5, "fused fwd")
    g_d = torch.autograd.grad(out_d.sum(), xd, create_graph=True)[0]
    close(g_d, g, 1e-5, "input grad")
    pen_d = ((g_d.norm(2, dim=1) - 1) ** 2).mean() + out_d.mean()
    pen_d.backward()
    for a, b_, n in zip(dps, ps, ("W1", "b1", "w2", "b2")):
        close(a.grad, b_.grad, 2e-5, "double-backward grad " + n)


def test_graph_replay_with_tick():
    torch.manual_seed(1)
    B, K, N, R = 64, 20, 48, 4
    ring = torch.randn(R, B, K)
    W, b = torch.randn(N, K), torch.randn(N)
    rd, Wd, bd = ring.to(DEV), W.to(DEV), b.to(DEV)
    y = torch.zeros(B, N, device=DEV)
    acc = torch.zeros(N, K, device=DEV)
    ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
    torch.cuda.synchronize()

    def body(s):
        ops.linear_fwd(rd[0], Wd, bd, y, "sigmoid", M=B,
                       x_slot=ops.slot(ctr.data_ptr(), 1, 0, R, B * K), stream=s)
        ops.linear_bwd_dw(y, rd[0], acc, None, M=B, accumulate=True,
                          x_slot=ops.slot(ctr.data_ptr(), 1, 0, R, B * K), stream=s)
        ops.tick(ctr, 1, stream=s)

    g = ops.Graph().capture(body)
    ref = torch.zeros(N, K)
    for it in range(6):
        g.launch()
        yy = torch.sigmoid(F.linear(ring[it % R], W, b))
        ref += yy.t() @ ring[it % R]
    torch.cuda.synchronize()
    assert ctr.item() == 6
    close(y, yy, 1e-5, "graph y")
    close(acc, ref, 1e-5, "graph acc")


def test_began_update_matches_python_controller_and_plateau_scheduler():
    """gm_began_update vs be_gan.py:189-195 executed with torch's own ReduceLROnPlateau (small
    patience so that learning-rate halvings actually happen)."""
    from torch.optim.lr_scheduler import ReduceLROnPlateau
    from generative_models_amd import ops_fused as of
    torch.manual_seed(0)
    gamma, lam, patience, lrD, lrG = 0.5, 1e-3, 3, 1e-4, 2e-4
    pD, pG = torch.zeros(1, requires_grad=True), torch.zeros(1, requires_grad=True)
    oD, oG = torch.optim.Adam([pD], lr=lrD), torch.optim.Adam([pG], lr=lrG)
    sD = ReduceLROnPlateau(oD, factor=0.50, threshold=0.01, patience=patience)
    sG = ReduceLROnPlateau(oG, factor=0.50, threshold=0.01, patience=patience)
    st = torch.zeros(8, device=DEV)
    st[4] = 1.0; st[5] = 1.0
    dst = torch.zeros(8, dtype=torch.float64, device=DEV)
    dst[0] = float("inf"); dst[1] = lrD; dst[2] = lrG; dst[3] = lrD; dst[4] = lrG
    ist = torch.zeros(2, dtype=torch.int64, device=DEV)
    ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
    K = 0.0
    for step in range(40):
        DX = torch.tensor(30.0 - 0.4 * min(step, 8) + 0.01 * (step % 3))      # plateaus after 8 steps
        DG = torch.tensor(12.0 + 0.3 * step)
        convergence = (DX + torch.abs(gamma * DX - DG)).item()
        K_update = (K + lam * (gamma * DX - DG)).item()
        K = min(max(0, K_update), 1)
        sD.step(convergence); sG.step(convergence)
        st[1] = DX.item(); st[2] = DG.item()
        of.began_update(st, dst, ist, gamma, lam, patience, ctr)
        got = st.cpu()
        assert abs(got[0].item() - K) <= 1e-7, (step, got[0].item(), K)
        assert abs(got[3].item() - convergence) <= 1e-5
        assert abs(got[4].item() * lrD - oD.param_groups[0]["lr"]) <= 1e-12, step
        assert abs(got[5].item() * lrG - oG.param_groups[0]["lr"]) <= 1e-12, step
    assert oD.param_groups[0]["lr"] < lrD, "the test must exercise at least one halving"
    assert ctr.item() == 40


def test_info_q_loss_vs_torch():
    from generative_models_amd import ops_fused as of
    torch.manual_seed(2)
    B, zd, nd, nc = 48, 8, 10, 10
    q = torch.randn(B, nd + nc, requires_grad=True)
    noise = torch.zeros(B, zd + nd + nc)
    noise[:, :zd] = torch.randn(B, zd)
    cat = torch.randint(0, nd, (B,))
    noise[torch.arange(B), zd + cat] = 1
    noise[:, zd + nd:] = torch.randn(B, nc)
    loss = F.cross_entropy(q[:, :nd], torch.max(noise[:, zd:zd + nd], 1)[1]) + \
        F.mse_loss(q[:, nd:], noise[:, zd + nd:])
    loss.backward()
    dq = torch.empty(B, nd + nc, device=DEV)
    out = torch.zeros(2, device=DEV)
    of.info_q_loss(q.detach().to(DEV), noise.to(DEV), ops.NO_SLOT, B, zd, nd, nc, dq, out,
                   ops.slot(0, 0, 1, 0, 1))
    close(out[1], loss.detach(), 2e-6, "MI loss", atol=1e-6)
    close(dq, q.grad, 1e-5, "d MI / dq")


def test_l1_rows_and_dx_add():
    from generative_models_amd import ops_fused as of
    torch.manual_seed(4)
    B, I, H = 40, 64, 48
    Y, X = torch.randn(2 * B, I), torch.randn(2 * B, I)
    Kv = torch.tensor([0.37], device=DEV)
    dY, rows = torch.empty(2 * B, I, device=DEV), torch.empty(2 * B, device=DEV)
    of.l1_rows(Y.to(DEV), X.to(DEV), 2 * B, B, Kv, dY, rows)
    close(rows, (Y - X).abs().sum(1), 1e-6, "row sums")
    coef = torch.cat([torch.full((B, 1), 1.0 / B), torch.full((B, 1), -0.37 / B)])
    close(dY, torch.sign(Y - X) * coef, 1e-6, "L1 grad")
    dA, W = torch.randn(B, H), torch.randn(H, I) / 7
    below, add = torch.rand(B, I), torch.randn(B, I)
    out = torch.empty(B, I, device=DEV)
    ops.linear_bwd_dx(dA.to(DEV), W.to(DEV), out, below=below.to(DEV), epi="sigmoid",
                      add=add.to(DEV), add_scale=-1.0)
    close(out, (dA @ W - add) * below * (1 - below), 1e-5, "dx with addend")
