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


def close64(got, ref64, ref32, what=""):
    """The HIP result against an fp64 evaluation of the same expression, bounded by TWICE the error the fp32 CPU
    result (MKL / ATen, its own summation order) has against that fp64 value, plus one fp32 ulp of the output's scale
    (exact cases: M = 1, identity operands).  An MKL-vs-HIP comparison with a sqrt(K) allowance cannot tell which of
    the two is off; this can."""
    got, ref32, ref64 = got.detach().cpu().double(), ref32.detach().cpu().double(), ref64.detach().cpu().double()
    scale = max(ref64.abs().max().item(), 1e-30)
    e_hip = (got - ref64).abs().max().item() / scale
    e_cpu = (ref32 - ref64).abs().max().item() / scale
    bound = 2.0 * e_cpu + 2.0 ** -23
    assert e_hip <= bound, "%s: HIP err %.3e (scaled) > 2 x CPU fp32 err %.3e + 1 ulp; scale %.3e" % (what, e_hip, e_cpu, scale)


def act_cpu(y, act):
    return F.relu(y) if act == "relu" else torch.sigmoid(y) if act == "sigmoid" else y


SHAPES = [  # (M, K, N)  -- layer shapes of the path + ragged edges
    (256, 784, 400), (512, 784, 400), (256, 400, 784), (256, 20, 400), (256, 400, 1),
    (64, 784, 400), (1024, 784, 400), (336, 784, 400), (256, 400, 40), (256, 40, 400),
    (7, 13, 5), (33, 65, 31), (1, 784, 400), (256, 400, 20), (100, 64, 48),
    # > 512 tiles of 32x32: several rounds of workgroups (batch >= 1024, wide layers)
    (2048, 784, 400), (1100, 130, 500), (1025, 77, 519), (64, 1000, 900), (1024, 400, 784),
]


@pytest.mark.parametrize("M,K,N", SHAPES)
@pytest.mark.parametrize("act", ["id", "relu", "sigmoid"])
def test_linear_fwd(M, K, N, act):
    torch.manual_seed(M * 1000 + K + N)
    x, W, b = torch.randn(M, K), torch.randn(N, K) / K ** 0.5, torch.randn(N)
    ref = act_cpu(F.linear(x, W, b), act)
    ref64 = act_cpu(F.linear(x.double(), W.double(), b.double()), act)
    y = torch.full((M, N), float("nan"), device=DEV)
    ops.linear_fwd(x.to(DEV), W.to(DEV), b.to(DEV), y, act)
    close64(y, ref64, ref, "fwd %s" % act)
    # no bias
    y2 = torch.empty(M, N, device=DEV)
    ops.linear_fwd(x.to(DEV), W.to(DEV), None, y2, act)
    close64(y2, act_cpu(F.linear(x.double(), W.double()), act), act_cpu(F.linear(x, W), act), "fwd nobias")


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
    def expr(dA_, W_, below_):
        r = dA_ @ W_
        if epi == "relu":
            r = r * (below_ > 0)
        elif epi == "sigmoid":
            r = r * (below_ * (1 - below_))
        return r
    dX = torch.full((M, K), float("nan"), device=DEV)
    ops.linear_bwd_dx(dA.to(DEV), W.to(DEV), dX, below.to(DEV) if epi != "id" else None, epi)
    close64(dX, expr(dA.double(), W.double(), below.double()), expr(dA, W, below), "dx %s" % epi)


@pytest.mark.parametrize("M,K,N", SHAPES)
def test_linear_bwd_dw(M, K, N):
    torch.manual_seed(M + K + N * 13)
    dA, X = torch.randn(M, N), torch.randn(M, K)
    dW = torch.full((N, K), float("nan"), device=DEV)
    db = torch.full((N,), float("nan"), device=DEV)
    ops.linear_bwd_dw(dA.to(DEV), X.to(DEV), dW, db)
    tol = 2e-6 * max(1, M ** 0.5 / 4)
    close64(dW, dA.double().t() @ X.double(), dA.t() @ X, "dW")
    # (db: ATen's column sum is pairwise-blocked and more exact than any sequential fp32 sum: keep the sqrt(M) bound)
    close(db, dA.double().sum(0), tol, "db")
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


@pytest.mark.parametrize("B,zd,nd,nc", [(48, 8, 10, 10), (256, 20, 10, 10), (300, 8, 7, 5)])
def test_info_q_loss_vs_torch(B, zd, nd, nc):
    """(10, 10): the register-resident instantiation for info_gan.py's code widths; other widths: the generic loops."""
    from generative_models_amd import ops_fused as of
    torch.manual_seed(2)
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


@pytest.mark.parametrize("variant,out_act", [("ns", "sigmoid"), ("w", "id"), ("ls", "id")])
@pytest.mark.parametrize("gen_mode", [False, True])
@pytest.mark.parametrize("B,Hd", [(256, 400), (24, 37)])
def test_fused_head_pair(variant, out_act, gen_mode, B, Hd):
    """head_fwd_loss (+ dH written while the row is hot) and head_bwd (gw2, gb2, loss) against autograd
    on the same tail: s = act(h.w2 + b2) -> loss (oracle formulas, ns_gan.py:191-214)."""
    from generative_models_amd import ops_fused as of
    torch.manual_seed(B + Hd)
    R = B if gen_mode else 2 * B
    H = torch.relu(torch.randn(R, Hd)).requires_grad_(True)
    w2 = (torch.randn(1, Hd) / Hd ** 0.5).requires_grad_(True)
    b2 = torch.randn(1).requires_grad_(True)
    hyper = [0.0, 1.0, 1.0]
    s = act_cpu(H @ w2.t() + b2, out_act)[:, 0]
    eps = 1e-8
    if variant == "ns":
        loss = -torch.mean(torch.log(s + eps)) if gen_mode else \
            -torch.mean(torch.log(s[:B] + eps)) - torch.mean(torch.log(1 - s[B:] + eps))
    elif variant == "w":
        loss = -torch.mean(s) if gen_mode else -(torch.mean(s[:B]) - torch.mean(s[B:]))
    else:
        loss = 0.5 * torch.mean((s - 1.0) ** 2) if gen_mode else \
            0.5 * torch.mean((s[:B] - 1.0) ** 2) + 0.5 * torch.mean((s[B:] - 0.0) ** 2)
    loss.backward()
    dH_ref = H.grad * (H.detach() > 0)      # ReluBackward of the hidden layer folded in
    Hd_, w2d, b2d = H.detach().to(DEV), w2.detach().to(DEV).clone(), b2.detach().to(DEV).clone()
    S = torch.empty(R, device=DEV); dS = torch.empty(R, device=DEV); rl = torch.empty(R, device=DEV)
    dH = torch.full((R, Hd), 7.0, device=DEV)
    gw2 = torch.empty(1, Hd, device=DEV); gb2 = torch.empty(1, device=DEV)
    out = torch.zeros(1, device=DEV)
    of.head_fwd_loss(variant, gen_mode, Hd_, w2d, b2d, out_act, B, hyper, 1.0 / B, None, S, dS, rl,
                     dH=dH)
    of.head_bwd(Hd_, dS, w2d, rl, None, None if gen_mode else gw2, None if gen_mode else gb2, out,
                ops.NO_SLOT, 1.0 / B, gen_mode, B)
    close(S, s, 1e-5, "scores")
    close(out[0], loss, 1e-5, "loss", atol=1e-6)
    close(dH, dH_ref, 1e-5, "dH")
    if not gen_mode:
        close(gw2, w2.grad, 1e-5, "gw2")
        close(gb2, b2.grad, 1e-5, "gb2", atol=1e-9)
    # the unfused split (dH from head_bwd) must agree bit for bit
    dH2 = torch.empty_like(dH)
    of.head_fwd_loss(variant, gen_mode, Hd_, w2d, b2d, out_act, B, hyper, 1.0 / B, None, S, dS, rl)
    of.head_bwd(Hd_, dS, w2d, rl, dH2, None, None, out, ops.NO_SLOT, 1.0 / B, gen_mode, B)
    assert torch.equal(dH, dH2)


@pytest.mark.parametrize("B,Hd", [(256, 400), (1024, 400), (24, 37), (3, 5)])
def test_head_fwd_loss_final(B, Hd):
    """The last-workgroup finalisation (loss scalar + tick, no head_bwd launch) equals head_bwd's
    scalar path, re-arms its counter, and is repeatable launch after launch."""
    from generative_models_amd import ops_fused as of
    torch.manual_seed(B)
    H = torch.relu(torch.randn(B, Hd)).to(DEV)
    w2 = (torch.randn(1, Hd) / Hd ** 0.5).to(DEV); b2 = torch.randn(1).to(DEV)
    S = torch.empty(B, device=DEV); dS = torch.empty(B, device=DEV); rl = torch.empty(B, device=DEV)
    dH = torch.empty(B, Hd, device=DEV)
    ref = torch.zeros(1, device=DEV)
    of.head_fwd_loss("ns", True, H, w2, b2, "sigmoid", B, [], 1.0 / B, None, S, dS, rl, dH=dH)
    of.head_bwd(H, dS, w2, rl, None, None, None, ref, ops.NO_SLOT, 1.0 / B, True, B)
    dH_ref = dH.clone()
    out = torch.zeros(4, device=DEV)
    ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
    done = torch.zeros(1, dtype=torch.int32, device=DEV)
    for k in range(4):                      # slot k of the loss ring, counter-addressed
        dH.zero_()
        of.head_fwd_loss("ns", True, H, w2, b2, "sigmoid", B, [], 1.0 / B, None, S, dS, rl, dH=dH,
                         final=dict(loss_out=out, loss_slot=ops.slot(ctr.data_ptr(), 1, 0, 4, 1),
                                    done=done, tick=ctr))
    torch.cuda.synchronize()
    assert ctr.item() == 4 and done.item() == 0
    assert torch.equal(dH, dH_ref)
    close(out, ref.expand(4), 1e-6, "final loss")
    s_cpu = torch.sigmoid(H.cpu() @ w2.cpu().t() + b2.cpu())[:, 0]
    close(out[0], -torch.mean(torch.log(s_cpu + 1e-8)), 1e-5, "loss vs torch")


@pytest.mark.parametrize("B,I,Hd", [(256, 784, 400), (24, 36, 20), (512, 784, 400)])
def test_dw_adam_with_head_in_one_launch(B, I, Hd):
    """gm_linear_bwd_dw_adam_head == gm_head_bwd_fused followed by gm_linear_bwd_dw_adam, bit for
    bit (the head workgroups only ride in the GEMM's grid; the arithmetic is the same code)."""
    import torch.nn as nn
    from generative_models_amd import ops_fused as of
    from generative_models_amd.engine import FlatParams, _Linear

    def run(grouped):
        torch.manual_seed(7)
        net = nn.Sequential(nn.Linear(I, Hd), nn.Linear(Hd, 1))
        fp = FlatParams(net.parameters(), DEV)
        fp.m.normal_().mul_(1e-3); fp.v.uniform_(0.0, 1e-4)
        L1, L2 = _Linear(fp, net[0]), _Linear(fp, net[1])
        X2 = torch.rand(2 * B, I).to(DEV)
        H = torch.relu(torch.randn(2 * B, Hd)).to(DEV)
        S = torch.empty(2 * B, device=DEV); dS = torch.empty_like(S); rl = torch.empty_like(S)
        dH = torch.empty(2 * B, Hd, device=DEV)
        loss = torch.zeros(1, device=DEV)
        sched = torch.from_numpy(ops.adam_schedule(2e-4, 4)).to(DEV)
        adam = dict(sched=sched, sched_slot=ops.slot(0, 0, 2, 0, 1), clamp=0.0)
        of.head_fwd_loss("ns", False, H, L2.W, L2.b, "sigmoid", B, [], 1.0 / B, None, S, dS, rl, dH=dH)
        if grouped:
            ops.linear_bwd_dw_adam_head(dH, X2, L1, adam,
                                        dict(H=H, dS=dS, lin=L2, rowloss=rl, loss_out=loss,
                                             loss_slot=ops.NO_SLOT, inv_b=1.0 / B, B=B, adam=adam),
                                        M=2 * B)
        else:
            of.head_bwd(H, dS, L2.W, rl, None, L2.gW, L2.gb, loss, ops.NO_SLOT, 1.0 / B, False, B,
                        lin=L2, adam=adam)
            ops.linear_bwd_dw_adam(dH, X2, L1, adam, M=2 * B)
        torch.cuda.synchronize()
        return [t.clone() for t in (fp.flat, fp.grad, fp.m, fp.v, loss)]

    a, b = run(True), run(False)
    for x, y, name in zip(a, b, ("params", "grads", "exp_avg", "exp_avg_sq", "loss")):
        assert torch.equal(x, y), name
    assert a[1].abs().sum().item() > 0 and torch.isfinite(a[0]).all()


@pytest.mark.parametrize("M,K,N", [(512, 20, 400), (256, 20, 400), (2048, 20, 400), (37, 20, 50), (100, 16, 400),
                                   (336, 32, 404), (64, 4, 32), (1, 20, 400), (512, 28, 784)])
@pytest.mark.parametrize("act", ["id", "relu", "sigmoid"])
def test_short_reduction_forward_equals_the_16_wave_kernel_bit_for_bit(M, K, N, act):
    """K <= 32 on 16-byte aligned operands runs gemm16_k32_fwd_kernel (one wave per 16 x 32 piece, no cross-wave
    reduction); the same rows at a row stride that is not a multiple of 4 floats take the 16-wave split-reduction
    kernel.  Same fragments, same MFMA chains per chunk, partial tiles added in wave order onto 0: the same bits
    (also through a ring slot on the operand, and with the x_hat rider of WGAN-GP)."""
    torch.manual_seed(M + 31 * K + N)
    x, W, b = torch.randn(M, K), torch.randn(N, K) * 0.3, torch.randn(N) * 0.1
    xa = x.to(DEV)
    wide = torch.zeros(M, K + 1, device=DEV)
    wide[:, :K] = xa
    xu = wide[:, :K]                                             # ld = K + 1: not the 16-byte path
    Wd, bd = W.to(DEV), b.to(DEV)
    for bias in (bd, None):
        y1, y2 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
        ops.linear_fwd(xa, Wd, bias, y1, act)
        ops.linear_fwd(xu, Wd, bias, y2, act)
        assert torch.equal(y1, y2)
        ref = act_cpu(F.linear(x.double(), W.double(), b.double() if bias is not None else None), act)
        assert (y1.cpu().double() - ref).abs().max().item() < 1e-4
    # ring slot on the operand (the generator's noise ring): slot 2 of 3
    ring = torch.randn(3, M, K, device=DEV)
    ctr = torch.full((1,), 2, dtype=torch.int64, device=DEV)
    slot = ops.slot(ctr.data_ptr(), 1, 0, 3, M * K)
    y3, y4 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    ops.linear_fwd(ring.view(-1, K), Wd, bd, y3, act, M=M, x_slot=slot)
    ops.linear_fwd(ring[2].contiguous(), Wd, bd, y4, act)
    assert torch.equal(y3, y4)


@pytest.mark.parametrize("M,K,N,B,I", [(512, 20, 400, 256, 784), (256, 20, 400, 256, 784),
                                       (33, 13, 31, 7, 10), (64, 20, 400, 100, 36)])
def test_linear_fwd_with_gather_riding(M, K, N, B, I):
    """gm_linear_fwd_gather == gm_linear_fwd + gm_gather_rows, bit for bit (aligned shapes ride
    in the GEMM's grid, the ragged case takes the two-launch fallback inside the library)."""
    torch.manual_seed(M + B)
    x, W, b = torch.randn(M, K).to(DEV), (torch.randn(N, K) / K ** 0.5).to(DEV), torch.randn(N).to(DEV)
    data = torch.rand(1000, I).to(DEV)
    ring = torch.randint(0, 1000, (3, B)).to(DEV)
    ctr = torch.full((1,), 2, dtype=torch.int64, device=DEV)
    slot = ops.slot(ctr.data_ptr(), 1, 0, 3, B)
    y1, y2 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    o1, o2 = torch.zeros(B, I, device=DEV), torch.zeros(B, I, device=DEV)
    ops.linear_fwd_gather(x, W, b, y1, "relu", data, ring.view(-1), o1, idx_slot=slot)
    ops.linear_fwd(x, W, b, y2, "relu")
    ops.gather_rows(data, ring.view(-1), o2, idx_slot=slot)
    assert torch.equal(y1, y2) and torch.equal(o1, o2)
    assert torch.equal(o1.cpu(), data.cpu()[ring[2].cpu()])


@pytest.mark.parametrize("B,H,I,Z", [(256, 400, 784, 20), (1024, 400, 784, 20), (24, 20, 36, 8),
                                     (32, 31, 33, 5)])
def test_dw_adam_pair_in_one_launch(B, H, I, Z):
    """gm_linear_bwd_dw_adam_pair == two gm_linear_bwd_dw_adam launches, bit for bit below 1024 rows (incl. the
    unaligned two-launch fallback), to rounding at B = 1024 (wide tiles, interleaved fragments)."""
    import torch.nn as nn
    from generative_models_amd.engine import FlatParams, _Linear

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
            ops.linear_bwd_dw_adam_pair(dict(dA=dX, X=Hg, lin=L2, adam=adam),
                                        dict(dA=dH, X=z, lin=L1, adam=adam))
        else:
            ops.linear_bwd_dw_adam(dX, Hg, L2, adam)
            ops.linear_bwd_dw_adam(dH, z, L1, adam)
        torch.cuda.synchronize()
        return [t.clone() for t in (fp.flat, fp.grad, fp.m, fp.v)]

    a, b = run(True), run(False)
    for x, y, name in zip(a, b, ("params", "grads", "exp_avg", "exp_avg_sq")):
        if B >= 1024:
            # reductions of >= 1024 rows: the pair runs BOTH gradients on the interleaved-fragment kernel (a
            # different order of the four k inside an MFMA step), a lone 400 x 21 gradient keeps the 16-byte form
            assert float((x - y).abs().max()) <= 2e-6 * max(1.0, float(y.abs().max())), name
        else:
            assert torch.equal(x, y), name
    assert a[1].abs().sum().item() > 0 and torch.isfinite(a[0]).all()


@pytest.mark.parametrize("B,H,I,Z", [(512, 400, 784, 20), (336, 400, 784, 20), (1024, 400, 784, 20), (24, 20, 36, 8),
                                     (32, 31, 33, 5)])
def test_dw_adam_pair_closing_a_vae_batch(B, H, I, Z):
    """gm_linear_bwd_dw_adam_pair_finalize == gm_linear_bwd_dw_adam_pair followed by gm_sum_finalize2_tick, bit for
    bit (incl. the unaligned fallback to separate launches): parameters, gradients, both Adam moments, both sums at
    the slot the counter named BEFORE the launch, counter advanced exactly once per launch, arrival counter re-armed.
    The Adam schedule is read through the same counter the launch advances -- a tick ahead of a late workgroup's
    epilogue would give that workgroup the next step's step size."""
    import torch.nn as nn
    from generative_models_amd import ops_fused as of
    from generative_models_amd.engine import FlatParams, _Linear
    steps = 3

    def run(fused):
        torch.manual_seed(11)
        net = nn.Sequential(nn.Linear(H, 2 * Z), nn.Linear(I, H))       # the encoder's two layers (vae.py:80-98)
        fp = FlatParams(net.parameters(), DEV)
        fp.m.normal_().mul_(1e-3); fp.v.uniform_(0.0, 1e-4)
        ML, E1 = _Linear(fp, net[0]), _Linear(fp, net[1])
        X, dHe = torch.rand(B, I).to(DEV), torch.randn(B, H).to(DEV)
        He, dml = torch.relu(torch.randn(B, H)).to(DEV), torch.randn(B, 2 * Z).to(DEV)
        ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
        sched = torch.from_numpy(ops.adam_schedule(2e-4, steps + 1)).to(DEV)
        adam = dict(sched=sched, sched_slot=ops.slot(ctr.data_ptr(), 1, 0, 0, 1), clamp=0.0)   # as VAEEngine._issue
        pa, pb = (torch.rand(B * 28) * 30).to(DEV), torch.rand((B * Z + 255) // 256).to(DEV)
        oa, ob = torch.zeros(steps + 1, device=DEV), torch.zeros(steps + 1, device=DEV)
        slot = ops.slot(ctr.data_ptr(), 1, 0, 0, 1)
        done = torch.zeros(1, dtype=torch.int32, device=DEV)
        a1, a2 = dict(dA=dHe, X=X, lin=E1, adam=adam), dict(dA=dml, X=He, lin=ML, adam=adam)
        for _ in range(steps):
            if fused:
                ops.linear_bwd_dw_adam_pair_finalize(a1, a2, dict(pa=pa, na=pa.numel(), out_a=oa, slot_a=slot, pb=pb,
                                                                  nb=pb.numel(), out_b=ob, slot_b=slot, done=done,
                                                                  tick=ctr))
            else:
                ops.linear_bwd_dw_adam_pair(a1, a2)
                of.sum_finalize2(pa, pa.numel(), oa, slot, pb, pb.numel(), ob, slot, tick=ctr)
            pa.mul_(0.5)                                     # the next batch's partials differ
        torch.cuda.synchronize()
        assert int(ctr) == steps and int(done) == 0
        return [t.clone() for t in (fp.flat, fp.grad, fp.m, fp.v, oa, ob)]

    a, b = run(True), run(False)
    for x, y, name in zip(a, b, ("params", "grads", "exp_avg", "exp_avg_sq", "sum a", "sum b")):
        assert torch.equal(x, y), name
    assert a[4][:steps].min().item() > 0 and a[4][steps].item() == 0.0 and torch.isfinite(a[0]).all()


@pytest.mark.parametrize("B,I,Hd", [(256, 784, 400), (24, 36, 20), (33, 30, 17)])
def test_dx_with_scalar_head_riding(B, I, Hd):
    """gm_linear_bwd_dx_head == gm_head_bwd (generator mode: loss scalar + tick) followed by
    gm_linear_bwd_dx, bit for bit; the counter advances exactly once per launch."""
    from types import SimpleNamespace
    from generative_models_amd import ops_fused as of
    torch.manual_seed(B)
    H = torch.relu(torch.randn(B, Hd)).to(DEV)
    W1 = (torch.randn(Hd, I) / I ** 0.5).to(DEV)
    Xg = torch.rand(B, I).to(DEV)
    L2 = SimpleNamespace(W=(torch.randn(1, Hd) / Hd ** 0.5).to(DEV), b=torch.randn(1).to(DEV))
    S = torch.empty(B, device=DEV); dS = torch.empty(B, device=DEV); rl = torch.empty(B, device=DEV)
    dH = torch.empty(B, Hd, device=DEV)
    of.head_fwd_loss("ns", True, H, L2.W, L2.b, "sigmoid", B, [], 1.0 / B, None, S, dS, rl, dH=dH)
    ref_loss = torch.zeros(3, device=DEV); dX_ref = torch.empty(B, I, device=DEV)
    of.head_bwd(H, dS, L2.W, rl, None, None, None, ref_loss, ops.slot(0, 0, 1, 0, 1), 1.0 / B, True, B)
    ops.linear_bwd_dx(dH, W1, dX_ref, below=Xg, epi="sigmoid")
    ctr = torch.ones(1, dtype=torch.int64, device=DEV)
    loss = torch.zeros(3, device=DEV); dX = torch.empty(B, I, device=DEV)
    ops.linear_bwd_dx_head(dH, W1, dX,
                           dict(H=H, dS=dS, lin=L2, rowloss=rl, loss_out=loss,
                                loss_slot=ops.slot(ctr.data_ptr(), 1, 0, 3, 1), inv_b=1.0 / B, B=B,
                                gen_mode=True, tick=ctr), below=Xg, epi="sigmoid")
    torch.cuda.synchronize()
    assert ctr.item() == 2
    assert torch.equal(dX, dX_ref) and torch.equal(loss, ref_loss) and loss[1].item() != 0.0


# ---------------------------------------------------------------------------------------------
# Folded critic head (round 3): the N = 1 layer has no launch of its own (include/gm_hip.h,
# gm_head_fold_args): partial dots in the hidden layer's forward epilogue, scores / losses / dS rebuilt
# in the consumers' prologues, dH formed in registers.  ns_gan.py:57-60,191-192,214.
# ---------------------------------------------------------------------------------------------
def _oracle_tail_loss(variant, s, B, gen_mode):
    """The reference's loss lines on scores s (ns_gan.py:191-192,214; w_gan.py; ls_gan.py:192-193,213)."""
    eps = 1e-8
    if variant == "ns":
        return -torch.mean(torch.log(s + eps)) if gen_mode else \
            -torch.mean(torch.log(s[:B] + eps)) - torch.mean(torch.log(1 - s[B:] + eps))
    if variant == "w":
        return -torch.mean(s) if gen_mode else -(torch.mean(s[:B]) - torch.mean(s[B:]))
    return 0.5 * torch.mean((s - 1.0) ** 2) if gen_mode else \
        0.5 * torch.mean((s[:B] - 1.0) ** 2) + 0.5 * torch.mean((s[B:] - 0.0) ** 2)


@pytest.mark.parametrize("M,I,Hd", [(512, 784, 400), (256, 784, 400), (48, 36, 20), (33, 30, 17), (2048, 784, 400)])
def test_folded_head_forward_leaves_partial_dots_and_snapshot(M, I, Hd):
    """gm_linear_fwd_headpart: the hidden layer is bit-identical to gm_linear_fwd's, the per-tile
    partial dots add up to h . w2 (fp64 reference), the snapshot holds (w2, b2), and a second launch
    reproduces every bit."""
    from types import SimpleNamespace
    torch.manual_seed(M + Hd)
    X = torch.rand(M, I).to(DEV)
    W1 = (torch.randn(Hd, I) / I ** 0.5).to(DEV); b1 = (torch.randn(Hd) * 0.1).to(DEV)
    L2 = SimpleNamespace(W=(torch.randn(1, Hd) / Hd ** 0.5).to(DEV), b=torch.randn(1).to(DEV))
    fold = ops.HeadFold(M, Hd, DEV)
    Y0, Y1 = torch.empty(M, Hd, device=DEV), torch.full((M, Hd), -5.0, device=DEV)
    ops.linear_fwd(X, W1, b1, Y0, "relu")
    ops.linear_fwd_headpart(X, W1, b1, Y1, "relu", L2, fold)
    torch.cuda.synchronize()
    if M < 1024:
        assert torch.equal(Y0, Y1)
    else:                                           # gm_linear_fwd takes the LDS macro-tile kernel there: other summation order
        close(Y1, Y0, 2e-6, "hidden layer")
    assert torch.equal(fold.snap[:Hd], L2.W.view(-1)) and fold.snap[Hd].item() == L2.b.item()
    ref = (Y1.double().cpu() @ L2.W.double().cpu().t())[:, 0]
    assert bool((fold.part[:, fold.nparts:] == 0).all())
    got = fold.part.double().cpu().sum(1)
    scale = (Y1.abs().double().cpu() @ L2.W.abs().double().cpu().t())[:, 0].max().item()
    assert (got - ref).abs().max().item() <= 2e-6 * scale
    p1 = fold.part.clone()
    fold.part[:, :fold.nparts].fill_(9.0)
    ops.linear_fwd_headpart(X, W1, b1, Y1, "relu", L2, fold)
    torch.cuda.synchronize()
    assert torch.equal(p1, fold.part)


@pytest.mark.parametrize("variant,out_act", [("ns", "sigmoid"), ("w", "sigmoid"), ("ls", "sigmoid"), ("w", "id")])
@pytest.mark.parametrize("B,I,Hd", [(256, 784, 400), (24, 36, 20), (512, 784, 400), (100, 64, 48)])
def test_folded_head_critic_step(variant, out_act, B, I, Hd):
    """Critic step tail, folded (2 launches: forward + partial dots; layer-1 weight gradient + Adam with
    the head backward riding, dH formed in registers) against (a) autograd on the CPU through the
    oracle's loss lines and (b) the unfolded 3-launch sequence: gradients, loss, Adam'd parameters."""
    import torch.nn as nn
    from generative_models_amd import ops_fused as of
    from generative_models_amd.engine import FlatParams, _Linear
    hyper = [0.0, 1.0, 1.0]

    def run(folded):
        torch.manual_seed(11)
        net = nn.Sequential(nn.Linear(I, Hd), nn.Linear(Hd, 1))
        ref = [p.detach().clone().requires_grad_(True) for p in net.parameters()]
        fp = FlatParams(net.parameters(), DEV)
        fp.m.normal_().mul_(1e-3); fp.v.uniform_(0.0, 1e-4)
        L1, L2 = _Linear(fp, net[0]), _Linear(fp, net[1])
        X2c = torch.bernoulli(torch.full((2 * B, I), 0.3))
        X2 = X2c.to(DEV)
        H = torch.empty(2 * B, Hd, device=DEV)
        S = torch.zeros(2 * B, device=DEV); dS = torch.zeros_like(S); rl = torch.zeros_like(S)
        loss = torch.zeros(1, device=DEV)
        sched = torch.from_numpy(ops.adam_schedule(2e-4, 4)).to(DEV)
        adam = dict(sched=sched, sched_slot=ops.slot(0, 0, 2, 0, 1), clamp=0.0)
        head = dict(H=H, lin=L2, loss_out=loss, loss_slot=ops.NO_SLOT, inv_b=1.0 / B, B=B, adam=adam)
        if folded:
            fold = ops.HeadFold(2 * B, Hd, DEV)
            ops.linear_fwd_headpart(X2, L1.W, L1.b, H, "relu", L2, fold)
            fa = fold.args(variant, out_act, hyper, S=S, dS=dS, rowloss=rl)
            ops.linear_bwd_dw_adam_head_fold(H, X2, L1, adam, head, fa)
        else:
            dH = torch.empty(2 * B, Hd, device=DEV)
            ops.linear_fwd(X2, L1.W, L1.b, H, "relu")
            of.head_fwd_loss(variant, False, H, L2.W, L2.b, out_act, B, hyper, 1.0 / B, None, S, dS, rl, dH=dH)
            ops.linear_bwd_dw_adam_head(dH, X2, L1, adam, dict(head, dS=dS, rowloss=rl))
        torch.cuda.synchronize()
        # CPU autograd through the reference's loss lines
        h = torch.relu(X2c @ ref[0].t() + ref[1])
        s = act_cpu(h @ ref[2].t() + ref[3], out_act)[:, 0]
        lref = _oracle_tail_loss(variant, s, B, False)
        lref.backward()
        return dict(flat=fp.flat.clone(), grad=fp.grad.clone(), m=fp.m.clone(), v=fp.v.clone(), loss=loss.clone(),
                    S=S.clone(), dS=dS.clone(), rl=rl.clone(), views=[g.clone() for g in fp.gviews],
                    ref_grads=[p.grad for p in ref], ref_loss=lref.detach(), ref_s=s.detach())

    a, b = run(True), run(False)
    close(a["loss"][0], a["ref_loss"], 1e-5, "loss vs autograd", atol=1e-6)
    close(a["S"], a["ref_s"], 1e-5, "scores vs autograd")
    for g, r, n in zip(a["views"], a["ref_grads"], ("gW1", "gb1", "gw2", "gb2")):
        close(g, r, 2e-5, n + " vs autograd", atol=2e-8)
    for k in ("grad", "flat", "m", "v", "loss", "S", "dS", "rl"):
        close(a[k], b[k], 2e-5, k + " folded vs unfolded", atol=2e-8)
    a2 = run(True)                                  # bitwise reproducible launch after launch
    for k in ("grad", "flat", "m", "v", "loss", "S", "dS", "rl"):
        assert torch.equal(a[k], a2[k]), k


@pytest.mark.parametrize("variant,out_act", [("ns", "sigmoid"), ("ls", "sigmoid"), ("w", "id")])
@pytest.mark.parametrize("B,I,Hd", [(256, 784, 400), (24, 36, 20), (100, 64, 48), (33, 36, 20)])
def test_folded_head_generator_step(variant, out_act, B, I, Hd):
    """Generator step, folded: forward + partial dots, then dX through layer 1 with the loss / tick
    workgroup riding (dH formed in registers) against autograd on the CPU and the unfolded sequence;
    the counter advances exactly once."""
    from types import SimpleNamespace
    from generative_models_amd import ops_fused as of
    torch.manual_seed(B + I)
    hyper = [0.0, 1.0, 1.0]
    Xc = torch.rand(B, I).requires_grad_(True)
    W1c = torch.randn(Hd, I) / I ** 0.5; b1c = torch.randn(Hd) * 0.1
    w2c = torch.randn(1, Hd) / Hd ** 0.5; b2c = torch.randn(1)
    s = act_cpu(torch.relu(Xc @ W1c.t() + b1c) @ w2c.t() + b2c, out_act)[:, 0]
    lref = _oracle_tail_loss(variant, s, B, True)
    lref.backward()
    Xg, W1, b1 = Xc.detach().to(DEV), W1c.to(DEV), b1c.to(DEV)
    L2 = SimpleNamespace(W=w2c.to(DEV), b=b2c.to(DEV))
    below = torch.rand(B, I).to(DEV)                      # sigmoid output of the layer below (its gradient epilogue)
    H = torch.empty(B, Hd, device=DEV)
    fold = ops.HeadFold(B, Hd, DEV)
    S = torch.zeros(B, device=DEV); dS = torch.zeros(B, device=DEV); rl = torch.zeros(B, device=DEV)
    ctr = torch.ones(1, dtype=torch.int64, device=DEV)
    loss = torch.zeros(3, device=DEV); dX = torch.empty(B, I, device=DEV)
    ops.linear_fwd_headpart(Xg, W1, b1, H, "relu", L2, fold)
    ops.linear_bwd_dx_head_fold(H, W1, dX, dict(H=H, lin=L2, loss_out=loss, loss_slot=ops.slot(ctr.data_ptr(), 1, 0, 3, 1),
                                                inv_b=1.0 / B, B=B, gen_mode=True, tick=ctr),
                                fold.args(variant, out_act, hyper, S=S, dS=dS, rowloss=rl), below=below, epi="sigmoid")
    torch.cuda.synchronize()
    assert ctr.item() == 2
    close(loss[1], lref.detach(), 1e-5, "loss vs autograd", atol=1e-6)
    b_ = below.cpu()
    close(dX, Xc.grad * (b_ * (1 - b_)), 2e-5, "dX vs autograd", atol=2e-9)
    # unfolded sequence on the same inputs
    S2 = torch.empty(B, device=DEV); dS2 = torch.empty(B, device=DEV); rl2 = torch.empty(B, device=DEV)
    dH = torch.empty(B, Hd, device=DEV); dX2 = torch.empty(B, I, device=DEV); loss2 = torch.zeros(1, device=DEV)
    of.head_fwd_loss(variant, True, H, L2.W, L2.b, out_act, B, hyper, 1.0 / B, None, S2, dS2, rl2, dH=dH)
    of.head_bwd(H, dS2, L2.W, rl2, None, None, None, loss2, ops.NO_SLOT, 1.0 / B, True, B)
    ops.linear_bwd_dx(dH, W1, dX2, below=below, epi="sigmoid")
    close(dX, dX2, 2e-5, "dX folded vs unfolded", atol=2e-9)
    close(S, S2, 1e-5, "scores"); close(dS, dS2, 2e-5, "dS", atol=1e-9); close(rl, rl2, 1e-5, "row losses", atol=1e-7)
    close(loss[1], loss2[0], 1e-5, "loss folded vs unfolded", atol=1e-6)


def test_folded_head_refuses_shapes_it_cannot_carry():
    """No silent fallback: widths that are not multiples of 4 (no 16-byte operand path) raise."""
    from types import SimpleNamespace
    from generative_models_amd._lib import GMError
    B, I, Hd = 8, 30, 17
    H = torch.zeros(B, Hd, device=DEV); W1 = torch.zeros(Hd, I, device=DEV); dX = torch.zeros(B, I, device=DEV)
    L2 = SimpleNamespace(W=torch.zeros(1, Hd, device=DEV), b=torch.zeros(1, device=DEV))
    fold = ops.HeadFold(B, Hd, DEV)
    loss = torch.zeros(1, device=DEV)
    with pytest.raises(GMError):
        ops.linear_bwd_dx_head_fold(H, W1, dX, dict(H=H, lin=L2, loss_out=loss, loss_slot=ops.NO_SLOT, inv_b=1.0 / B,
                                                    B=B, gen_mode=True), fold.args("ns", "sigmoid"))


# ---------------------------------------------------------------------------------------------
# Bit-packed resident dataset (utils.py:31 binarises MNIST): the gather from
# 1 bit / pixel rows must equal the gather from the fp32 rows, on its own and riding in a GEMM launch
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,I,B", [(500, 784, 256), (77, 64, 16), (40, 100, 24), (33, 37, 9)])
def test_packed_dataset_gather_equals_fp32_gather(N, I, B):
    torch.manual_seed(N + I)
    data = torch.bernoulli(torch.full((N, I), 0.3)).cuda()
    packed = ops.PackedData(data)
    assert packed.shape == (N, I) and packed.wpr == (I + 31) // 32
    assert packed.nbytes() * 8 < data.numel() * 4 or I < 32        # ~32x smaller
    idx = torch.randint(0, N, (B,), device="cuda")
    a, b = torch.empty(B, I, device="cuda"), torch.full((B, I), -1.0, device="cuda")
    ops.gather_rows(data, idx, a)
    ops.gather_rows(packed, idx, b)
    assert torch.equal(a, data[idx]) and torch.equal(b, a)
    # riding in the generator's first forward launch (engine: gm_linear_fwd_gather_bits)
    x, W, bias = torch.randn(2 * B, 20, device="cuda"), torch.randn(48, 20, device="cuda"), torch.zeros(48, device="cuda")
    y1, y2 = torch.empty(2 * B, 48, device="cuda"), torch.empty(2 * B, 48, device="cuda")
    o1, o2 = torch.empty(B, I, device="cuda"), torch.full((B, I), -1.0, device="cuda")
    ops.linear_fwd_gather(x, W, bias, y1, "relu", data, idx, o1)
    ops.linear_fwd_gather(x, W, bias, y2, "relu", packed, idx, o2)
    assert torch.equal(y1, y2) and torch.equal(o1, o2) and torch.equal(o2, data[idx])
    assert not ops.PackedData.is_binary(torch.rand(4, 4))


# ---------------------------------------------------------------------------------------------
# Bit-packed rows as a GEMM operand: the gather copies words, the folded critic step's two
# launches expand them in registers -- every output bit equal to the fp32-row launches'
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,I,B", [(500, 784, 256), (77, 64, 16), (40, 100, 24), (300, 2100, 9)])
def test_packed_gather_copies_the_rows_as_words(N, I, B):
    torch.manual_seed(N + I)
    data = torch.bernoulli(torch.full((N, I), 0.3)).cuda()
    packed = ops.PackedData(data)
    idx = torch.randint(0, N, (B,), device="cuda")
    out = torch.full((B, packed.wpr), -1, dtype=torch.int32, device="cuda")
    ops.gather_rows(packed, idx, out)
    assert torch.equal(out, packed.bits[idx])
    x, W, bias = torch.randn(2 * B, 20, device="cuda"), torch.randn(48, 20, device="cuda"), torch.zeros(48, device="cuda")
    y1, y2 = torch.empty(2 * B, 48, device="cuda"), torch.empty(2 * B, 48, device="cuda")
    out2 = torch.full((B, packed.wpr), -1, dtype=torch.int32, device="cuda")
    ops.linear_fwd(x, W, bias, y1, "relu")
    ops.linear_fwd_gather(x, W, bias, y2, "relu", packed, idx, out2)
    assert torch.equal(y1, y2) and torch.equal(out2, out)


@pytest.mark.parametrize("variant,out_act", [("ns", "sigmoid"), ("w", "id")])
@pytest.mark.parametrize("B,I,Hd,rows", [(256, 784, 400, 256), (64, 784, 400, 64), (64, 36, 20, 64), (32, 64, 48, 32),
                                         (352, 784, 400, 352), (96, 100, 72, 32), (256, 784, 400, 128)])
def test_folded_critic_step_reads_packed_rows_bit_identically(variant, out_act, B, I, Hd, rows):
    """gm_linear_fwd_headpart_bits / gm_linear_bwd_dw_adam_head_fold_bits: the first `rows` rows of [x ; G(z)] come
    from the packed copy, the fp32 rows behind them hold garbage -- hidden layer, partial dots, gradients, Adam'd
    parameters, moments and loss must equal the fp32-row launches bit for bit (same MFMA sequence on the same values)."""
    import torch.nn as nn
    from generative_models_amd.engine import FlatParams, _Linear
    hyper = [0.0, 1.0, 1.0]
    torch.manual_seed(B + I)
    data = torch.bernoulli(torch.full((rows, I), 0.3))
    fake = torch.rand(2 * B - rows, I)

    def run(bits):
        torch.manual_seed(11)
        net = nn.Sequential(nn.Linear(I, Hd), nn.Linear(Hd, 1))
        fp = FlatParams(net.parameters(), DEV)
        fp.m.normal_().mul_(1e-3); fp.v.uniform_(0.0, 1e-4)
        L1, L2 = _Linear(fp, net[0]), _Linear(fp, net[1])
        X2 = torch.cat([data, fake]).to(DEV)
        xb = None
        if bits:
            pk = ops.PackedData(data.to(DEV))
            X2[:rows] = float("nan")                      # never read
            xb = (pk.bits, pk.wpr, rows)
        H = torch.empty(2 * B, Hd, device=DEV)
        S = torch.zeros(2 * B, device=DEV); dS = torch.zeros_like(S); rl = torch.zeros_like(S)
        loss = torch.zeros(1, device=DEV)
        sched = torch.from_numpy(ops.adam_schedule(2e-4, 4)).to(DEV)
        adam = dict(sched=sched, sched_slot=ops.slot(0, 0, 2, 0, 1), clamp=0.0)
        head = dict(H=H, lin=L2, loss_out=loss, loss_slot=ops.NO_SLOT, inv_b=1.0 / B, B=B, adam=adam)
        fold = ops.HeadFold(2 * B, Hd, DEV)
        ops.linear_fwd_headpart(X2, L1.W, L1.b, H, "relu", L2, fold, xbits=xb)
        fa = fold.args(variant, out_act, hyper, S=S, dS=dS, rowloss=rl)
        ops.linear_bwd_dw_adam_head_fold(H, X2, L1, adam, head, fa, xbits=xb)
        torch.cuda.synchronize()
        return dict(H=H, part=fold.part.clone(), flat=fp.flat.clone(), grad=fp.grad.clone(), m=fp.m.clone(),
                    v=fp.v.clone(), loss=loss.clone(), S=S, dS=dS, rl=rl)

    a, b = run(False), run(True)
    for k in a:
        assert not torch.isnan(b[k]).any(), k
        assert torch.equal(a[k], b[k]), k


def test_packed_operand_refuses_what_it_cannot_carry():
    """No fp32 copy exists behind packed rows: a row count that is not whole 32-row tiles is an error, never a
    silent read of the unwritten rows."""
    from types import SimpleNamespace
    B, I, Hd = 24, 64, 48
    pk = ops.PackedData(torch.bernoulli(torch.full((B, I), 0.5)).to(DEV))
    X2 = torch.zeros(2 * B, I, device=DEV); H = torch.zeros(2 * B, Hd, device=DEV)
    W1 = torch.zeros(Hd, I, device=DEV); b1 = torch.zeros(Hd, device=DEV)
    L2 = SimpleNamespace(W=torch.zeros(1, Hd, device=DEV), b=torch.zeros(1, device=DEV))
    fold = ops.HeadFold(2 * B, Hd, DEV)
    from generative_models_amd._lib import GMError
    with pytest.raises(GMError):
        ops.linear_fwd_headpart(X2, W1, b1, H, "relu", L2, fold, xbits=(pk.bits, pk.wpr, B))


def test_stage_in_gate_waits_for_the_host_and_times_out_without_hanging():
    """gm_stage_in_gated: the kernel copies only after the fill counter in pinned host memory covers
    its iterations; a counter that never advances ends in a bounded wait + the time-out flag (the GPU
    is never left hanging on a host that died)."""
    import ctypes
    import threading
    import time
    from generative_models_amd import _lib
    R, n = 8, 64
    host = torch.zeros(R, n).pin_memory()
    dev = torch.full((R, n), -1.0, device="cuda")
    gate = torch.zeros(2, dtype=torch.int64).pin_memory()
    hp, gp = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.call("gm_host_device_ptr", host.data_ptr(), ctypes.byref(hp))
    _lib.call("gm_host_device_ptr", gate.data_ptr(), ctypes.byref(gp))
    segs = (_lib.StageSeg * 1)(_lib.StageSeg(hp.value, dev.data_ptr(), n * 4))
    st = ops.stream_ptr()
    it = 3                                              # iterations [3, 5) -> ring slots 3, 4

    def launch(timeout_s):
        _lib.call("gm_stage_in_gated", st, segs, 1, ops.slot(0, 0, it, R, 1), 2, gp.value,
                  ops.slot(0, 0, it, 0, 1), timeout_s, None, 256)

    # 1. the host fills AFTER the launch: the kernel must wait for it
    def fill():
        time.sleep(0.05)
        host[3:5] = torch.arange(2 * n, dtype=torch.float32).view(2, n)
        gate.numpy()[0] = 5                            # iterations < 5 are written
    th = threading.Thread(target=fill)
    launch(5.0)
    th.start()
    torch.cuda.synchronize()
    th.join()
    assert int(gate[1]) == 0
    assert torch.equal(dev[3:5].cpu(), host[3:5]) and bool((dev[:3] == -1).all()) and bool((dev[5:] == -1).all())
    # 2. the fill never comes: bounded wait, flag raised, the stream keeps working
    it = 6
    t0 = time.perf_counter()
    launch(0.05)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert int(gate[1]) == 1 and 0.04 < dt < 2.0, (int(gate[1]), dt)
    x = torch.ones(4, device="cuda") * 2
    assert float(x.sum()) == 8.0


@pytest.mark.parametrize("B,Bl,w,dtype", [(64, 16, 20, torch.float32), (24, 8, 1, torch.int64), (12, 3, 5, torch.float32)])
def test_stage_in_strided_pieces_copy_only_the_ranks_rows(B, Bl, w, dtype):
    """gm_stage_seg with blocks / block strides: a data-parallel rank stages ITS rows of every [B, w] draw
    of the global batch (host ring: global rows; device ring: the rank's rows) -- pieces of an iteration
    that are adjacent neither in the source nor in the destination; 16-byte and 4-byte paths."""
    import ctypes
    from generative_models_amd import _lib
    R, m, rank = 8, 3, 2                                 # ring of 8 iterations, 3 draws per iteration
    host = (torch.arange(R * m * B * w, dtype=torch.float32).view(R * m, B, w) + 1).to(dtype).pin_memory()
    dev = torch.full((R * m, Bl, w), -1, dtype=dtype, device="cuda")
    hp = ctypes.c_void_p()
    _lib.call("gm_host_device_ptr", host.data_ptr(), ctypes.byref(hp))
    es = host.element_size()
    wb = w * es
    seg = _lib.StageSeg(hp.value + rank * Bl * wb, dev.data_ptr(), m * Bl * wb, m, 0, B * wb, Bl * wb)
    segs = (_lib.StageSeg * 1)(seg)
    it, k = 5, 2                                          # iterations [5, 7) -> ring slots 5, 6
    _lib.call("gm_stage_in", ops.stream_ptr(), segs, 1, ops.slot(0, 0, it, R, 1), k)
    torch.cuda.synchronize()
    want = torch.full_like(dev, -1).cpu()
    want[it * m:(it + k) * m] = host[it * m:(it + k) * m, rank * Bl:(rank + 1) * Bl]
    assert torch.equal(dev.cpu(), want)


@pytest.mark.parametrize("M,K,N,rows", [(512, 400, 784, 256), (96, 48, 64, 40), (2048, 400, 784, 1024)])
def test_linear_fwd_with_interp_epilogue_equals_separate_launches(M, K, N, rows):
    """gm_linear_fwd_interp: the generator's last layer also writes WGAN-GP's x_hat for its first
    `rows` rows -- bit-identical to gm_linear_fwd followed by gm_interp (w_gp_gan.py:197-201), for the
    split-reduction kernel and the LDS macro-tile kernel (M >= 1024) alike."""
    from generative_models_amd import ops_fused as of
    torch.manual_seed(M + N)
    dev = "cuda"
    x = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev) * 0.1
    real = torch.bernoulli(torch.full((rows, N), 0.3, device=dev))
    eps = torch.rand(rows, device=dev)
    y0, y1 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    xh0, xh1 = torch.zeros(rows, N, device=dev), torch.full((rows, N), -3.0, device=dev)
    ops.linear_fwd(x, W, b, y0, "sigmoid")
    of.interp(eps, ops.NO_SLOT, real, y0[:rows], xh0)
    ops.linear_fwd_interp(x, W, b, y1, "sigmoid", eps, ops.NO_SLOT, real, xh1, rows)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1) and torch.equal(xh0, xh1)
    e, r, g = eps.cpu()[:, None], real.cpu(), y0[:rows].cpu()
    ref = e * r + (1 - e) * g                          # torch CPU: two rounded products, one add
    assert torch.equal(xh1.cpu(), ref)
