"""Per-op parity of the variant-specific fused kernels (csrc/gm_fused.hip) against torch -- each
kernel on its own, through the C-ABI, so that two compensating errors cannot hide behind an
end-to-end trainer test.  References are torch autograd in fp64 of the reference's own
expressions:
  WGAN-GP   w_gp_gan.py:195-218   (gm_interp, gm_gp_u, gm_gp_norm, gm_gp_dw2)
  DRAGAN    dra_gan.py:198-223    (gm_std_all, gm_dragan_xhat, gm_dragan_rows, gm_dragan_head_bwd)
  VAE       vae.py:100-106,203,212 (gm_vae_reparam, gm_vae_reparam_bwd, gm_sqerr_sigmoid_bwd)
Tolerance: 2e-6 relative to the tensor's scale for element-wise outputs (fp32 kernels vs an fp64
reference), 1e-5 for sums over the batch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from generative_models_amd import ops, ops_fused as of  # noqa: E402

DEV = "cuda"
LAM = 10.0


def rel(got, ref):
    ref = ref.to(torch.float64)
    return float((got.to(torch.float64) - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def critic(B=48, I=100, H=72, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return dict(B=B, I=I, H=H, W1=(r(H, I) / I ** 0.5), b1=r(H) * 0.1, w2=(r(1, H) / H ** 0.5),
                b2=r(1) * 0.1 + 0.3, x=torch.rand(B, I, generator=g), gz=torch.rand(B, I, generator=g))


@pytest.mark.parametrize("B,I", [(16, 64), (48, 100), (256, 784)])
def test_interp(B, I):
    """x_hat = eps*x + (1-eps)*G(z), eps [B,1] broadcast (w_gp_gan.py:197-201)."""
    torch.manual_seed(1)
    eps, x, g = torch.rand(B), torch.rand(B, I), torch.rand(B, I)
    out = torch.empty(B, I, device=DEV)
    of.interp(eps.to(DEV), ops.NO_SLOT, x.to(DEV), g.to(DEV), out)
    ref = eps[:, None].double() * x.double() + (1 - eps[:, None].double()) * g.double()
    assert rel(out.cpu(), ref) < 2e-6


def test_wgangp_penalty_chain_vs_autograd():
    """gp_u, gp_norm (incl. a zero-gradient row), gp_dw2 against autograd through the ReLU critic."""
    c = critic()
    B, I, H = c["B"], c["I"], c["H"]
    f64 = lambda t: t.double().clone().requires_grad_(True)
    W1, b1, w2, b2 = f64(c["W1"]), f64(c["b1"]), f64(c["w2"]), f64(c["b2"])
    eps = torch.rand(B, 1, generator=torch.Generator().manual_seed(3)).double()
    xh = (eps * c["x"].double() + (1 - eps) * c["gz"].double()).requires_grad_(True)
    # two rows whose critic output is clamped by the output ReLU: their input gradient is exactly 0
    with torch.no_grad():
        b2_eff = b2.clone()
    h = torch.relu(xh @ W1.t() + b1)
    a2 = h @ w2.t() + b2_eff
    dead = (a2.detach().view(-1) <= 0)
    s = torch.relu(a2)
    grads = torch.autograd.grad(s, xh, grad_outputs=torch.ones_like(s), create_graph=True)[0]
    n = grads.norm(2, dim=1)
    P = LAM * torch.mean((n - 1) ** 2)
    dW1, dw2 = torch.autograd.grad(P, [W1, w2], allow_unused=True)

    d = lambda t: t.detach().float().to(DEV).contiguous()
    Hh, Sh = d(h), d(s.view(-1))
    U = torch.empty(B, H, device=DEV)
    of.gp_u(Sh, Hh, d(w2), U)
    u_ref = ((a2.detach() > 0).double() * (h.detach() > 0).double() * w2.detach())
    assert rel(U.cpu(), u_ref) < 2e-6
    Gr = torch.empty(B, I, device=DEV)
    ops.linear_bwd_dx(U, d(W1), Gr)                               # g = u W1
    assert rel(Gr.cpu(), grads.detach()) < 5e-6
    if dead.any():                                                # norm 0 -> gamma 0 (torch's subgradient)
        assert float(Gr[dead.to(DEV)].abs().max()) == 0.0
    Gam, pen = torch.empty(B, I, device=DEV), torch.empty(B, device=DEV)
    inv_b = float(np.float32(1.0) / np.float32(B))
    of.gp_norm(Gr, Gam, pen, LAM, inv_b)
    n_d = n.detach()
    gamma_ref = torch.where(n_d[:, None] > 0, (2 * LAM / B) * (n_d[:, None] - 1) * grads.detach() /
                            n_d[:, None].clamp_min(1e-300), torch.zeros_like(grads.detach()))
    assert rel(Gam.cpu(), gamma_ref) < 5e-6
    assert rel(pen.cpu(), (n_d - 1) ** 2) < 5e-6
    if dead.any():
        assert float(Gam[dead.to(DEV)].abs().max()) == 0.0 and torch.all(pen[dead.to(DEV)] == 1.0)
    # second backward: dP/dW1 = u^T gamma, dP/dw2 = sum_b m2 m1 . (gamma W1^T)
    gW1 = torch.zeros(H, I, device=DEV)
    ops.linear_bwd_dw(U, Gam, gW1, None, accumulate=True)
    assert rel(gW1.cpu(), dW1) < 1e-5
    T = torch.empty(B, H, device=DEV)
    ops.linear_fwd(Gam, d(W1), None, T, "id")
    gw2 = torch.full((1, H), 0.5, device=DEV)                     # accumulates on top of what is there
    of.gp_dw2(Sh, Hh, T, gw2)
    assert rel(gw2.cpu() - 0.5, dw2) < 1e-5


def test_gp_norm_zero_row_is_exactly_zero():
    B, I = 8, 40
    g = torch.randn(B, I)
    g[3] = 0.0
    Gam, pen = torch.empty(B, I, device=DEV), torch.empty(B, device=DEV)
    of.gp_norm(g.to(DEV), Gam, pen, LAM, 1.0 / B)
    assert float(Gam[3].abs().max()) == 0.0 and float(pen[3]) == 1.0
    assert torch.isfinite(Gam).all()


@pytest.mark.parametrize("B,Z", [(16, 8), (512, 20), (336, 20)])
def test_vae_reparam_forward_and_backward(B, Z):
    """z = mu + eps*exp(lv/2), kl = sum 0.5(mu^2 + e^lv - lv - 1) and d(recon+kl)/d[mu|lv] from dz."""
    torch.manual_seed(2)
    ml, eps, dz = torch.randn(B, 2 * Z) * 0.5, torch.randn(B, Z), torch.randn(B, Z)
    m64 = ml.double().requires_grad_(True)
    mu, lv = m64[:, :Z], m64[:, Z:]
    z_ref = mu + eps.double() * torch.exp(lv / 2)
    kl_ref = torch.sum(0.5 * (mu ** 2 + torch.exp(lv) - lv - 1))
    (z_ref * dz.double()).sum().add(kl_ref).backward()
    z, kl = torch.empty(B, Z, device=DEV), torch.zeros(3, device=DEV)
    of.vae_reparam(ml.to(DEV), eps.to(DEV), z, kl, B, Z, kl_slot=ops.slot(0, 0, 1, 0, 1))
    assert rel(z.cpu(), z_ref.detach()) < 2e-6
    kl_ref = float(kl_ref.detach())
    assert abs(float(kl[1]) - kl_ref) <= 1e-5 * max(1.0, abs(kl_ref))
    assert float(kl[0]) == 0.0 and float(kl[2]) == 0.0             # only its own slot
    dml = torch.empty(B, 2 * Z, device=DEV)
    of.vae_reparam_bwd(ml.to(DEV), eps.to(DEV), dz.to(DEV), dml, B, Z)
    assert rel(dml.cpu(), m64.grad) < 5e-6


@pytest.mark.parametrize("B,I", [(16, 64), (512, 784), (336, 784)])
def test_sqerr_sigmoid_backward(B, I):
    """recon = sum (x - sigmoid(a))^2 (vae.py:203): per-row partial sums and d recon / d a."""
    torch.manual_seed(3)
    x = (torch.rand(B, I) < 0.13).float()
    a = (torch.randn(B, I)).double().requires_grad_(True)
    xr = torch.sigmoid(a)
    loss = torch.sum((x.double() - xr) ** 2)
    loss.backward()
    dA, part = torch.empty(B, I, device=DEV), torch.empty(B, device=DEV)
    of.sqerr_sigmoid_bwd(x.to(DEV), xr.detach().float().to(DEV), dA, part, B)
    assert rel(dA.cpu(), a.grad) < 5e-6
    assert rel(part.cpu(), ((x.double() - xr.detach()) ** 2).sum(1)) < 5e-6
    out = torch.zeros(2, device=DEV)
    of.sum_finalize(part, B, out, out_slot=ops.slot(0, 0, 1, 0, 1))
    assert abs(float(out[1]) - float(loss)) <= 1e-5 * float(loss) and float(out[0]) == 0.0


@pytest.mark.parametrize("B,H,I", [(512, 400, 784), (336, 400, 784), (16, 48, 64), (40, 32, 36), (1024, 400, 784)])
def test_reconstruction_loss_in_the_decoder_forward_epilogue(B, H, I):
    """gm_linear_fwd_sqerr (vae.py:75-77 + :203): x_hat and dA bit-identical to gm_linear_fwd(sigmoid)
    followed by gm_sqerr_sigmoid_bwd, the row-tile partials add up to sum (x - x_hat)^2, stale entries
    of the partial array beyond this batch's rows are not read, and a relaunch reproduces every bit."""
    torch.manual_seed(B + I)
    x = (torch.rand(B, I) < 0.13).float().to(DEV)
    h = torch.relu(torch.randn(B, H)).to(DEV)
    W = (torch.randn(I, H) / H ** 0.5).to(DEV); b = (torch.randn(I) * 0.1).to(DEV)
    y0, dA0, part0 = torch.empty(B, I, device=DEV), torch.empty(B, I, device=DEV), torch.empty(B, device=DEV)
    ops.linear_fwd(h, W, b, y0, "sigmoid")
    of.sqerr_sigmoid_bwd(x, y0, dA0, part0, B)
    ldp = ((I + 31) // 32 + 3) // 4 * 4
    part = torch.zeros(B + 8, ldp, device=DEV)
    part[B:] = 77.0                                           # rows of an earlier, larger batch
    y1, dA1 = torch.empty(B, I, device=DEV), torch.full((B, I), 5.0, device=DEV)
    ops.linear_fwd_sqerr(h, W, b, y1, x, dA1, part, M=B)
    torch.cuda.synchronize()
    if B < 1024:
        assert torch.equal(y0, y1) and torch.equal(dA0, dA1)
    else:                                                     # gm_linear_fwd takes the LDS kernel there
        assert rel(y1.cpu(), y0.cpu().double()) < 2e-6
    ref_rows = ((x.double() - y1.double()) ** 2).sum(1).cpu()
    assert rel(part[:B].double().sum(1).cpu(), ref_rows) < 2e-6
    out = torch.zeros(2, device=DEV)
    of.sum_finalize(part, B * ldp, out, out_slot=ops.slot(0, 0, 1, 0, 1))
    assert abs(float(out[1]) - float(ref_rows.sum())) <= 2e-6 * float(ref_rows.sum()) and float(out[0]) == 0.0
    p1 = part.clone()
    ops.linear_fwd_sqerr(h, W, b, y1, x, dA1, part, M=B)
    torch.cuda.synchronize()
    assert torch.equal(p1, part)


@pytest.mark.parametrize("B,Z,H", [(512, 20, 400), (336, 20, 400), (16, 8, 48), (37, 4, 20)])
def test_reparam_backward_in_the_dx_epilogue(B, Z, H):
    """gm_linear_bwd_dx_reparam (autograd of vae.py:100-106,210-212 behind the decoder's first layer):
    dz and d loss / d [mu | log_var] bit-identical to gm_linear_bwd_dx + gm_vae_reparam_bwd, reading the
    noise through a ring slot."""
    torch.manual_seed(B + Z)
    dH = torch.randn(B, H).to(DEV)
    W = (torch.randn(H, Z) / Z ** 0.5).to(DEV)
    ml = (torch.randn(B, 2 * Z) * 0.5).to(DEV)
    ring = torch.randn(3, B, Z).to(DEV)                       # eps ring, slot 2 is this step's
    slot = ops.slot(0, 0, 2, 3, B * Z)
    dz0, dml0 = torch.empty(B, Z, device=DEV), torch.empty(B, 2 * Z, device=DEV)
    ops.linear_bwd_dx(dH, W, dz0)
    of.vae_reparam_bwd(ml, ring.view(-1), dz0, dml0, B, Z, eps_slot=slot)
    dz1, dml1 = torch.empty(B, Z, device=DEV), torch.full((B, 2 * Z), 3.0, device=DEV)
    ops.linear_bwd_dx_reparam(dH, W, dz1, ml, ring.view(-1), dml1, eps_slot=slot)
    torch.cuda.synchronize()
    assert torch.equal(dz0, dz1) and torch.equal(dml0, dml1)
    # and against autograd (fp64)
    mu = ml[:, :Z].double().cpu().requires_grad_(True); lv = ml[:, Z:].double().cpu().requires_grad_(True)
    z = mu + ring[2].double().cpu() * torch.exp(lv / 2)
    kl = torch.sum(0.5 * (mu ** 2 + torch.exp(lv) - lv - 1))
    (torch.sum(z * dz1.double().cpu()) + kl).backward()
    assert rel(dml1[:, :Z].cpu(), mu.grad) < 5e-6 and rel(dml1[:, Z:].cpu(), lv.grad) < 5e-6


@pytest.mark.parametrize("B,I", [(16, 64), (256, 784)])
def test_std_all_is_the_unbiased_std_of_the_whole_batch(B, I):
    """images.data.std() (dra_gan.py:204): over all B*I elements, Bessel-corrected."""
    torch.manual_seed(4)
    x = (torch.rand(B, I) < 0.13).float()
    out = torch.empty(1, device=DEV)
    of.std_all(x.to(DEV), B, out)
    assert abs(float(out) - float(x.double().std())) < 1e-6


@pytest.mark.parametrize("B,I,ld", [(256, 784, 784), (64, 100, 128), (7, 33, 33), (3, 2, 5), (1024, 784, 784)])
def test_std_kernels_share_a_workspace_across_launches(B, I, ld):
    """The 64 workgroups' partial sums + arrival counter live in a caller-owned workspace that re-arms
    itself: repeated launches (strided rows, widths that are not a multiple of 4, the data-parallel
    (sum, sum of squares) form) keep giving the fp64 answer, and equal inputs give equal bits."""
    torch.manual_seed(B + I)
    ws = of.std_workspace(DEV)
    out, sums, out2 = torch.empty(1, device=DEV), torch.empty(2, device=DEV), torch.empty(1, device=DEV)
    first = None
    for k in range(4):
        buf = torch.rand(B, ld, device=DEV) * (k + 1)
        x = buf[:, :I]
        of.std_all(x, B, out, ws=ws)
        want = float(x.double().cpu().std())
        assert abs(float(out) - want) < 2e-6 * max(1.0, want)
        of.std_sums(x, B, sums, ws=ws)
        of.std_from_sums(sums, B * I, out2)
        xd = x.double().cpu()
        assert abs(float(sums[0]) - float(xd.sum())) <= 1e-6 * float(xd.sum())
        assert abs(float(sums[1]) - float((xd * xd).sum())) <= 1e-6 * float((xd * xd).sum())
        assert abs(float(out2) - want) < 1e-4 * max(1.0, want)      # the sums travel as fp32 (engine: exchanged)
        of.std_all(x, B, out2, ws=ws)
        assert torch.equal(out, out2)
    assert int(ws.view(torch.int32)[2 * 2 * 64]) == 0               # counter re-armed


def test_dragan_xhat():
    """x_hat = delta*x + (1-delta)*(x + C*std*U)  (dra_gan.py:200-205)."""
    B, I = 48, 100
    torch.manual_seed(5)
    x, delta, U = (torch.rand(B, I) < 0.13).float(), torch.rand(B), torch.rand(B, I)
    std = x.std()
    out = torch.empty(B, I, device=DEV)
    of.dragan_xhat(x.to(DEV), delta.to(DEV), ops.NO_SLOT, U.to(DEV), ops.NO_SLOT,
                   std.reshape(1).to(DEV), out, B)
    ref = delta[:, None].double() * x.double() + (1 - delta[:, None].double()) * (x.double() + std.double() * U.double())
    assert rel(out.cpu(), ref) < 2e-6


def test_dragan_penalty_chain_vs_autograd():
    """gm_gp_u (sigmoid critic: m2 = 1) + dX GEMM + gm_dragan_rows + gm_dragan_head_bwd + the three
    accumulating GEMMs of engine._issue_dra_backward against autograd's double backward through the
    SIGMOID critic (the sigma'' path)."""
    c = critic(seed=7)
    B, I, H = c["B"], c["I"], c["H"]
    f64 = lambda t: t.double().clone().requires_grad_(True)
    W1, b1, w2, b2 = f64(c["W1"]), f64(c["b1"]), f64(c["w2"]), f64(c["b2"])
    xh = (c["x"].double() + 0.3 * c["gz"].double()).requires_grad_(True)
    h = torch.relu(xh @ W1.t() + b1)
    s = torch.sigmoid(h @ w2.t() + b2)
    grads = torch.autograd.grad(s, xh, grad_outputs=torch.ones_like(s), create_graph=True)[0]
    n = grads.norm(2, dim=1)
    P = LAM * torch.mean((n - 1) ** 2)
    dW1, db1, dw2, db2 = torch.autograd.grad(P, [W1, b1, w2, b2])

    d = lambda t: t.detach().float().to(DEV).contiguous()
    Xh, Hh, Sh = d(xh), d(h), d(s.view(-1))
    U, V = torch.empty(B, H, device=DEV), torch.empty(B, I, device=DEV)
    of.gp_u(Sh, Hh, d(w2), U)                                     # m1 . w2
    ops.linear_bwd_dx(U, d(W1), V)                                # v = (m1 . w2) W1
    dv, da2, pen = torch.empty(B, I, device=DEV), torch.empty(B, device=DEV), torch.empty(B, device=DEV)
    inv_b = float(np.float32(1.0) / np.float32(B))
    of.dragan_rows(Sh, V, dv, da2, pen, LAM, inv_b, B)
    assert rel(pen.cpu(), (n.detach() - 1) ** 2) < 1e-5
    gW1, gb1 = torch.zeros(H, I, device=DEV), torch.zeros(H, device=DEV)
    gw2, gb2 = torch.zeros(1, H, device=DEV), torch.zeros(1, device=DEV)
    ops.linear_bwd_dw(U, dv, gW1, None, accumulate=True)          # (m1 . w2)^T dv
    T = torch.empty(B, H, device=DEV)
    ops.linear_fwd(dv, d(W1), None, T, "id")                      # du = dv W1^T
    dA1 = torch.empty(B, H, device=DEV)
    of.dragan_head_bwd(Hh, T, da2, d(w2), gw2, gb2, dA1, B)
    ops.linear_bwd_dw(dA1, Xh, gW1, gb1, accumulate=True)         # da1^T x_hat, sum da1
    assert rel(gW1.cpu(), dW1) < 2e-5
    assert rel(gb1.cpu(), db1) < 2e-5
    assert rel(gw2.cpu(), dw2) < 2e-5
    assert rel(gb2.cpu(), db2) < 2e-5


def test_head_gp_and_stacked_weight_gradient():
    """The fused WGAN-GP critic step's building blocks: gm_head_gp (D(x_hat)'s N = 1 layer + u in one
    launch) == linear + relu + gp_u; gm_gp_dw2_store == gp_dw2 on a zeroed buffer; the stacked
    weight-gradient GEMM [u ; dH]^T [gamma ; X] with ones_from == the two separate GEMMs, and its
    bias gradient only sees the dH rows."""
    torch.manual_seed(6)
    B, I, H = 48, 100, 72
    h = torch.relu(torch.randn(B, H)).to(DEV)
    w2, b2 = (torch.randn(1, H) / H ** 0.5).to(DEV), torch.tensor([0.05], device=DEV)
    s, u = torch.empty(B, device=DEV), torch.empty(B, H, device=DEV)
    of.head_gp(h, w2, b2, s, u)
    s_ref = torch.relu(h.double() @ w2.double().t() + b2.double()).view(-1)
    assert rel(s.cpu(), s_ref.cpu()) < 2e-6
    u_ref = (s_ref[:, None] > 0).double() * (h.double() > 0).double() * w2.double()
    assert torch.equal(u.cpu().double(), u_ref.cpu())
    t = torch.randn(B, H, device=DEV)
    a, b = torch.zeros(1, H, device=DEV), torch.full((1, H), 7.0, device=DEV)
    of.gp_dw2(s, h, t, a)
    of.gp_dw2_store(s, h, t, b)
    assert torch.equal(a, b)
    # stacked dW: rows [0, B) = (u, gamma) penalty rows, rows [B, 3B) = (dH, X) first-order rows
    from types import SimpleNamespace
    gam, X = torch.randn(B, I, device=DEV), torch.rand(2 * B, I, device=DEV)
    dH = torch.randn(2 * B, H, device=DEV) * 0.1
    DU, XX = torch.cat([u, dH]).contiguous(), torch.cat([gam, X]).contiguous()
    z = lambda *sh: torch.zeros(*sh, device=DEV)
    lin = SimpleNamespace(W=z(H, I), b=z(H), gW=z(H, I), gb=z(H), mW=z(H * I), vW=z(H * I), mb=z(H), vb=z(H))
    head_lin = SimpleNamespace(W=w2.clone(), b=b2.clone(), gW=z(1, H), gb=z(1), mW=z(H), vW=z(H), mb=z(1), vb=z(1))
    dS, rl, lo = torch.randn(2 * B, device=DEV) / B, torch.rand(2 * B, device=DEV), z(1)
    Hd = torch.relu(torch.randn(2 * B, H, device=DEV))
    add = torch.randn(H, device=DEV)
    head = dict(H=Hd, dS=dS, lin=head_lin, rowloss=rl, loss_out=lo, loss_slot=ops.NO_SLOT, inv_b=1.0 / B,
                B=B, adam=None, gw2_add=add)
    ops.linear_bwd_dw_adam_head(DU, XX, lin, None, head, M=3 * B, ones_from=B)
    gW_ref = u.double().t() @ gam.double() + dH.double().t() @ X.double()
    assert rel(lin.gW.cpu(), gW_ref.cpu()) < 2e-6
    assert rel(lin.gb.cpu(), dH.double().sum(0).cpu()) < 2e-6          # no bias share from the u rows
    gw2_ref = dS.double() @ Hd.double() + add.double()
    assert rel(head_lin.gW.view(-1).cpu(), gw2_ref.cpu()) < 2e-6
    # the penalty's share of w2's gradient summed by the head's own workgroups (no gm_gp_dw2_store launch):
    # same gradient as handing over the stored sums
    head2 = dict(head, gw2_add=None, pen=dict(s=s, h=h, t=t))
    lin.gW.zero_(); lin.gb.zero_(); head_lin.gW.zero_()
    ops.linear_bwd_dw_adam_head(DU, XX, lin, None, head2, M=3 * B, ones_from=B)
    pen_ref = ((s_ref[:, None] > 0) & (h.double() > 0)).double().mul(t.double()).sum(0)
    assert rel(head_lin.gW.view(-1).cpu(), (dS.double() @ Hd.double() + pen_ref).cpu()) < 2e-6
    assert rel(lin.gW.cpu(), gW_ref.cpu()) < 2e-6


@pytest.mark.parametrize("B,Z", [(256, 20), (512, 20), (336, 20), (37, 6), (100, 64), (48, 8), (70, 32)])
def test_bir_mmd_matches_fp64_autograd(B, Z):
    """gm_bir_mmd: Gaussian-kernel MMD of bir_vae.py:201-221 and d(lam * mmd)/dz vs fp64 autograd."""
    torch.manual_seed(B + Z)
    z = (torch.randn(B, Z) * 0.7 + 0.2).cuda()
    x = torch.randn(B, Z).cuda()
    part, dz, out = torch.zeros(B, device="cuda"), torch.zeros(B, Z, device="cuda"), torch.zeros(1, device="cuda")
    lam = 1000.0
    of.bir_mmd(z, x.view(-1), part, dz, B, Z, lam)
    of.sum_finalize(part, B, out, scale=lam)
    torch.cuda.synchronize()
    zd = z.double().cpu().requires_grad_(True)
    xd = x.double().cpu()

    def kern(a, b):
        return torch.exp(-((a.unsqueeze(1) - b.unsqueeze(0)) ** 2).mean(2) / Z)
    mmd = lam * (kern(xd, xd).sum() + kern(zd, zd).sum() - 2 * kern(xd, zd).sum())
    mmd.backward()
    assert abs(out.item() - mmd.item()) <= 1e-4 * max(10.0, abs(mmd.item())), (out.item(), mmd.item())
    g = zd.grad
    assert (dz.cpu().double() - g).abs().max().item() <= 2e-5 * max(1.0, g.abs().max().item())
    # forward only (evaluate): dz untouched
    dz.fill_(7.0)
    of.bir_mmd(z, x.view(-1), part, None, B, Z, lam)
    torch.cuda.synchronize()
    assert bool((dz == 7.0).all())


@pytest.mark.parametrize("B,Z", [(512, 20), (336, 20), (37, 6)])
def test_vae_reparam_wide_and_dual_finalize(B, Z):
    """gm_vae_reparam_wide + gm_sum_finalize2_tick == gm_vae_reparam + gm_sum_finalize: same z bit for
    bit, KL and reconstruction sums to fp64-summation order, counter advanced by the last launch."""
    torch.manual_seed(B)
    ml = (torch.randn(B, 2 * Z) * 0.5).cuda()
    eps = torch.randn(B * Z).cuda()
    z0, z1 = torch.empty(B, Z, device="cuda"), torch.empty(B, Z, device="cuda")
    kl0 = torch.zeros(1, device="cuda")
    of.vae_reparam(ml, eps, z0, kl0, B, Z)
    part_kl = torch.full(((B * Z + 255) // 256 + 3,), 7.0, device="cuda")
    n_kl = of.vae_reparam_wide(ml, eps, z1, part_kl, B, Z)
    rows = torch.rand(B, device="cuda") * 50
    out_a, out_b = torch.zeros(2, device="cuda"), torch.zeros(2, device="cuda")
    ctr = torch.zeros(1, dtype=torch.int64, device="cuda")
    of.sum_finalize2(rows, B, out_a, ops.slot(0, 0, 1, 0, 1), part_kl, n_kl, out_b, ops.slot(0, 0, 1, 0, 1), tick=ctr)
    torch.cuda.synchronize()
    assert torch.equal(z0, z1)
    assert abs(out_b[1].item() - kl0.item()) <= 1e-6 * max(1.0, abs(kl0.item()))
    assert abs(out_a[1].item() - rows.double().sum().item()) <= 1e-6 * rows.double().sum().item()
    assert out_a[0].item() == 0.0 and out_b[0].item() == 0.0 and int(ctr) == 1
    assert bool((part_kl[n_kl:] == 7.0).all())


@pytest.mark.parametrize("B,Z,N", [(512, 20, 400), (336, 20, 400), (100, 20, 400), (37, 8, 50), (512, 32, 400),
                                    (64, 16, 33), (2048, 20, 400)])
def test_vae_reparam_and_first_decoder_layer_in_one_launch(B, Z, N):
    """gm_vae_reparam_fwd == gm_vae_reparam_wide followed by gm_linear_fwd (vae.py:100-106, :113): z and the KL
    partials bit for bit (same workgroups, same code); h bit for bit wherever the separate forward runs the 16-wave
    kernel (its waves 0 and 1 own the two 16-deep chunks and the reduction adds them in that order) and to fp32
    rounding of a 20-term dot where it runs the many-row kernel.  The unfused product-then-add of z is what torch's
    `mu + eps * std` computes."""
    torch.manual_seed(B + Z)
    ml = (torch.randn(B, 2 * Z) * 0.5).cuda()
    ring = torch.randn(3, B * Z).cuda()
    ctr = torch.zeros(1, dtype=torch.int64, device="cuda")
    slot = ops.slot(ctr.data_ptr(), 1, 0, 3, B * Z)            # ring slot = the device counter's value
    W = (torch.randn(N, Z) * 0.3).cuda()
    bias = torch.randn(N).cuda()
    n_part = (B * Z + 255) // 256
    z0, z1 = torch.empty(B, Z, device="cuda"), torch.empty(B, Z, device="cuda")
    p0, p1 = torch.full((n_part + 2,), 7.0, device="cuda"), torch.full((n_part + 2,), 7.0, device="cuda")
    h0, h1 = torch.empty(B, N, device="cuda"), torch.full((B, N), -3.0, device="cuda")
    for step in range(2):                                    # second ring slot on the second pass
        ctr.fill_(step)
        n0 = of.vae_reparam_wide(ml, ring.view(-1), z0, p0, B, Z, eps_slot=slot)
        ops.linear_fwd(z0, W, bias, h0, "relu", M=B)
        n1 = of.vae_reparam_fwd(ml, ring.view(-1), z1, p1, B, Z, W, bias, h1, "relu", eps_slot=slot)
        torch.cuda.synchronize()
        assert n0 == n1 == n_part
        assert torch.equal(z0, z1) and torch.equal(p0, p1)
        mu, lv = ml[:, :Z], ml[:, Z:]
        assert torch.equal(z1, mu + ring[step].view(B, Z) * torch.exp(lv / 2))
        if B < 1024:
            assert torch.equal(h0, h1)
        else:
            ref = torch.relu(z1.double() @ W.double().t() + bias.double())
            assert (h1.double() - ref).abs().max().item() <= 2 * (h0.double() - ref).abs().max().item() + 1e-6


@pytest.mark.parametrize("B,Z,H", [(512, 20, 400), (336, 20, 400), (100, 20, 400), (37, 8, 52), (64, 32, 128),
                                    (17, 4, 20)])
def test_vae_backward_mid_chain_in_one_launch(B, Z, H):
    """gm_vae_bwd_mid == gm_linear_bwd_dx_reparam (dz, d loss / d [mu | log_var]) followed by gm_linear_bwd_dx
    (dHe = (dml W_ml) . [He > 0]): vae.py:93-113 backwards between the two wide layers.  The summation orders are the
    separate launches' (asserted: bit-identical dml and dHe), and both against fp64 autograd."""
    torch.manual_seed(B + Z)
    dHdec = torch.randn(B, H).to(DEV)
    Wd1 = (torch.randn(H, Z) / Z ** 0.5).to(DEV)
    Wml = (torch.randn(2 * Z, H) / H ** 0.5).to(DEV)
    ml = (torch.randn(B, 2 * Z) * 0.5).to(DEV)
    He = torch.relu(torch.randn(B, H)).to(DEV)
    ring = torch.randn(3, B, Z).to(DEV)
    slot = ops.slot(0, 0, 2, 3, B * Z)
    dz0, dml0, dHe0 = torch.empty(B, Z, device=DEV), torch.empty(B, 2 * Z, device=DEV), torch.empty(B, H, device=DEV)
    ops.linear_bwd_dx_reparam(dHdec, Wd1, dz0, ml, ring.view(-1), dml0, eps_slot=slot)
    ops.linear_bwd_dx(dml0, Wml, dHe0, below=He, epi="relu")
    dml1, dHe1 = torch.full((B, 2 * Z), 3.0, device=DEV), torch.full((B, H), -3.0, device=DEV)
    of.vae_bwd_mid(dHdec, Wd1, ml, ring.view(-1), dml1, Wml, He, dHe1, B, eps_slot=slot)
    torch.cuda.synchronize()
    assert torch.equal(dml0, dml1)
    assert torch.equal(dHe0, dHe1)
    d = lambda t: t.double().cpu()
    mu, lv, e = d(ml[:, :Z]), d(ml[:, Z:]), d(ring[2])
    dz = d(dHdec) @ d(Wd1)
    ref_dml = torch.cat([dz + mu, dz * e * torch.exp(lv / 2) / 2 + 0.5 * (torch.exp(lv) - 1)], 1)
    ref_dHe = (ref_dml @ d(Wml)) * (d(He) > 0)
    assert (d(dml1) - ref_dml).abs().max().item() <= 2e-5 * max(1.0, ref_dml.abs().max().item())
    assert (d(dHe1) - ref_dHe).abs().max().item() <= 2e-5 * max(1.0, ref_dHe.abs().max().item())
