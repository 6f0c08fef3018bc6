"""be_gan.py:109-258 on the GAN engine: the pieces of the step BEGAN replaces (gan_steps.py) and its device-side
controller / schedulers."""
import torch

from . import ops
from .engine import GANEngine


class BEGANEngine(GANEngine):
    """be_gan.py:109-258 as a hipGraph: autoencoder critic (784->400->784), per-row L1 losses, the
    proportional controller K and both ReduceLROnPlateau schedulers kept in device memory and
    advanced by a one-thread kernel at the end of every iteration (which also carries the tick) --
    the reference's four `.item()` syncs per step disappear."""

    def __init__(self, model, data, B, device, use_graph=True, world_size=1, rank=0,
                 process_group=None, force_dp=False):
        super().__init__("be", model, data, B, device, use_graph=use_graph, world_size=world_size,
                         rank=rank, process_group=process_group, force_dp=force_dp)
        Bl, I = self.Bl, self.I
        z = lambda *s, **k: torch.zeros(*s, device=device, **k)
        self.Yd, self.dY, self.rows = z(2 * Bl, I), z(2 * Bl, I), z(2 * Bl)
        self.st, self.dst, self.ist = z(8), z(8, dtype=torch.float64), z(2, dtype=torch.int64)

    # fusions that do not apply here
    def _tick_in_head(self):
        return False

    def _adam_in_epilogue(self, net):
        return False                      # Adam needs the device-side lr scale: separate launch

    def configure(self, n_iters, G_lr, D_lr, D_steps, GAMMA=0.5, LAMBDA=1e-3, K=0.0, patience=0,
                  **kw):
        resume = kw.get("resume")
        self.gamma, self.lam, self.patience = float(GAMMA), float(LAMBDA), int(patience)
        super().configure(n_iters, G_lr, D_lr, D_steps, resume=resume,
                          extra_config=dict(GAMMA=self.gamma, LAMBDA=self.lam, patience=self.patience))
        self.st.zero_()
        self.st[0] = float(K)
        self.st[4] = 1.0
        self.st[5] = 1.0
        self.dst.zero_()
        self.dst[0] = float("inf")                 # ReduceLROnPlateau.best (mode='min')
        self.dst[1], self.dst[2], self.dst[3], self.dst[4] = D_lr, G_lr, D_lr, G_lr
        self.ist.zero_()
        if resume is not None:
            # the controller K (be_gan.py:189-191) and both ReduceLROnPlateau schedulers (:133-136,
            # :194-195: best, bad-epoch counts, current lr scales) continue where the run stopped;
            # the K argument of this train() call is superseded by the saved K
            be = resume["began"]
            self.st.copy_(be["st"]); self.dst.copy_(be["dst"]); self.ist.copy_(be["ist"])

    def optim_state(self):
        st = super().optim_state()
        cpu = lambda t: t.detach().cpu().clone()
        st["began"] = {"st": cpu(self.st), "dst": cpu(self.dst), "ist": cpu(self.ist)}
        st["config"] = dict(self.run_config)
        return st

    def _D_rest(self, st, it, j):
        from . import ops_fused as of
        Bl, d = self.Bl, self.D_steps
        D1, D2 = self.D1, self.D2
        X2, Hd, Yd, dY, dHd = self.X2, self.Hd, self.Yd, self.dY, self.dHd
        ops.linear_fwd(X2, D1.W, D1.b, Hd, "relu", M=2 * Bl, stream=st)            # encoder
        ops.linear_fwd(Hd, D2.W, D2.b, Yd, "id", M=2 * Bl, stream=st)              # decoder
        of.l1_rows(Yd, X2, 2 * Bl, Bl, self.st, dY, self.rows, B_global=self.B, stream=st)   # K = st[0]
        of.began_dloss(self.rows, Bl, self.st, self.lossD, self._slot(it, d, j, 0, 1), B_global=self.B,
                       stream=st)
        if self._dp():
            # DX, DG are means over the GLOBAL batch (be_gan.py:189-195): the per-rank partial means
            # in st[1:3] are summed over ranks before the K controller / plateau schedulers read them
            self._exchange_scalars(st, self.st[1:], 2)
        ops.linear_bwd_dx(dY, D2.W, dHd, below=Hd, epi="relu", M=2 * Bl, stream=st)
        if self.pair_dw:
            # both weight gradients of the autoencoder critic as one launch (plain gradients: Adam needs the
            # device-side lr scale and stays a launch of its own)
            ops.linear_bwd_dw_adam_pair(dict(dA=dY, X=Hd, lin=D2, adam=None, M=2 * Bl),
                                        dict(dA=dHd, X=X2, lin=D1, adam=None, M=2 * Bl), stream=st)
        else:
            ops.linear_bwd_dw(dY, Hd, D2.gW, D2.gb, M=2 * Bl, stream=st)
            ops.linear_bwd_dw(dHd, X2, D1.gW, D1.gb, M=2 * Bl, stream=st)

    def _lr_scale(self, net):
        return self.st[4:5] if net == "D" else self.st[5:6]

    def _issue_D_post(self, st, it, j):
        if self._peer():
            return
        ops.adam(self.fD.flat, self.fD.grad, self.fD.m, self.fD.v, self.schedD,
                 self._slot(it, self.D_steps, j, 0, 1), lr_scale=self.st[4:5], stream=st)

    def _G_critic(self, st, it):
        from . import ops_fused as of
        Bl = self.Bl
        D1, D2 = self.D1, self.D2
        Hd, Yd, dY, dHd, Xg = self.Hd, self.Yd, self.dY, self.dHd, self.Xg2
        ops.linear_fwd(Xg, D1.W, D1.b, Hd, "relu", M=Bl, stream=st)
        ops.linear_fwd(Hd, D2.W, D2.b, Yd, "id", M=Bl, stream=st)
        of.l1_rows(Yd, Xg, Bl, Bl, None, dY, self.rows, B_global=self.B, stream=st)
        of.sum_finalize(self.rows, Bl, self.lossG, scale=self.inv_b,
                        out_slot=self._slot(it, 1, self.g_off, 0, 1), stream=st)
        ops.linear_bwd_dx(dY, D2.W, dHd, below=Hd, epi="relu", M=Bl, stream=st)
        # G(z) enters |D(G(z)) - G(z)| twice: through D and directly (-sign/B = -dY)
        ops.linear_bwd_dx(dHd, D1.W, self.dXg, below=Xg, epi="sigmoid", M=Bl, add=dY, add_scale=-1.0,
                          stream=st)

    def _issue_G_post(self, st, it):
        if self._peer():
            return
        ops.adam(self.fG.flat, self.fG.grad, self.fG.m, self.fG.v, self.schedG,
                 self._G_sched_slot(it), lr_scale=self.st[5:6], stream=st)

    def _issue_end(self, st, it):
        from . import ops_fused as of
        of.began_update(self.st, self.dst, self.ist, self.gamma, self.lam, self.patience,
                        self.ctr if self.use_graph else None, stream=st)

    def K_value(self):
        return float(self.st[0].item())
