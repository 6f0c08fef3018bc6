"""The launch sequences of one D_steps x train_D + train_G iteration (ns_gan.py:117-160 and siblings) as mix-ins of
engine.GANEngine: the engine's core (rings, host draws, graphs, run()) does not know what a step is made of.

  CriticStep      batch gather, G(zD), then one of three critic structures -- folded head (separable losses; RaGAN /
                  Fisher on one GPU), fused head (penalty variants, many-row launches, data parallel), unfused
  InfoQStep       InfoGAN's train_Q + MI optimizer (info_gan.py:269-304)
  GeneratorStep   G(zG), D(G(zG)) in generator mode, the generator's two weight gradients
  PenaltySteps    WGAN-GP / DRAGAN: forward pieces and the hand-derived second backward

Every method reads the engine's buffers and switches through `self`; BEGAN overrides pieces of these
(began_engine.BEGANEngine)."""
import torch

from . import ops


class CriticStep:
    # ---- pieces of one critic step (composed sequentially, or as parallel graph branches) -----
    def _gather_rides(self):
        """The batch gather rides in the grid of the generator's first forward launch."""
        return self.ride_gather

    def _packed_operand(self):
        """The critic step reads its real rows as BITS: the gather copies the selected rows of the
        1-bit resident dataset as words (100 B per MNIST row instead of 3136 B of fp32) and the folded step's two
        launches -- hidden layer forward, layer-1 weight gradient -- expand them in registers
        (gm_linear_fwd_headpart_bits, gm_linear_bwd_dw_adam_head_fold_bits).  Bit-identical losses and parameters
        (tests/test_gpu_trainers.py); GM_PACKED_OPERAND=1 turns it on -- measured neither faster nor slower than the
        fp32 rows on one MI355X (profiles/r05_experiments.md section 9), so the default stays the path every other
        variant shares.  Needs the folded step of a separable loss and a batch of whole 32-row tiles."""
        import os
        if os.environ.get("GM_PACKED_OPERAND", "0") != "1":
            return False
        return isinstance(self.data, ops.PackedData) and self._fold_head() and \
            self.variant not in ("ra", "fisher") and self.Bl % 32 == 0 and self.I % 4 == 0

    def _xbits(self):
        """(words, words per row, rows) of the packed real rows, or None."""
        if not self._packed_operand():
            return None
        if getattr(self, "Xbits", None) is None:
            self.Xbits = torch.zeros(self.Bl, self.data.wpr, dtype=torch.int32, device=self.device)
        return (self.Xbits, self.data.wpr, self.Bl)

    def _gather_args(self, it, j):
        Bl, d, R = self.Bl, self.D_steps, self.R
        r0 = self.ring_r0                         # this rank's rows of the (device) index ring
        xb = self._xbits()
        return dict(data=self.data, idx=self.idx_ring.view(-1)[r0:], out=self.X2 if xb is None else xb[0], B=Bl,
                    idx_slot=self._slot(it, d, j, R * d, self.ring_B))

    def _D_gather(self, st, it, j):
        if self._gather_rides():
            return                                  # done by _D_gen's first launch
        ops.gather_rows(stream=st, **self._gather_args(it, j))

    def _batch_gen(self):
        """Both generator forwards of an iteration (critic step's G(zD), generator step's G(zG))
        read the same G parameters: with D_steps == 1 they run as ONE launch pair on 2B rows (the
        noise ring stores [zD; zG] back to back -- on a data-parallel rank that needs the rank-local
        device rings, where the rank's zD rows and zG rows of an iteration are adjacent)."""
        return self.batch_gen_env and self.D_steps == 1 and \
            (self.world == 1 or self._local_rings())

    def _local_rings(self):
        """Data parallel: device rings hold only this rank's rows of every draw (see _alloc_rings)."""
        import os
        return self.world > 1 and self.variant != "dra"

    def _D_gen(self, st, it, j):
        Bl, d, R = self.Bl, self.D_steps, self.R
        G1, G2 = self.G1, self.G2
        zD_slot = self._slot(it, d, j, R * d, self.zD_stride)
        zbase = self.zD_base[self.ring_r0 * self.Z:].view(-1, self.Z)
        rows = 2 * Bl if (self._batch_gen() and not self._standalone_G) else Bl
        if self._gather_rides():
            ops.linear_fwd_gather(zbase, G1.W, G1.b, self.HG, "relu", M=rows, x_slot=zD_slot,
                                  stream=st, **self._gather_args(it, j))
        else:
            ops.linear_fwd(zbase, G1.W, G1.b, self.HG, "relu", M=rows, x_slot=zD_slot, stream=st)
        if self._interp_in_gen():
            # WGAN-GP: x_hat = eps*x + (1-eps)*G(zD) written by this launch's epilogue for its first
            # Bl rows (the real rows were gathered by the previous launch's rider)
            r0 = self.ring_r0
            ops.linear_fwd_interp(self.HG, G2.W, G2.b, self.XX[Bl:], "sigmoid", self.eps_ring.view(-1)[r0:],
                                  self._slot(it, d, j, R * d, self.ring_B), self.XX[:Bl], self.Xh, Bl, M=rows,
                                  stream=st)
        else:
            ops.linear_fwd(self.HG, G2.W, G2.b, self.XX[Bl:], "sigmoid", M=rows, stream=st)

    def _interp_in_gen(self):
        import os
        # (the DAG experiment runs the gather on a side stream, concurrently with this launch)
        return self.variant == "wgp"

    # ---- the critic step behind the generator's forward: three builders, one per launch structure -------------------
    #   folded     separable losses (+ RaGAN / Fisher on one GPU): hidden layer forward with the head's partial dots,
    #              then ONE launch for the layer-1 weight gradient + head backward + both Adam steps
    #   fused head penalty variants (WGAN-GP, DRAGAN) and anything the fold does not take (many-row launches, data
    #              parallel): head_fwd_loss + a grouped / stacked weight-gradient launch
    #   unfused    N = 1 GEMV + loss kernel (+ the scalar exchanges of RaGAN / Fisher under data parallelism) + separate
    #              gradient launches
    def _D_rest(self, st, it, j):
        if self._fold_head():
            return self._critic_folded(st, it, j)
        if self.fuse_head and self.variant not in ("ra", "fisher"):
            return self._critic_fused_head(st, it, j)
        return self._critic_unfused(st, it, j)

    def _critic_folded(self, st, it, j):
        Bl, d = self.Bl, self.D_steps
        D1, D2 = self.D1, self.D2
        X2, Hd, S2, dS = self.X2, self.Hd, self.S2, self.dS
        loss_slot = self._slot(it, d, j, 0, 1)
        # 2 launches instead of 3: hidden layer forward (+ partial dots of the head), then the
        # layer-1 weight gradient (+Adam) with the head's backward workgroups riding -- scores,
        # row losses and dS are rebuilt from the partial dots in that launch's prologue
        xb = self._xbits()                                # real rows as bits: X2[:Bl] is never written or read
        ops.linear_fwd_headpart(X2, D1.W, D1.b, Hd, "relu", D2, self.fold, M=2 * Bl, stream=st, xbits=xb)
        adam = self._adam_args("D", self._slot(it, d, j, 0, 1)) if self._adam_in_epilogue("D") else None
        fa = self.fold.args(self.loss_key, self.out_act, self.hyper, S=S2, dS=dS, rowloss=self.rowloss,
                            pen=self.aux if self.variant == "fisher" else None)   # Fisher: lambda lives in aux
        head = dict(H=Hd, lin=D2, loss_out=self.lossD, loss_slot=loss_slot, inv_b=self.inv_b, B=Bl,
                    adam=adam)
        ops.linear_bwd_dw_adam_head_fold(Hd, X2, D1, adam, head, fa, M=2 * Bl, stream=st, xbits=xb)
        if self.variant == "fisher":
            from . import ops_fused as of
            of.fisher_commit(self.aux, stream=st)        # lambda <- its successor (fisher_gan.py:155-156)

    def _critic_forward(self, st, it, j):
        """D's hidden layer on [x ; G(z)] (WGAN-GP / DRAGAN: on [x_hat ; x ; G(z)] as one 3B-row launch) and the
        penalty's forward pieces; returns (aux, hyper) of the loss."""
        Bl = self.Bl
        D1 = self.D1
        merged = self.variant in ("wgp", "dra") and self.merge_fwd3
        if merged:
            if self.variant == "wgp":
                self._gp_prepare(st, it, j)
            else:
                self._dra_prepare(st, it, j)
            ops.linear_fwd(self.XX4[:3 * Bl], D1.W, D1.b, self.HH3, "relu", M=3 * Bl, stream=st)
        else:
            ops.linear_fwd(self.X2, D1.W, D1.b, self.Hd, "relu", M=2 * Bl, stream=st)
        aux, hyper = (self.aux if self.variant == "fisher" else None), self.hyper
        if self.variant == "wgp":
            self._issue_gp_forward(st, it, j, fwd_done=merged)
            aux, hyper = self.pen, (0.0,) * 7 + (self.gp_lambda,)
        if self.variant == "dra":
            self._issue_dra_forward(st, it, j, fwd_done=merged)
            aux, hyper = self.pen, tuple(self.hyper) + (0.0,) * (7 - len(self.hyper)) + (self.gp_lambda,)
        return aux, hyper

    def _critic_dw1(self, st, it, j):
        """Layer-1 weight gradient on its own (+ Adam in its epilogue on one GPU), then what is left of the penalty."""
        Bl, d = self.Bl, self.D_steps
        if self._adam_in_epilogue("D"):
            ops.linear_bwd_dw_adam(self.dHd, self.X2, self.D1, self._adam_args("D", self._slot(it, d, j, 0, 1)),
                                   M=2 * Bl, stream=st)
        else:
            ops.linear_bwd_dw(self.dHd, self.X2, self.D1.gW, self.D1.gb, M=2 * Bl, stream=st)

    def _critic_penalty_backward(self, st):
        if self.variant == "wgp" and not self._wgp_stacked():
            self._issue_gp_backward(st)
        if self.variant == "dra" and not self._dra_stacked():
            self._issue_dra_backward(st)

    def _critic_fused_head(self, st, it, j):
        from . import ops_fused as of
        Bl, d = self.Bl, self.D_steps
        D1, D2 = self.D1, self.D2
        X2, Hd, S2, dS, dHd = self.X2, self.Hd, self.S2, self.dS, self.dHd
        loss_slot = self._slot(it, d, j, 0, 1)
        aux, hyper = self._critic_forward(st, it, j)
        of.head_fwd_loss(self.loss_key, False, Hd, D2.W, D2.b, self.out_act, Bl, hyper,
                         self.inv_b, aux, S2, dS, self.rowloss, dH=dHd, stream=st)
        adam = self._adam_args("D", self._slot(it, d, j, 0, 1)) if self._adam_in_epilogue("D") else None
        if not self.group_head:
            of.head_bwd(Hd, dS, D2.W, self.rowloss, None, D2.gW, D2.gb, self.lossD, loss_slot,
                        self.inv_b, False, Bl, lin=D2, adam=adam, stream=st)
            self._critic_dw1(st, it, j)
            self._critic_penalty_backward(st)
            return
        # head backward + first-layer weight gradient (+ both Adam steps when they are
        # fused: one GPU, nothing accumulates into these gradients later): ONE launch
        head = dict(H=Hd, dS=dS, lin=D2, rowloss=self.rowloss, loss_out=self.lossD,
                    loss_slot=loss_slot, inv_b=self.inv_b, B=Bl, adam=adam)
        if self._wgp_stacked():
            # dW1 = [u ; dH]^T [gamma ; x ; G(z)] over 3B rows (the penalty rows do not reach
            # db1), gw2 += the penalty's share (computed by _issue_gp_forward)
            if self.pen_in_head:
                head["pen"] = dict(s=self.Sh, h=self.Hh, t=self.T)    # summed by the head workgroups
            else:
                head["gw2_add"] = self.gw2_pen
            ops.linear_bwd_dw_adam_head(self.DU, self.XX4, D1, adam, head, M=3 * Bl, ones_from=Bl,
                                        stream=st)
        elif self._dra_stacked():
            # the penalty's second backward first (it reads W1 and w2, which this launch steps): t = dv W1^T,
            # then dA1 and the sigma'' path's share of (gw2, gb2) on their own; then ONE launch:
            # dW1 over 4B rows (+ db1 from the last 3B), the head's backward with both shares added, Adam x 2
            self._issue_dra_backward(st, stacked=True)
            head["gw2_add"], head["gb2_add"] = self.gw2_pen, self.gb2_pen
            ops.linear_bwd_dw_adam_head(self.DU4, self.XX5, D1, adam, head, M=4 * Bl, ones_from=Bl,
                                        stream=st)
        else:
            ops.linear_bwd_dw_adam_head(dHd, X2, D1, adam, head, M=2 * Bl, stream=st)
        self._critic_penalty_backward(st)

    def _critic_unfused(self, st, it, j):
        Bl, d = self.Bl, self.D_steps
        D2 = self.D2
        Hd, S2, dS, dHd = self.Hd, self.S2, self.dS, self.dHd
        loss_slot = self._slot(it, d, j, 0, 1)
        aux, hyper = self._critic_forward(st, it, j)
        ops.linear_fwd(Hd, D2.W, D2.b, S2.view(-1, 1), self.out_act, M=2 * Bl, stream=st)
        loss = lambda **kw: ops.gan_loss(
            self.loss_key, False, S2[:Bl], S2[Bl:], Bl, self.out_act, self.lossD, dS[:Bl], dS[Bl:],
            hyper=hyper, inv_b=self.inv_b, loss_slot=loss_slot, aux=aux, db=D2.gb, stream=st, **kw)
        if self._dp() and self.variant == "ra":
            # mean(D(G(z))) and sum(du) span the GLOBAL batch (ra_gan.py:204)
            loss(phase=1, pre=self.pre)
            self._exchange_scalars(st, self.pre, 1)
            loss(phase=2, pre=self.pre)
            self._exchange_scalars(st, self.pre[1:], 1)
            loss(phase=3, pre=self.pre)
        elif self._dp() and self.variant == "fisher":
            # the four moments span the GLOBAL batch; lambda's ascent is then identical on every
            # rank (fisher_gan.py:214-223,155-156); rank 0 reports the (global) loss
            loss(phase=1, pre=self.pre)
            self._exchange_scalars(st, self.pre, 4)
            loss(phase=2, pre=self.pre, loss_scale=1.0 if self.rank == 0 else 0.0)
        else:
            loss()
        ops.linear_bwd_dw(dS.view(-1, 1), Hd, D2.gW, None, M=2 * Bl, stream=st)
        ops.linear_bwd_dx(dS.view(-1, 1), D2.W, dHd, below=Hd, epi="relu", M=2 * Bl, stream=st)
        self._critic_dw1(st, it, j)
        self._critic_penalty_backward(st)

    def _issue_D_pre(self, st, it, j):
        self._D_gather(st, it, j)
        self._D_gen(st, it, j)
        self._D_rest(st, it, j)

    def _adam_args(self, net, sched_slot):
        sched = {"D": self.schedD, "G": self.schedG}.get(net)
        if net == "MI":
            sched = self.schedMI
        return dict(sched=sched, sched_slot=sched_slot, clamp=self.clip if net == "D" else 0.0)

    def _issue_D_post(self, st, it, j):
        if self._adam_in_epilogue("D") or self._peer():
            return                  # already applied by the gradient epilogues / the all-gather kernel
        ops.adam(self.fD.flat, self.fD.grad, self.fD.m, self.fD.v, self.schedD,
                 self._slot(it, self.D_steps, j, 0, 1), clamp=self.clip, stream=st)


class InfoQStep:
    # ---- InfoGAN train_Q (info_gan.py:269-304) + MI_optimizer.step: runs after the generator
    # step, i.e. after the folded tick -> every slot is addressed with post=True ---------------
    def _issue_Q(self, st, it):
        from . import ops_fused as of
        Bl, R = self.Bl, self.R
        G1, G2, Q1, Q2 = self.G1, self.G2, self.Q1, self.Q2
        Hg, Xg = self.Hg2, self.Xg2                   # free again: the generator step is done
        zbase = self.zQ_ring.view(-1)[self.ring_r0 * self.Z:].view(-1, self.Z)
        z_slot = self._slot(it, 1, 0, R, self.ring_B * self.Z, post=True)
        s_slot = self._slot(it, 1, 0, 0, 1, post=True)
        ops.linear_fwd(zbase, G1.W, G1.b, Hg, "relu", M=Bl, x_slot=z_slot, stream=st)
        ops.linear_fwd(Hg, G2.W, G2.b, Xg, "sigmoid", M=Bl, stream=st)
        ops.linear_fwd(Xg, Q1.W, Q1.b, self.Hq, "relu", M=Bl, stream=st)
        ops.linear_fwd(self.Hq, Q2.W, Q2.b, self.Qo, "id", M=Bl, stream=st)
        of.info_q_loss(self.Qo, zbase, z_slot, Bl, self.zd, self.nd, self.nc, self.dQo, self.lossMI,
                       s_slot, B_global=self.B, stream=st)
        fused = self.fuse_adam and self._single()
        if fused:
            adam = self._adam_args("MI", s_slot)
            dw = lambda dA, X, lin, **kw: ops.linear_bwd_dw_adam(dA, X, lin, adam, M=Bl, stream=st, **kw)
            g1, g2 = self.G1mi, self.G2mi
        else:
            dw = lambda dA, X, lin, **kw: ops.linear_bwd_dw(dA, X, lin.gW, lin.gb, M=Bl, stream=st, **kw)
            g1, g2 = G1, G2
        # every dX reads a layer's weights before that layer's dW(+Adam) launch
        ops.linear_bwd_dx(self.dQo, Q2.W, self.dHq, below=self.Hq, epi="relu", M=Bl, stream=st)
        ops.linear_bwd_dx(self.dHq, Q1.W, self.dXg, below=Xg, epi="sigmoid", M=Bl, stream=st)
        ops.linear_bwd_dx(self.dXg, G2.W, self.dHg, below=Hg, epi="relu", M=Bl, stream=st)
        if self.pair_dw and fused:
            # the four weight gradients as two paired launches (round 4; the big GEMM first: its tile serves both)
            ops.linear_bwd_dw_adam_pair(dict(dA=self.dHq, X=Xg, lin=Q1, adam=adam, M=Bl),
                                        dict(dA=self.dQo, X=self.Hq, lin=Q2, adam=adam, M=Bl), stream=st)
            ops.linear_bwd_dw_adam_pair(dict(dA=self.dXg, X=Hg, lin=g2, adam=adam, M=Bl),
                                        dict(dA=self.dHg, X=zbase, lin=g1, adam=adam, M=Bl, x_slot=z_slot), stream=st)
        else:
            dw(self.dQo, self.Hq, Q2)
            dw(self.dHq, Xg, Q1)
            dw(self.dXg, Hg, g2)
            dw(self.dHg, zbase, g1, x_slot=z_slot)
        if self._peer():
            # MI_optimizer.step (info_gan.py:148,207): G's gradient bucket with the MI optimizer's OWN
            # moments, and Q's bucket -- all-reduce + Adam in the gather kernels
            self._comms["G"].allreduce_adam(self.fG.grad, self.fG.flat, self.mi_m, self.mi_v, self.schedMI,
                                            s_slot, stream=st)
            self._comms["Q"].allreduce_adam(self.fQ.grad, self.fQ.flat, self.fQ.m, self.fQ.v, self.schedMI,
                                            s_slot, stream=st)
        elif not fused:
            ops.adam(self.fG.flat, self.fG.grad, self.mi_m, self.mi_v, self.schedMI, s_slot, stream=st)
            ops.adam(self.fQ.flat, self.fQ.grad, self.fQ.m, self.fQ.v, self.schedMI, s_slot, stream=st)


class GeneratorStep:
    # ---- pieces of the generator step (own Hg2/Xg2 buffers: its generator forward only needs G's
    # parameters, so it can run as a parallel branch of the critic step) ----------------------
    def _G_zslot(self, it):
        zbase = self.zG_base[self.ring_r0 * self.Z:].view(-1, self.Z)
        return zbase, self._slot(it, 1, 0, self.R, self.zG_stride)

    def _G_gen(self, st, it):
        if self._batch_gen() and not self._standalone_G:
            return                                  # done together with the critic step's G(z)
        G1, G2 = self.G1, self.G2
        zbase, zG_slot = self._G_zslot(it)
        ops.linear_fwd(zbase, G1.W, G1.b, self.Hg2, "relu", M=self.Bl, x_slot=zG_slot, stream=st)
        ops.linear_fwd(self.Hg2, G2.W, G2.b, self.Xg2, "sigmoid", M=self.Bl, stream=st)

    def _G_critic(self, st, it):
        """D(G(z)) forward, loss, and the backward through D down to d loss / d (pre-sigmoid G)."""
        Bl = self.Bl
        D1, D2 = self.D1, self.D2
        Hd, S2, dS, dHd, Xg = self.Hd, self.S2, self.dS, self.dHd, self.Xg2
        loss_slot = self._slot(it, 1, self.g_off, 0, 1)
        if self._fold_head_G():
            tick = self.ctr if self._tick_in_head() else None
            ops.linear_fwd_headpart(Xg, D1.W, D1.b, Hd, "relu", D2, self.fold, M=Bl, stream=st)
            fa = self.fold.args(self.loss_key, self.out_act, self.hyper, S=S2, dS=dS, rowloss=self.rowloss)
            ops.linear_bwd_dx_head_fold(
                Hd, D1.W, self.dXg, dict(H=Hd, lin=D2, loss_out=self.lossG, loss_slot=loss_slot,
                                         inv_b=self.inv_b, B=Bl, gen_mode=True, tick=tick),
                fa, below=Xg, epi="sigmoid", M=Bl, stream=st)
            return
        ops.linear_fwd(Xg, D1.W, D1.b, Hd, "relu", M=Bl, stream=st)
        if self.fuse_head:
            from . import ops_fused as of
            tick = self.ctr if self._tick_in_head() else None
            of.head_fwd_loss(self.loss_key, True, Hd, D2.W, D2.b, self.out_act, Bl, self.hyper,
                             self.inv_b, None, S2, dS, self.rowloss, dH=dHd, stream=st)
            if self.ride_head_dx:
                # the head's one scalar workgroup (loss + tick) rides in the dX launch
                ops.linear_bwd_dx_head(
                    dHd, D1.W, self.dXg,
                    dict(H=Hd, dS=dS, lin=D2, rowloss=self.rowloss, loss_out=self.lossG,
                         loss_slot=loss_slot, inv_b=self.inv_b, B=Bl, gen_mode=True, tick=tick),
                    below=Xg, epi="sigmoid", M=Bl, stream=st)
                return
            of.head_bwd(Hd, dS, D2.W, self.rowloss, None, None, None, self.lossG, loss_slot,
                        self.inv_b, True, Bl, tick=tick, stream=st)
        else:
            ops.linear_fwd(Hd, D2.W, D2.b, S2.view(-1, 1), self.out_act, M=Bl, stream=st)
            ops.gan_loss(self.loss_key, True, None, S2, Bl, self.out_act, self.lossG, None, dS,
                         hyper=self.hyper, inv_b=self.inv_b, loss_slot=loss_slot, stream=st)
            ops.linear_bwd_dx(dS.view(-1, 1), D2.W, dHd, below=Hd, epi="relu", M=Bl, stream=st)
        ops.linear_bwd_dx(dHd, D1.W, self.dXg, below=Xg, epi="sigmoid", M=Bl, stream=st)

    # everything below runs AFTER the generator step's head kernel, i.e. after the folded tick
    def _G_sched_slot(self, it):
        return self._slot(it, 1, self.g_off, 0, 1, post=True)

    def _G_dh(self, st, it):
        ops.linear_bwd_dx(self.dXg, self.G2.W, self.dHg, below=self.Hg2, epi="relu", M=self.Bl,
                          stream=st)

    def _G_dw2(self, st, it):
        if self._adam_in_epilogue("G"):             # updates G2.W: must come after _G_dh read it
            ops.linear_bwd_dw_adam(self.dXg, self.Hg2, self.G2,
                                   self._adam_args("G", self._G_sched_slot(it)), M=self.Bl, stream=st)
        else:
            ops.linear_bwd_dw(self.dXg, self.Hg2, self.G2.gW, self.G2.gb, M=self.Bl, stream=st)

    def _G_dw1(self, st, it):
        zbase = self.zG_base[self.ring_r0 * self.Z:].view(-1, self.Z)
        zG_slot = self._slot(it, 1, 0, self.R, self.zG_stride, post=True)
        if self._adam_in_epilogue("G"):
            ops.linear_bwd_dw_adam(self.dHg, zbase, self.G1,
                                   self._adam_args("G", self._G_sched_slot(it)), M=self.Bl,
                                   x_slot=zG_slot, stream=st)
        else:
            ops.linear_bwd_dw(self.dHg, zbase, self.G1.gW, self.G1.gb, M=self.Bl, x_slot=zG_slot,
                              stream=st)

    def _G_dh_dw1(self, st, it):
        self._G_dh(st, it)
        self._G_dw1(st, it)

    def _issue_G_pre(self, st, it):
        self._G_gen(st, it)
        self._G_critic(st, it)
        self._G_dh(st, it)                          # reads G2.W before _G_dw2 may update it
        if self.pair_dw:
            # both weight gradients of the generator (+ their Adam steps on one GPU): ONE launch
            adam = self._adam_args("G", self._G_sched_slot(it)) if self._adam_in_epilogue("G") else None
            zbase = self.zG_base[self.ring_r0 * self.Z:].view(-1, self.Z)
            zG_slot = self._slot(it, 1, 0, self.R, self.zG_stride, post=True)
            ops.linear_bwd_dw_adam_pair(
                dict(dA=self.dXg, X=self.Hg2, lin=self.G2, adam=adam, M=self.Bl),
                dict(dA=self.dHg, X=zbase, lin=self.G1, adam=adam, M=self.Bl, x_slot=zG_slot),
                stream=st)
            return
        self._G_dw2(st, it)
        self._G_dw1(st, it)

    def _issue_G_post(self, st, it):
        if self._adam_in_epilogue("G") or self._peer():
            return
        ops.adam(self.fG.flat, self.fG.grad, self.fG.m, self.fG.v, self.schedG,
                 self._G_sched_slot(it), stream=st)


class PenaltySteps:
    # -- WGAN-GP penalty: w_gp_gan.py:195-218, hand-derived second backward -----
    def _gp_prepare(self, st, it, j):
        """x_hat, unless the generator's last launch already wrote it."""
        from . import ops_fused as ops_gp
        Bl, d, R = self.Bl, self.D_steps, self.R
        if not self._interp_in_gen():
            eps_slot = self._slot(it, d, j, R * d, self.ring_B)
            ops_gp.interp(self.eps_ring.view(-1)[self.ring_r0:], eps_slot, self.X2[:Bl], self.X2[Bl:], self.Xh,
                          stream=st)

    def _issue_gp_forward(self, st, it, j, fwd_done=False):
        from . import ops_fused as ops_gp
        Bl = self.Bl
        D1, D2 = self.D1, self.D2
        if not fwd_done:
            self._gp_prepare(st, it, j)
            ops.linear_fwd(self.Xh, D1.W, D1.b, self.Hh, "relu", M=Bl, stream=st)
        if self._wgp_stacked():
            ops_gp.head_gp(self.Hh, D2.W, D2.b, self.Sh, self.U, stream=st)     # D(x_hat), u: one launch
        else:
            ops.linear_fwd(self.Hh, D2.W, D2.b, self.Sh.view(-1, 1), "relu", M=Bl, stream=st)
            ops_gp.gp_u(self.Sh, self.Hh, D2.W, self.U, stream=st)              # u = m2*(m1.w2)
        ops.linear_bwd_dx(self.U, D1.W, self.Gr, M=Bl, stream=st)                # g = u W1
        ops_gp.gp_norm(self.Gr, self.Gam, self.pen, self.gp_lambda, self.inv_b, stream=st)
        if self._wgp_stacked():
            # second backward, w2's share, BEFORE the stacked dW1 launch steps W1:
            # t = gamma W1^T, gw2_pen = sum_b m2 m1 . t   (w_gp_gan.py:215)
            ops.linear_fwd(self.Gam, D1.W, None, self.T, "id", M=Bl, stream=st)
            if not self.pen_in_head:
                ops_gp.gp_dw2_store(self.Sh, self.Hh, self.T, self.gw2_pen, stream=st)

    # -- DRAGAN penalty: dra_gan.py:198-223; sigmoid critic => second-order terms --
    def _dra_prepare(self, st, it, j):
        """x_hat = x + (1 - delta) * C * std(x) * U  (dra_gan.py:200-205)."""
        from . import ops_fused as of
        Bl, d, R = self.Bl, self.D_steps, self.R
        x = self.X2[:Bl]
        if self._dp():
            # images.data.std() is over the GLOBAL batch (dra_gan.py:204): (sum x, sum x^2) of my rows,
            # summed over ranks, then the unbiased std of B*I elements
            of.std_sums(x, Bl, self.pre[8:], ws=self.std_ws, stream=st)
            self._exchange_scalars(st, self.pre[8:], 2)
            of.std_from_sums(self.pre[8:], self.B * self.I, self.stdv, stream=st)
        else:
            of.std_all(x, Bl, self.stdv, ws=self.std_ws, stream=st)       # images.data.std()
        r0 = self.ring_r0                                                          # my rows of the draws
        of.dragan_xhat(x, self.delta_ring.view(-1)[r0:], self._slot(it, d, j, R * d, self.ring_B),
                       self.U_ring.view(-1)[r0 * self.I:], self._slot(it, d, j, R * d, self.ring_B * self.I),
                       self.stdv, self.Xh, Bl, stream=st)

    def _issue_dra_forward(self, st, it, j, fwd_done=False):
        from . import ops_fused as of
        Bl = self.Bl
        D1, D2 = self.D1, self.D2
        if not fwd_done:
            self._dra_prepare(st, it, j)
            ops.linear_fwd(self.Xh, D1.W, D1.b, self.Hh, "relu", M=Bl, stream=st)
        ops.linear_fwd(self.Hh, D2.W, D2.b, self.Sh.view(-1, 1), "sigmoid", M=Bl, stream=st)
        of.gp_u(self.Sh, self.Hh, D2.W, self.U, stream=st)                          # m1 . w2 (sigma > 0)
        ops.linear_bwd_dx(self.U, D1.W, self.Gr, M=Bl, stream=st)                   # v = (m1.w2) W1
        of.dragan_rows(self.Sh, self.Gr, self.Gam, self.da2, self.pen, self.gp_lambda, self.inv_b,
                       Bl, stream=st)                                               # Gam = dv

    def _issue_dra_backward(self, st, stacked=False):
        from . import ops_fused as of
        Bl = self.Bl
        D1, D2 = self.D1, self.D2
        if stacked:
            ops.linear_fwd(self.Gam, D1.W, None, self.T, "id", M=Bl, stream=st)      # T = dv W1^T
            of.dragan_head_bwd(self.Hh, self.T, self.da2, D2.W, self.gw2_pen, self.gb2_pen, self.dA1, Bl,
                               store=True, stream=st)
            return
        ops.linear_bwd_dw(self.U, self.Gam, D1.gW, None, M=Bl, accumulate=True, stream=st)
        ops.linear_fwd(self.Gam, D1.W, None, self.T, "id", M=Bl, stream=st)          # T = dv W1^T
        of.dragan_head_bwd(self.Hh, self.T, self.da2, D2.W, D2.gW, D2.gb, self.dA1, Bl, stream=st)
        ops.linear_bwd_dw(self.dA1, self.Xh, D1.gW, D1.gb, M=Bl, accumulate=True, stream=st)

    def _issue_gp_backward(self, st):
        from . import ops_fused as ops_gp
        Bl = self.Bl
        D1, D2 = self.D1, self.D2
        ops.linear_bwd_dw(self.U, self.Gam, D1.gW, None, M=Bl, accumulate=True, stream=st)
        ops.linear_fwd(self.Gam, D1.W, None, self.T, "id", M=Bl, stream=st)      # gamma W1^T
        ops_gp.gp_dw2(self.Sh, self.Hh, self.T, D2.gW, stream=st)
