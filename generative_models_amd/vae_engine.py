"""vae.py / ae.py / bir_vae.py `compute_batch` + train loop + evaluate on hipGraphs: VAEEngine and the two engines
that reuse its machinery (AEEngine: no sampling, no KL; BIRVAEEngine: numpy's generator, MMD instead of KL)."""
import numpy as np
import torch

from . import ops
from ._lib import GMError
from .engine import CHUNK, FlatParams, GANEngine, HostReplay, NumpyReplay, _Linear, _align4


class VAEEngine:
    """vae.py:144-167 (train loop) + :214-223 (evaluate) as hipGraphs: one graph per distinct
    batch size (full batches and the ragged last one, 50 000 mod 512 = 336), a device step counter
    selecting index rows / eps rows / Adam-schedule rows / loss slots."""

    def __init__(self, model, device, use_graph=True, world_size=1, rank=0, process_group=None,
                 force_dp=False):
        self.model, self.device, self.use_graph = model, device, use_graph
        enc, dec = model.encoder, model.decoder
        plist = [enc.linear.weight, enc.linear.bias,
                 (enc.mu.weight, enc.log_var.weight), (enc.mu.bias, enc.log_var.bias),
                 dec.linear.weight, dec.linear.bias, dec.recon.weight, dec.recon.bias]
        self._dp_init(plist, world_size, rank, process_group, force_dp)
        self.fp = FlatParams(plist, device, grad_alloc=self._grad_alloc)
        fp = self.fp
        self.E1, self.D1, self.D2 = _Linear(fp, enc.linear), _Linear(fp, dec.linear), \
            _Linear(fp, dec.recon)
        Z, H = enc.mu.weight.shape
        self.Z, self.H, self.I = Z, H, enc.linear.weight.shape[1]
        i_w = [i for i, p in enumerate(fp.params) if p is enc.mu.weight][0]
        i_b = [i for i, p in enumerate(fp.params) if p is enc.mu.bias][0]
        o_w, o_b = fp.offsets[i_w], fp.offsets[i_b]

        class _Packed:          # [mu ; log_var] as one 2Z x H layer
            W = fp.flat[o_w:o_w + 2 * Z * H].view(2 * Z, H)
            b = fp.flat[o_b:o_b + 2 * Z]
            gW = fp.grad[o_w:o_w + 2 * Z * H].view(2 * Z, H)
            gb = fp.grad[o_b:o_b + 2 * Z]
            mW, vW = fp.m[o_w:o_w + 2 * Z * H], fp.v[o_w:o_w + 2 * Z * H]
            mb, vb = fp.m[o_b:o_b + 2 * Z], fp.v[o_b:o_b + 2 * Z]
        self.ML = _Packed
        self._common_init(device)

    has_eps = True              # the VAE draws eps per batch (vae.py:104); the plain AE does not

    # ---- data parallel: every batch's rows are split over the ranks; the losses are
    # SUMS (vae.py:203,212), so the gradient all-reduce is a plain sum with no 1/N and the per-rank
    # loss slots add up to the reference's values -------------------------------------------------
    def _dp_init(self, plist, world, rank, pg, force_dp):
        import os
        self.world, self.rank, self.pg, self.force_dp = world, rank, pg, bool(force_dp)
        self.comm, self._grad_alloc = None, None
        if world > 1 or force_dp:
            if os.environ.get("GM_DP_COMM", "peer") != "peer":
                raise GMError("data-parallel VAE / AE exchange gradients with the in-graph peer "
                              "communicator (GM_DP_COMM=peer)")
            from . import dp
            n = 0
            for item in plist:
                for q in (item if isinstance(item, (tuple, list)) else (item,)):
                    n += q.numel()
                n = _align4(n)
            self.comm = dp.PeerComm(n, world, rank, pg)
            ok = self.comm.selfcheck(self.device)
            if world > 1:
                import torch.distributed as dist
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
                if dist.get_backend(pg) == "nccl":
                    flag = flag.to(self.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=pg)
                ok = bool(flag.item())
            if not ok:
                raise GMError("peer exchange self-check failed: data-parallel VAE / AE needs hipIpc peer "
                              "mappings between the ranks' GPUs")
            self._grad_alloc = lambda k: self.comm.grad_buffer()[:k]

    def _dp(self):
        return self.world > 1 or self.force_dp

    def _rows(self, b):
        """Rows [lo, hi) of a batch of b rows owned by this rank (ragged batches split as evenly as
        integer division allows)."""
        return b * self.rank // self.world, b * (self.rank + 1) // self.world

    def read_losses(self, buf, lo, n):
        """Loss slots [lo, lo+n) summed over ranks (each rank holds its rows' partial sums)."""
        t = buf[lo:lo + n]
        if self.world > 1:
            import torch.distributed as dist
            from . import dp
            self.comm.check()
            t = t.cpu() if dist.get_backend(self.pg) != "nccl" else t.clone()
            dp.allreduce_sum_(t, self.pg)
        return t.cpu().numpy()

    def _common_init(self, device):
        from . import _respect_cpu_quota
        _respect_cpu_quota(force=False)             # once per process, when the first engine is built
        self.ctr = torch.zeros(1, dtype=torch.int64, device=device)
        self.graphs = {}
        self._bufB = None
        import os
        self.fuse_adam = True
        self.pair_dw = True
        self.graph_iters = max(1, int(os.environ.get("GM_GRAPH_ITERS", "32")))   # batches per graph
        self.prefetch_gather = True
        # launch fusions of round 3 (each replaces a ~5 us latency-bound launch by an epilogue)
        self.fuse_sqerr = True
        self.fuse_reparam_bwd = True
        self.fuse_reparam_fwd = True
        self.fuse_bwd_mid = True
        self.fin_in_dw = True
        self.fin_done = torch.zeros(1, dtype=torch.int32, device=device)

    def _alloc(self, B):
        if self._bufB == B:
            return
        dev, I, H, Z = self.device, self.I, self.H, self.Z
        z = lambda *s: torch.zeros(*s, device=dev)
        self.X, self.He, self.ml, self.Zs = z(B, I), z(B, H), z(B, 2 * Z), z(B, Z)
        self.Xb = (self.X, z(B, I))                 # batch i of a multi-batch graph reads Xb[i % 2]
        self.Hdec, self.Xr, self.dA = z(B, H), z(B, I), z(B, I)
        self.dHdec, self.dZ, self.dml, self.dHe = z(B, H), z(B, Z), z(B, 2 * Z), z(B, H)
        self.part = z(B)
        self._sq_alloc(B)
        self.part_kl = z((B * Z + 255) // 256)
        self._bufB = B
        self.graphs = {}

    def _slot(self, t, mul, add, ring, stride):
        if self.use_graph:
            return ops.slot(self.ctr.data_ptr(), mul, add, ring, stride)
        i = t * mul + add
        if ring > 0:
            i %= ring
        return ops.slot(0, 0, i, 0, stride)

    # -- reconstruction loss in the decoder's last forward (ops.linear_fwd_sqerr; GM_VAE_FUSE_SQERR=0: the
    # separate gm_sqerr_sigmoid_bwd launch) -- per (row, 32-column tile) partials, rows ldp floats apart
    def _sq_alloc(self, B):
        ldp = _align4((self.I + 31) // 32)
        self.part2 = torch.zeros(B, ldp, device=self.device)

    def _recon_fwd(self, st, x_in, lin, X, b):
        """x_hat = sigmoid(lin(x_in)), dA = d sum((X - x_hat)^2) / d (pre-sigmoid), and the loss partials.
        Returns (partials tensor, number of floats to sum)."""
        from . import ops_fused as of_
        if self.fuse_sqerr:
            ops.linear_fwd_sqerr(x_in, lin.W, lin.b, self.Xr, X, self.dA, self.part2, M=b, stream=st)
            return self.part2, b * self.part2.shape[1]
        ops.linear_fwd(x_in, lin.W, lin.b, self.Xr, "sigmoid", M=b, stream=st)
        of_.sqerr_sigmoid_bwd(X, self.Xr, self.dA, self.part, b, stream=st)
        return self.part, b

    def _gather_plan(self, pos, of):
        """Batch `pos` of a graph of `of` equal-size batches: (its image buffer, whether it gathers its
        own rows, the buffer the NEXT batch's rows are prefetched into or None).  Inside a multi-batch
        graph the gather of batch i+1 rides in a small forward GEMM of batch i (gm_linear_fwd_gather:
        extra workgroups of that launch), so only the graph's first batch pays a gather launch."""
        X = self.Xb[pos % 2]
        nxt = self.Xb[(pos + 1) % 2] if (self.prefetch_gather and pos + 1 < of) else None
        return X, (pos == 0 or not self.prefetch_gather), nxt

    def _fwd_with_prefetch(self, st, t, lo, b, x, lin, y, act, nxt):
        """linear_fwd, carrying the gather of the next batch's rows (ring slot t + 1) when asked to."""
        if nxt is None:
            ops.linear_fwd(x, lin.W, lin.b, y, act, M=b, stream=st)
        else:
            ops.linear_fwd_gather(x, lin.W, lin.b, y, act, self.data, self.idx_ring.view(-1)[lo:], nxt, M=b,
                                  B=b, idx_slot=self._slot(t, 1, 1, self.R, self.B), stream=st)

    def _issue(self, st, t, b, train, pos=0, of=1):
        """One batch of size b: forward + losses (+ backward + Adam when train)."""
        from . import ops_fused as of_
        of = max(1, of)
        R, B, Z = self.R, self.B, self.Z
        E1, ML, D1, D2 = self.E1, self.ML, self.D1, self.D2
        idx_slot = self._slot(t, 1, 0, R, B)
        eps_slot = self._slot(t, 1, 0, R, B * Z)
        loss_slot = self._slot(t, 1, 0, 0, 1)
        recon_out, kl_out = (self.recon, self.kl) if train else (self.vrecon, self.vkl)
        lo, hi = self._rows(b)                       # this rank's rows of the batch
        b = hi - lo
        X, own, nxt = self._gather_plan(pos, of)
        if own:
            ops.gather_rows(self.data, self.idx_ring.view(-1)[lo:], X, B=b, idx_slot=idx_slot, stream=st)
        ops.linear_fwd(X, E1.W, E1.b, self.He, "relu", M=b, stream=st)
        self._fwd_with_prefetch(st, t, lo, b, self.He, ML, self.ml, "id", nxt)
        eps_base = self.eps_ring.view(-1)[lo * Z:]
        if self.fuse_reparam_fwd and Z <= 32 and Z % 4 == 0:
            # reparameterisation + the decoder's first layer: ONE launch (the GEMM workgroups form z from
            # (mu, log_var, eps) themselves; bit-identical to the two launches)
            n_kl = of_.vae_reparam_fwd(self.ml, eps_base, self.Zs, self.part_kl, b, Z, D1.W, D1.b, self.Hdec, "relu",
                                       eps_slot=eps_slot, stream=st)
        else:
            n_kl = of_.vae_reparam_wide(self.ml, eps_base, self.Zs, self.part_kl, b, Z, eps_slot=eps_slot, stream=st)
            ops.linear_fwd(self.Zs, D1.W, D1.b, self.Hdec, "relu", M=b, stream=st)
        part, n_part = self._recon_fwd(st, self.Hdec, D2, X, b)
        if train:
            sched_slot = self._slot(t, 1, 0, 0, 1)
            if self.fuse_adam and not self._dp():
                # Adam (weight_decay 1e-5, vae.py:139-142) folded into every dW epilogue; each dX
                # GEMM reads a layer's weights BEFORE that layer's dW launch updates them
                adam = dict(sched=self.sched, sched_slot=sched_slot)
                dw = lambda dA, X, lin: ops.linear_bwd_dw_adam(dA, X, lin, adam, M=b,
                                                               weight_decay=self.wd, stream=st)
            else:
                adam = None
                dw = lambda dA, X, lin: ops.linear_bwd_dw(dA, X, lin.gW, lin.gb, M=b, stream=st)
            if self.pair_dw:
                # weight gradients of two layers as ONE launch once both their inputs exist (the dX
                # GEMMs that still read those weights are issued first)
                dw2 = lambda a1, a2: ops.linear_bwd_dw_adam_pair(
                    dict(dA=a1[0], X=a1[1], lin=a1[2], adam=adam, M=b),
                    dict(dA=a2[0], X=a2[1], lin=a2[2], adam=adam, M=b),
                    weight_decay=self.wd if adam is not None else 0.0, stream=st)
            else:
                dw2 = lambda a1, a2: (dw(*a1), dw(*a2))
            ops.linear_bwd_dx(self.dA, D2.W, self.dHdec, below=self.Hdec, epi="relu", M=b, stream=st)
            # gm_vae_bwd_mid keeps ONE hidden width (decoder's = encoder's) of at most 512 in its workgroup; other
            # models (hidden_dim 800 / 1024, unequal widths) take the two generic dX launches below
            mid = (self.fuse_bwd_mid and self.fuse_reparam_bwd and Z <= 32 and self.H % 4 == 0 and self.H <= 512
                   and D1.W.shape[0] == self.H and ML.W.shape[1] == self.H)
            if mid:
                # dz, d loss / d [mu | log_var] and dHe: the two narrow GEMMs between the decoder's and the encoder's
                # wide ones as ONE launch, 16 rows per workgroup (reads D1.W and ML.W before the pairs step them)
                of_.vae_bwd_mid(self.dHdec, D1.W, self.ml, eps_base, self.dml, ML.W, self.He, self.dHe, b,
                                eps_slot=eps_slot, stream=st)
            elif self.fuse_reparam_bwd:
                # dz and, in the same launch's epilogue, d loss / d [mu | log_var] (vae.py:100-106,210-212)
                ops.linear_bwd_dx_reparam(self.dHdec, D1.W, self.dZ, self.ml, eps_base, self.dml, M=b,
                                          eps_slot=eps_slot, stream=st)
            else:
                ops.linear_bwd_dx(self.dHdec, D1.W, self.dZ, M=b, stream=st)
            dw2((self.dA, self.Hdec, D2), (self.dHdec, self.Zs, D1))
            if not self.fuse_reparam_bwd:
                of_.vae_reparam_bwd(self.ml, eps_base, self.dZ, self.dml, b, Z,
                                   eps_slot=eps_slot, stream=st)
            if not mid:
                ops.linear_bwd_dx(self.dml, ML.W, self.dHe, below=self.He, epi="relu", M=b, stream=st)
            if self.fin_in_dw and self.pair_dw and adam is not None and not self._dp():
                # the batch's LAST launch: the encoder's two weight gradients + Adam, both loss sums (vae.py:203, :212)
                # in one more workgroup of the same grid, the counter tick by the last workgroup to finish
                ops.linear_bwd_dw_adam_pair_finalize(
                    dict(dA=self.dHe, X=X, lin=E1, adam=adam, M=b), dict(dA=self.dml, X=self.He, lin=ML, adam=adam, M=b),
                    dict(pa=part, na=n_part, out_a=recon_out, slot_a=loss_slot, pb=self.part_kl, nb=n_kl, out_b=kl_out,
                         slot_b=loss_slot, done=self.fin_done, tick=self.ctr if self.use_graph else None),
                    weight_decay=self.wd, stream=st)
                return
            dw2((self.dHe, X, E1), (self.dml, self.He, ML))    # (the big GEMM first: its tile shape serves both)
            self._optimizer_step(st, sched_slot)
        # both loss sums (vae.py:203, :212) are the step's LAST launch, which also carries the counter tick
        of_.sum_finalize2(part, n_part, recon_out, loss_slot, self.part_kl, n_kl, kl_out, loss_slot,
                          tick=self.ctr if self.use_graph else None, stream=st)

    def _optimizer_step(self, st, sched_slot):
        """optimizer.step() (vae.py:162) when it is not fused into the dW epilogues: data parallel ->
        gradient SUM over ranks + Adam in the exchange's gather kernel."""
        if self._dp():
            self.comm.allreduce_adam(self.fp.grad, self.fp.flat, self.fp.m, self.fp.v, self.sched, sched_slot,
                                     weight_decay=self.wd, stream=st)
        elif not self.fuse_adam:
            ops.adam(self.fp.flat, self.fp.grad, self.fp.m, self.fp.v, self.sched, sched_slot,
                     weight_decay=self.wd, stream=st)

    def configure(self, B, n_train_steps, lr, weight_decay, resume=None):
        dev = self.device
        self._alloc(B)
        self.B, self.wd = B, float(weight_decay)
        self.fp.rebind()
        self.fp.reset_state()
        self.fp.grad.zero_()
        self.step0 = 0
        self.run_config = {"B": int(B), "lr": float(lr), "weight_decay": float(weight_decay)}
        if resume is not None:                       # see GANEngine.configure
            saved = resume.get("config")
            if saved is not None and not resume.get("lenient", False):
                diff = {k: (saved[k], self.run_config[k]) for k in self.run_config
                        if k in saved and saved[k] != self.run_config[k]}
                if diff:
                    raise GMError("checkpoint was written by a run with different settings (saved, now): "
                                  "%s; load_checkpoint(path, strict=False) overrides" % diff)
            if resume["m"].numel() != self.fp.m.numel():
                raise GMError("checkpoint optimizer state does not match this model")
            self.fp.m.copy_(resume["m"]); self.fp.v.copy_(resume["v"])
            self.step0 = int(resume["step"])
        self.steps_planned = n_train_steps
        # buffers whose addresses the captured graphs hold are kept across train() calls (grow-only),
        # so a second train() with the same batch size / weight decay replays instead of re-capturing
        self._moved = False
        self.sched = GANEngine._pbuf(self, "sched", ops.adam_schedule(lr, max(1, n_train_steps),
                                                                      start=self.step0 + 1))
        self.recon = GANEngine._pbuf(self, "recon", max(1, n_train_steps))
        self.kl = GANEngine._pbuf(self, "kl", max(1, n_train_steps))
        self.R = CHUNK
        if getattr(self, "_ring_B", None) != B:
            self.idx_ring = torch.zeros(self.R, B, dtype=torch.int64, device=dev)
            self.eps_ring = torch.zeros(self.R, B, self.Z, device=dev)
            self.stage = [dict(idx=torch.zeros(self.R, B, dtype=torch.int64).pin_memory(),
                               eps=torch.zeros(self.R, B, self.Z).pin_memory(), event=None)
                          for _ in range(2)]
            self._ring_B = B
            self._moved = True
        vkey = (B, self.wd, self.use_graph, self.fuse_adam, self.pair_dw, self.prefetch_gather)
        if self._moved or getattr(self, "_vkey", None) != vkey:
            self.graphs = {}
        self._vkey = vkey
        self.t_train = 0

    def optim_state(self):
        torch.cuda.synchronize()
        return {"m": self.fp.m.detach().cpu().clone(), "v": self.fp.v.detach().cpu().clone(),
                "step": self.step0 + self.steps_planned, "config": dict(self.run_config)}

    def _graph(self, b, train, k=1):
        """hipGraph of k consecutive batches of size b (the device counter advances per batch)."""
        key = (b, train, self.data.data_ptr(), k)
        if key not in self.graphs:
            torch.cuda.synchronize()
            # full batches: every power-of-two size at once (an epoch's chunking asks for different
            # sizes from pass to pass; capturing them one by one would land inside later passes)
            sizes = [k]
            if b == self.B and self.use_graph:
                sizes, n = [], 1
                while n <= self.graph_iters:
                    sizes.append(n)
                    n *= 2
                if k not in sizes:
                    sizes.append(k)
            for n in sizes:
                kk = (b, train, self.data.data_ptr(), n)
                if kk not in self.graphs:
                    self.graphs[kk] = ops.Graph().capture(
                        lambda st, n=n: [self._issue(st, 0, b, train, pos=i, of=n) for i in range(n)])
        return self.graphs[key]

    def run_pass(self, data, perm, train, t0):
        """One pass over `data` in the order `perm` (host int64 tensor): batches of B rows, last
        one ragged.  Global eps draws (vae.py:104) happen here, batch by batch, in order.
        t0: first loss/schedule slot.  Returns number of batches."""
        self.data = data
        B, R, Z = self.B, self.R, self.Z
        n = perm.numel()
        nb = (n + B - 1) // B
        self.ctr.fill_(t0)
        done, which = 0, 0
        while done < nb:
            t = t0 + done
            cnt = min(R - (t % R), nb - done)
            s = self.stage[which]
            which ^= 1
            if s["event"] is not None:
                s["event"].synchronize()
            sizes = [min(B, n - (done + k) * B) for k in range(cnt)]
            lo = done * B
            hi = min(n, lo + cnt * B)
            s["idx"].view(-1)[:hi - lo].copy_(perm[lo:hi])    # rows of B indices, last one ragged
            self._draw_chunk(s, sizes)
            r = t % R
            self.idx_ring[r:r + cnt].copy_(s["idx"][:cnt], non_blocking=True)
            self._upload_chunk(s, r, cnt)
            ev = torch.cuda.Event()
            ev.record()
            s["event"] = ev
            k = 0
            while k < len(sizes):
                b = sizes[k]
                if not self.use_graph:
                    self._issue(ops.stream_ptr(), t + k, b, train)
                    k += 1
                    continue
                run = 1
                while k + run < len(sizes) and sizes[k + run] == b:
                    run += 1
                # the run of equal-size batches as power-of-two graphs, largest first (each size is
                # captured once, on first use): only a graph's FIRST batch pays its own gather launch
                piece = 1
                while piece * 2 <= min(run, self.graph_iters):
                    piece *= 2
                self._graph(b, train, piece).launch()
                k += piece
            done += cnt
        return nb

    def _torch_normal_rows(self, dst, sizes):
        """torch.randn(b, Z) per batch of the chunk (global CPU generator, in order) into dst[k]: the
        full batches in one C call (HostReplay), a ragged last batch on its own."""
        B, Z = self.B, self.Z
        nfull = sum(1 for b in sizes if b == B)
        from ._lib import DRAW_NORMAL
        replay = HostReplay.available() and B * Z >= 16
        if replay and nfull:
            replay = HostReplay.run([HostReplay.op(DRAW_NORMAL, B * Z, dst, B * Z * 4)], nfull)
        for k, b in enumerate(sizes):
            if replay and b == B:
                continue
            if replay and b * Z >= 16 and HostReplay.run([HostReplay.op(DRAW_NORMAL, b * Z, dst[k], 0)], 1):
                continue
            dst[k].view(-1)[:b * Z].normal_()

    def _draw_chunk(self, s, sizes):
        """HOST draws of the chunk's batches, in the reference's order (vae.py:104)."""
        if self.has_eps:
            self._torch_normal_rows(s["eps"], sizes)

    def _upload_chunk(self, s, r, cnt):
        if self.has_eps:
            self.eps_ring[r:r + cnt].copy_(s["eps"][:cnt], non_blocking=True)

    def alloc_val(self, n):
        if getattr(self, "vrecon", None) is None or self.vrecon.numel() < n:
            self.vrecon = torch.zeros(n, device=self.device)
            self.vkl = torch.zeros(n, device=self.device)
            self.graphs = {k: g for k, g in self.graphs.items() if k[1]}   # drop eval graphs


class AEEngine(VAEEngine):
    """ae.py:104-164 on the VAE engine's machinery: encoder layer (relu),
    decoder layer (sigmoid), squared-error loss; no sampling, so no eps ring and no KL term."""

    has_eps = False

    def __init__(self, model, device, use_graph=True, world_size=1, rank=0, process_group=None,
                 force_dp=False):
        self.model, self.device, self.use_graph = model, device, use_graph
        enc, dec = model.encoder, model.decoder
        plist = [enc.linear.weight, enc.linear.bias, dec.linear.weight, dec.linear.bias]
        self._dp_init(plist, world_size, rank, process_group, force_dp)
        self.fp = FlatParams(plist, device, grad_alloc=self._grad_alloc)
        self.E1, self.D2 = _Linear(self.fp, enc.linear), _Linear(self.fp, dec.linear)
        self.H, self.I = enc.linear.weight.shape
        self.Z = 1                                   # dummy width of the (unused) eps ring
        self._common_init(device)

    def _alloc(self, B):
        if self._bufB == B:
            return
        z = lambda *s: torch.zeros(*s, device=self.device)
        self.X, self.He, self.Xr, self.dA = z(B, self.I), z(B, self.H), z(B, self.I), z(B, self.I)
        self.Xb = (self.X, z(B, self.I))
        self.dHe, self.part = z(B, self.H), z(B)
        self._sq_alloc(B)
        self._bufB = B
        self.graphs = {}

    def _issue(self, st, t, b, train, pos=0, of=1):
        """One batch of size b: ae.py:147-160 (+ backward and Adam when train)."""
        from . import ops_fused as of_
        E1, D2 = self.E1, self.D2
        idx_slot = self._slot(t, 1, 0, self.R, self.B)
        loss_slot = self._slot(t, 1, 0, 0, 1)
        lo, hi = self._rows(b)                       # this rank's rows of the batch
        b = hi - lo
        X, own, nxt = self._gather_plan(pos, of)
        if own:
            ops.gather_rows(self.data, self.idx_ring.view(-1)[lo:], X, B=b, idx_slot=idx_slot, stream=st)
        self._fwd_with_prefetch(st, t, lo, b, X, E1, self.He, "relu", nxt)
        part, n_part = self._recon_fwd(st, self.He, D2, X, b)
        if train:
            sched_slot = self._slot(t, 1, 0, 0, 1)
            adam = dict(sched=self.sched, sched_slot=sched_slot) if (self.fuse_adam and not self._dp()) else None
            # dH reads the decoder weights before the paired dW launch updates them
            ops.linear_bwd_dx(self.dA, D2.W, self.dHe, below=self.He, epi="relu", M=b, stream=st)
            ops.linear_bwd_dw_adam_pair(dict(dA=self.dA, X=self.He, lin=D2, adam=adam, M=b),
                                        dict(dA=self.dHe, X=X, lin=E1, adam=adam, M=b),
                                        weight_decay=self.wd if adam is not None else 0.0, stream=st)
            self._optimizer_step(st, sched_slot)
        of_.sum_finalize(part, n_part, self.recon if train else self.vrecon, out_slot=loss_slot,
                        tick=self.ctr if self.use_graph else None, stream=st)


class BIRVAEEngine(VAEEngine):
    """bir_vae.py:119-232: encoder 784->400->mu, z = mu + eps with eps ~
    N(0, set_var) from NUMPY's global RNG (drawn on the host exactly as the reference does,
    bir_vae.py:92-94 -- a variance used as a standard deviation is part of the contract), decoder,
    loss = sum (x - x_hat)^2 + 1000 * MMD(z) with the Gaussian-kernel MMD against
    x = torch.randn(z.shape) (:203, global torch CPU generator).  `kl` / `vkl` hold the MMD terms."""

    LAMBDA = 1000.0

    def __init__(self, model, device, use_graph=True, world_size=1, rank=0, process_group=None,
                 force_dp=False):
        if world_size > 1 or force_dp:
            raise GMError("BIR-VAE's MMD couples every pair of rows of the batch: it does not shard on "
                          "the batch axis (run it on one GPU)")
        self.model, self.device, self.use_graph = model, device, use_graph
        enc, dec = model.encoder, model.decoder
        plist = [enc.linear.weight, enc.linear.bias, enc.mu.weight, enc.mu.bias,
                 dec.linear.weight, dec.linear.bias, dec.recon.weight, dec.recon.bias]
        self._dp_init(plist, 1, 0, None, False)
        self.fp = FlatParams(plist, device)
        fp = self.fp
        self.E1, self.MU = _Linear(fp, enc.linear), _Linear(fp, enc.mu)
        self.D1, self.D2 = _Linear(fp, dec.linear), _Linear(fp, dec.recon)
        self.Z, self.H = enc.mu.weight.shape
        self.I = enc.linear.weight.shape[1]
        self.set_var = float(model.set_var)
        self._common_init(device)

    def _alloc(self, B):
        if self._bufB == B:
            return
        dev, I, H, Z = self.device, self.I, self.H, self.Z
        z = lambda *s: torch.zeros(*s, device=dev)
        self.X, self.He, self.Mu, self.Zs = z(B, I), z(B, H), z(B, Z), z(B, Z)
        self.Xb = (self.X, z(B, I))
        self.Hdec, self.Xr, self.dA = z(B, H), z(B, I), z(B, I)
        self.dHdec, self.dZ, self.dZm, self.dHe = z(B, H), z(B, Z), z(B, Z), z(B, H)
        self.part, self.partm = z(B), z(B)
        self._sq_alloc(B)
        self._bufB = B
        self.graphs = {}

    def configure(self, B, n_train_steps, lr, weight_decay, resume=None):
        super().configure(B, n_train_steps, lr, weight_decay, resume=resume)
        if getattr(self, "_prior_B", None) != B or "prior" not in self.stage[0]:
            self.prior_ring = torch.zeros(self.R, B, self.Z, device=self.device)
            for s in self.stage:
                s["prior"] = torch.zeros(self.R, B, self.Z).pin_memory()
            self._prior_B = B
            self.graphs = {}

    def _draw_chunk(self, s, sizes):
        import numpy as np
        Z = self.Z
        # model(images) -> reparameterize: np.random.normal(0, set_var, mu.shape).float(), batch after batch on
        # numpy's global generator: replayed in C for the whole chunk (NumpyReplay), else through numpy itself
        if not (NumpyReplay.available() and NumpyReplay.fill(self.set_var, s["eps"], self.B, Z, sizes)):
            for k, b in enumerate(sizes):
                e = np.random.normal(loc=0.0, scale=self.set_var, size=(b, Z))
                s["eps"][k].view(-1)[:b * Z].copy_(torch.from_numpy(e).float().view(-1))
        # maximum_mean_discrepancy: torch.randn(z.shape) -- a different generator, order-independent
        self._torch_normal_rows(s["prior"], sizes)

    def _upload_chunk(self, s, r, cnt):
        self.eps_ring[r:r + cnt].copy_(s["eps"][:cnt], non_blocking=True)
        self.prior_ring[r:r + cnt].copy_(s["prior"][:cnt], non_blocking=True)

    def _issue(self, st, t, b, train, pos=0, of=1):
        from . import ops_fused as of_
        R, B, Z = self.R, self.B, self.Z
        E1, MU, D1, D2 = self.E1, self.MU, self.D1, self.D2
        idx_slot = self._slot(t, 1, 0, R, B)
        eps_slot = self._slot(t, 1, 0, R, B * Z)
        loss_slot = self._slot(t, 1, 0, 0, 1)
        recon_out, mmd_out = (self.recon, self.kl) if train else (self.vrecon, self.vkl)
        X, own, nxt = self._gather_plan(pos, of)
        if own:
            ops.gather_rows(self.data, self.idx_ring.view(-1), X, B=b, idx_slot=idx_slot, stream=st)
        ops.linear_fwd(X, E1.W, E1.b, self.He, "relu", M=b, stream=st)
        self._fwd_with_prefetch(st, t, 0, b, self.He, MU, self.Mu, "id", nxt)
        of_.bir_reparam(self.Mu, self.eps_ring.view(-1), self.Zs, b, Z, eps_slot=eps_slot, stream=st)
        ops.linear_fwd(self.Zs, D1.W, D1.b, self.Hdec, "relu", M=b, stream=st)
        part, n_part = self._recon_fwd(st, self.Hdec, D2, X, b)
        of_.bir_mmd(self.Zs, self.prior_ring.view(-1), self.partm, self.dZm if train else None, b, Z,
                   self.LAMBDA, prior_slot=eps_slot, stream=st)
        if train:
            sched_slot = self._slot(t, 1, 0, 0, 1)
            adam = dict(sched=self.sched, sched_slot=sched_slot) if self.fuse_adam else None
            dw2 = lambda a1, a2: ops.linear_bwd_dw_adam_pair(
                dict(dA=a1[0], X=a1[1], lin=a1[2], adam=adam, M=b),
                dict(dA=a2[0], X=a2[1], lin=a2[2], adam=adam, M=b),
                weight_decay=self.wd if adam is not None else 0.0, stream=st)
            # every dX reads a layer's weights BEFORE that layer's dW(+Adam) launch updates them
            ops.linear_bwd_dx(self.dA, D2.W, self.dHdec, below=self.Hdec, epi="relu", M=b, stream=st)
            # d loss / d z = decoder path + d(1000 * mmd)/dz ; z = mu + eps -> d/d mu is the same
            ops.linear_bwd_dx(self.dHdec, D1.W, self.dZ, M=b, add=self.dZm, add_scale=1.0, stream=st)
            dw2((self.dA, self.Hdec, D2), (self.dHdec, self.Zs, D1))
            ops.linear_bwd_dx(self.dZ, MU.W, self.dHe, below=self.He, epi="relu", M=b, stream=st)
            dw2((self.dZ, self.He, MU), (self.dHe, X, E1))
            self._optimizer_step(st, sched_slot)
        # reconstruction sum and 1000 * MMD in the step's last launch, which also carries the tick
        of_.sum_finalize2(part, n_part, recon_out, loss_slot, self.partm, b, mmd_out, loss_slot,
                          scale_b=self.LAMBDA, tick=self.ctr if self.use_graph else None, stream=st)
