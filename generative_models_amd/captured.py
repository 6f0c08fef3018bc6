"""The README extension contract at speed (/root/reference/README.md:29-65; loop: ns_gan.py:122-156).

The reference advertises ONE way of adding a model: subclass a Trainer and override `train_D` / `train_G`.  Such a
subclass cannot run on the fused engine (its loss is arbitrary PyTorch code), but everything AROUND the two hooks is
still the stock loop -- and that is where the reference's time goes (2.6 ms of DataLoader
reshuffle per step, a CPU `randn` + H2D per `compute_noise`, two `.item()` syncs, unfused Adam).  When only
`train_D` / `train_G` are overridden (stock `process_batch`, `compute_noise`, networks and loader), this module keeps
the device data path of the fused engine and replays the user's two hooks as ONE captured graph per D+G iteration:

  * the host replays the reference's global-CPU-generator protocol ahead of the device (engine.HostReplay: sampler
    seeds + randperm prefix, `randn(B, Z)` per critic step, `randn(B, Z)` for the generator step) into pinned rings,
    uploaded half a ring at a time;
  * inside the graph the batch is gathered from the resident (bit-packed) dataset (`gm_gather_rows[_bits]` with the
    ring slot resolved from a device counter), the stock `compute_noise` the hooks call returns a tensor that
    `gm_copy_slot_f32` fills from the noise ring, the hooks' forward runs on the HIP GEMM autograd Functions
    (ops.fused_linear), `backward()` on their backward kernels, Adam is `gm_adam` on one flat buffer per network with
    its bias-correction row resolved from the same counter, and the loss lands in a per-step device buffer
    (`gm_copy_slot_f32`) that is read back ONCE per epoch -- no `.item()` per step;
  * work the reference computes and throws away is skipped: G's parameters do not require grad while `train_D` runs,
    D's not while `train_G` runs (ns_gan.py:130,149 zero those gradients before anyone reads them).

The first iteration runs the same body EAGERLY inside a transaction (parameters, optimizer state and the generator
state are snapshotted): if the hooks consume the CPU generator themselves, call `compute_noise` other than once per
hook with (B, z_dim), or leave a parameter without a gradient, the snapshot is restored and the caller falls back to the
fully general loop -- results are the reference's either way.  If the hooks cannot be captured (a `.item()`, a host
branch on device data), the iterations keep running eagerly on the device data path.
"""
import numpy as np
import torch

from . import ops
from ._lib import GMError, DRAW_NORMAL, DRAW_SAMPLER, DrawOp

VARIANTS = ("ns", "mm", "w", "ls", "ra", "f")      # stock draws per step: sampler + randn(B, Z); randn(B, Z)
RING = 64                                          # iterations of draws per ring (two halves)


class NotCapturable(Exception):
    """The user's hooks do not fit the captured loop; nothing has been changed -- take the general path."""


class CapturedLoop:
    def __init__(self, trainer, data, B, device):
        from .engine import FlatParams, HostReplay
        if not HostReplay.available():
            raise NotCapturable("host replay unavailable")
        self.tr, self.data, self.B, self.dev = trainer, data, B, device
        m = trainer.model
        self.Z, self.I, self.N = m.z_dim, data.shape[1], data.shape[0]
        self.fD, self.fG = FlatParams(m.D.parameters(), device), FlatParams(m.G.parameters(), device)
        self.graph = None
        self.mode = "eager"

    # ---- per-train() state ------------------------------------------------------------------------------------
    def configure(self, n_iters, G_lr, D_lr, D_steps, clip, kwD, kwG):
        d, B, Z, dev = D_steps, self.B, self.Z, self.dev
        self.d, self.n_iters, self.clip, self.kwD, self.kwG = d, n_iters, clip, kwD, kwG
        self.fD.rebind(); self.fG.rebind()
        self.fD.reset_state(); self.fG.reset_state()          # optimizers are locals of train() (ns_gan.py:107-110)
        self.schedD = torch.from_numpy(ops.adam_schedule(D_lr, n_iters * d)).to(dev)
        self.schedG = torch.from_numpy(ops.adam_schedule(G_lr, n_iters)).to(dev)
        self.lossD = torch.zeros(n_iters * d, device=dev)
        self.lossG = torch.zeros(n_iters, device=dev)
        self.ctr = torch.zeros(1, dtype=torch.int64, device=dev)
        R = RING
        self.h_idx = torch.empty(R * d, B, dtype=torch.int64).pin_memory()
        self.h_z = torch.empty(R * (d + 1), B * Z).pin_memory()
        self.d_idx = torch.empty(R * d, B, dtype=torch.int64, device=dev)
        self.d_z = torch.empty(R * (d + 1), B * Z, device=dev)
        self.images = torch.empty(B, self.I, device=dev)
        self.zbuf = [torch.empty(B, Z, device=dev) for _ in range(d + 1)]
        self.ev = [None, None]
        self.graph = None
        self.mode = "eager"
        self.done = 0
        self._programs = {}

    def _program(self, r):
        """gm_draw_op list of ONE iteration (reference order: ns_gan.py:128-133 per critic step, :151 for the generator
        step) writing into the host rings from ring slot r on."""
        p = self._programs.get(r)
        if p is None:
            from .engine import HostReplay
            d, B, Z = self.d, self.B, self.Z
            prog = []
            for j in range(d):
                prog.append(HostReplay.op(DRAW_SAMPLER, B, self.h_idx[r * d + j], d * B * 8, a=self.N))
                prog.append(HostReplay.op(DRAW_NORMAL, B * Z, self.h_z[r * (d + 1) + j], (d + 1) * B * Z * 4))
            prog.append(HostReplay.op(DRAW_NORMAL, B * Z, self.h_z[r * (d + 1) + d], (d + 1) * B * Z * 4))
            p = self._programs[r] = (DrawOp * len(prog))(*prog)
        return p

    def _fill(self, it0, n):
        """Draws of iterations [it0, it0 + n) (inside one ring half) -> pinned rings -> device rings."""
        from .engine import HostReplay
        R, d = RING, self.d
        r, half = it0 % R, (it0 % R) // (R // 2)
        assert r + n <= (half + 1) * (R // 2)
        if self.ev[half] is not None:
            self.ev[half].synchronize()                         # the upload that last read this half has finished
        if not HostReplay.run(self._program(r), n):
            raise GMError("host replay refused the captured loop's draw program")
        self.d_idx[r * d:(r + n) * d].copy_(self.h_idx[r * d:(r + n) * d], non_blocking=True)
        self.d_z[r * (d + 1):(r + n) * (d + 1)].copy_(self.h_z[r * (d + 1):(r + n) * (d + 1)], non_blocking=True)
        self.ev[half] = torch.cuda.Event()
        self.ev[half].record()

    # ---- one D+G iteration (ns_gan.py:122-156), eagerly or under capture ------------------------------------------
    def _slot(self, mul, add, ring, stride):
        return ops.slot(self.ctr.data_ptr(), mul, add, ring, stride)

    def _adam(self, fp, sched, slot, clamp):
        grads = [p.grad for p in fp.params]
        if any(g is None for g in grads):
            raise NotCapturable("a parameter was left without a gradient")
        torch._foreach_copy_(fp.gviews, grads)
        ops.adam(fp.flat, fp.grad, fp.m, fp.v, sched, slot, clamp=clamp)

    def _iteration(self):
        tr, d, B, Z, R = self.tr, self.d, self.B, self.Z, RING
        m = tr.model
        Gp, Dp = list(m.G.parameters()), list(m.D.parameters())
        self.calls = []
        for j in range(d):
            ops.gather_rows(self.data, self.d_idx, self.images, B, idx_slot=self._slot(d, j, R * d, B))
            ops.copy_slot(self.d_z, self.zbuf[j], B * Z, src_slot=self._slot(d + 1, j, R * (d + 1), B * Z))
            self._next_noise = self.zbuf[j]
            for p in Dp:
                p.grad = None
            for p in Gp:
                p.requires_grad_(False)                        # ns_gan.py:149 zeroes G's gradients unread
            try:
                loss = tr.train_D(self.images, **self.kwD)
            finally:
                for p in Gp:
                    p.requires_grad_(True)
            if isinstance(loss, tuple):
                loss = loss[0]
            loss.backward()
            self._adam(self.fD, self.schedD, self._slot(d, j, 0, 1), self.clip)
            ops.copy_slot(loss.detach().reshape(1), self.lossD, 1, dst_slot=self._slot(d, j, 0, 1))
        ops.copy_slot(self.d_z, self.zbuf[d], B * Z, src_slot=self._slot(d + 1, d, R * (d + 1), B * Z))
        self._next_noise = self.zbuf[d]
        for p in Gp:
            p.grad = None
        for p in Dp:
            p.requires_grad_(False)                            # ns_gan.py:130 zeroes D's gradients unread
        try:
            loss = tr.train_G(self.images, **self.kwG)
        finally:
            for p in Dp:
                p.requires_grad_(True)
        loss.backward()
        self._adam(self.fG, self.schedG, self._slot(1, 0, 0, 1), 0.0)
        ops.copy_slot(loss.detach().reshape(1), self.lossG, 1, dst_slot=self._slot(1, 0, 0, 1))
        ops.tick(self.ctr)
        if len(self.calls) != d + 1 or any(c != (B, Z) for c in self.calls):
            raise NotCapturable("compute_noise was called %s; the stock loop draws (B, z_dim) once per hook" % self.calls)

    def _noise_hook(self, batch_size, z_dim):
        """Stands in for the STOCK compute_noise (ns_gan.py:218-220) while the loop runs: the tensor the prefetched
        draw of this hook call was copied into."""
        self.calls.append((int(batch_size), int(z_dim)))
        return self._next_noise

    def _with_hook(self, fn):
        tr = self.tr
        tr.__dict__["compute_noise"] = self._noise_hook         # instance attribute shadows the (stock) method
        try:
            return fn()
        finally:
            tr.__dict__.pop("compute_noise", None)

    # ---- transaction around the first iteration ---------------------------------------------------------------
    def _first_iteration(self):
        snap = dict(rng=torch.get_rng_state(), D=self.fD.flat.clone(), G=self.fG.flat.clone())
        try:
            self._fill(0, 1)
            rng_after_draws = torch.get_rng_state()
            self._with_hook(self._iteration)
            torch.cuda.synchronize()
            if not torch.equal(rng_after_draws, torch.get_rng_state()):
                raise NotCapturable("the hooks draw from the global CPU generator themselves")
        except BaseException:
            torch.cuda.synchronize()
            torch.set_rng_state(snap["rng"])
            self.fD.flat.copy_(snap["D"]); self.fG.flat.copy_(snap["G"])
            self.fD.reset_state(); self.fG.reset_state()
            raise
        self.done = 1

    def _capture(self):
        g = torch.cuda.CUDAGraph()
        try:
            torch.cuda.synchronize()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._iteration()
        except Exception as e:                                   # noqa: BLE001  (a sync / host read inside the hooks)
            self.capture_error = "%s: %s" % (type(e).__name__, str(e)[:200])
            torch.cuda.synchronize()
            return None
        return g

    # ---- public ------------------------------------------------------------------------------------------------
    def run(self, n, use_graph=True):
        """Iterations [done, done + n) of this train() call."""
        if self.done == 0:
            self._first_iteration()
            n -= 1
            if use_graph and n > 0:
                self.graph = self._with_hook(self._capture)
                self.mode = "graph" if self.graph is not None else "eager"
        half = RING // 2
        while n > 0:
            it = self.done
            k = min(n, half - it % half)
            self._fill(it, k)
            if self.graph is not None:
                for _ in range(k):
                    self.graph.replay()
            else:
                self._with_hook(lambda: [self._iteration() for _ in range(k)])
            self.done += k
            n -= k

    def losses(self, it0, it1):
        d = self.d
        G = self.lossG[it0:it1].cpu().numpy().astype(np.float64)             # one sync per epoch
        D = self.lossD[it0 * d:it1 * d].cpu().numpy().astype(np.float64).reshape(-1, d)
        return [float(x) for x in G], [float(np.mean(r)) for r in D]
