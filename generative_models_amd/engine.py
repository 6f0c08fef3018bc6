"""Fused-step engines: the reference's Trainer.train inner loop (ns_gan.py:117-160 and siblings;
vae.py:144-167) re-designed for MI355X.

Reference per step: DataLoader reshuffle on the host, 3 H2D copies, ~60 tiny autograd ops,
two `.item()` syncs.  Here, per iteration:

  host   : replays the reference's global-CPU-generator draw order in C
           (csrc/gm_hostrng.cpp: mt19937 + ATen's normal_/uniform_/random_/randint restated bit
           for bit, O(B) randperm prefix for the sampler) on a native worker thread, straight into
           PINNED host rings that mirror the device rings; graphs are enqueued ahead of their draws;
  device : hipGraphs of 1..128 iterations (each: D_steps critic steps + 1 generator step): a one-wave wait
           on the fill gate and a stage-in kernel that pulls the iterations' ring slots over PCIe, then
           MFMA GEMMs with fused bias/activation/activation-gradient epilogues (gather, critic head,
           Adam riding in their launches).  A device counter advanced by the graph itself selects
           the ring slot / Adam-schedule row / loss slot, so replays need no host-side arguments;
  sync   : none until the epoch ends (losses are read back in one copy).

Only the wasted work of the reference is skipped (G gradients during the D
step, D gradients during the G step); every observable -- parameters, loss lists, RNG stream
position -- matches the reference.

This module: the shared pieces (FlatParams, host RNG replay) and the GAN engine's CORE; what a GAN iteration launches is
gan_steps.py (mix-ins), BEGAN's overrides began_engine.py, the VAE / AE / BIR-VAE engines vae_engine.py (all re-exported
here)."""
import numpy as np
import torch

from . import ops
from ._lib import GMError
from .gan_steps import CriticStep, GeneratorStep, InfoQStep, PenaltySteps

CHUNK = 64          # iterations prefetched per host->device upload (VAE / AE passes)
GAN_RING = 128      # iterations of draws the GAN engines' host / device rings hold


def _align4(n):
    return (n + 3) // 4 * 4


class FlatParams:
    """Packs nn.Parameters into one flat fp32 device buffer (16-byte aligned segments) and
    re-points `.data` at views of it, so one Adam launch / one all-reduce bucket covers the net.
    The nn.Linear modules stay the parameter holders (state_dict keys)."""

    def __init__(self, params, device, grad_alloc=None):
        # an element may be a tuple of parameters: packed back to back with no padding between
        # them (e.g. the VAE's mu / log_var heads, used as ONE [2Z, H] matrix by the kernels)
        # grad_alloc(n): where the flat gradient bucket lives (data parallel: inside the peer
        # exchange region, so that it is all-reduced in place)
        self.params, offs, n = [], [], 0
        for item in params:
            group = item if isinstance(item, (tuple, list)) else (item,)
            for p in group:
                self.params.append(p)
                offs.append(n)
                n += p.numel()
            n = _align4(n)
        self.n = n
        self.offsets = offs
        self.flat = torch.zeros(n, device=device)
        self.grad = torch.zeros(n, device=device) if grad_alloc is None else grad_alloc(n)
        self.m = torch.zeros(n, device=device)
        self.v = torch.zeros(n, device=device)
        self.views, self.gviews = [], []
        for p, o in zip(self.params, offs):
            v = self.flat[o:o + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
            self.views.append(v)
            self.gviews.append(self.grad[o:o + p.numel()].view(p.shape))

    def still_bound(self):
        return all(p.data.data_ptr() == v.data_ptr() for p, v in zip(self.params, self.views))

    def rebind(self):
        """If the user replaced parameter storage (load_state_dict keeps it; .to() may not)."""
        for p, v in zip(self.params, self.views):
            if p.data.data_ptr() != v.data_ptr():
                v.copy_(p.data)
                p.data = v

    def reset_state(self):
        self.m.zero_()
        self.v.zero_()

    def expose_grads(self):
        for p, g in zip(self.params, self.gviews):
            p.grad = g


class _Linear:
    """Raw views of one nn.Linear inside a FlatParams (weights, bias, and their grads)."""

    def __init__(self, fp, lin, m=None, v=None):
        m = fp.m if m is None else m            # alternative Adam moments (InfoGAN's MI optimizer)
        v = fp.v if v is None else v
        iw = [i for i, p in enumerate(fp.params) if p is lin.weight][0]
        ib = [i for i, p in enumerate(fp.params) if p is lin.bias][0]
        self.W, self.b = fp.views[iw], fp.views[ib]
        self.gW, self.gb = fp.gviews[iw], fp.gviews[ib]
        ow, ob, nw, nb = fp.offsets[iw], fp.offsets[ib], lin.weight.numel(), lin.bias.numel()
        self.mW, self.vW = m[ow:ow + nw], v[ow:ow + nw]              # Adam moments (flat views)
        self.mb, self.vb = m[ob:ob + nb], v[ob:ob + nb]


# --------------------------------------------------------------------------------------------
# Host RNG protocol (parity mode).  Every draw below comes from torch's GLOBAL CPU generator in
# the reference's order, so after train() the generator is exactly where the reference leaves it.
# --------------------------------------------------------------------------------------------
def draw_sampler_indices(n, B, out):
    """One `next(iter(DataLoader(shuffle=True)))`: dataloader.py:706-710 (base seed, unused),
    sampler.py:163-165 (sampler seed), then the first B of randperm(n) on a private generator."""
    torch.empty((), dtype=torch.int64).random_()
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    ops.randperm_prefix(seed, n, B, out)
    return seed


def _rng_skip(n):
    """Advance torch's global CPU generator by n 32-bit outputs without producing them."""
    if n <= 0:
        return
    from . import _lib
    st = torch.get_rng_state()
    _lib.call("gm_mt19937_skip", st.data_ptr(), st.numel(), n)
    torch.set_rng_state(st)


_SKIP_OK = None


def _skip_supported():
    """gm_mt19937_skip understands this torch build's serialized CPU generator state (checked once,
    on a copy: the global generator is not touched).  If not, data-parallel ranks simply draw the
    full tensors -- slower on the host, same results."""
    global _SKIP_OK
    if _SKIP_OK is None:
        from . import _lib
        try:
            st = torch.get_rng_state().clone()
            _lib.call("gm_mt19937_skip", st.data_ptr(), st.numel(), 1)
            _SKIP_OK = True
        except Exception:                      # noqa: BLE001  (unknown layout: fall back for good)
            _SKIP_OK = False
    return _SKIP_OK


def draw_rows(dst, r0, r1, kind="normal"):
    """Fill rows [r0, r1) of the contiguous fp32 tensor dst[B, W] exactly as `dst.normal_()` /
    `dst.uniform_()` would, leave the other rows untouched, and leave the global CPU generator
    where the full draw would have left it.  A data-parallel rank only needs its own rows of each
    noise tensor; the other ranks' share of the stream is skipped (gm_mt19937_skip) instead of
    generated.  Relies on ATen's CPU kernels for contiguous fp32 tensors of >= 16 elements
    (DistributionTemplates.h normal_fill / uniform via a serial loop): one 32-bit output per
    element, element i depending only on outputs of its own 16-element group -- so a 16-aligned
    row range is self-contained.  Anything else falls back to the full draw."""
    B, W = dst.shape
    e0, e1, total = r0 * W, r1 * W, B * W
    aligned = (e0 % 16 == 0 and e1 % 16 == 0 and total % 16 == 0 and e1 - e0 >= 16
               and dst.is_contiguous() and dst.dtype == torch.float32)
    if not aligned or (r0 == 0 and r1 == B) or not _skip_supported():
        getattr(dst, kind + "_")()
        return
    _rng_skip(e0)
    getattr(dst.view(-1)[e0:e1], kind + "_")()
    _rng_skip(total - e1)


_POOL = None


def _prefetch_pool():
    """One process-wide worker thread for the host RNG replay (its first HIP call pays a one-time
    per-thread runtime initialisation, so engines share it; a single worker also keeps the global
    generator's draw order strictly sequential)."""
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=1, thread_name_prefix="gm-rng-prefetch")
    return _POOL


class HostReplay:
    """The reference's draw protocol replayed by ONE C call per sub-chunk of iterations
    (csrc/gm_hostrng.cpp: vectorized mt19937 + ATen's float uniform_/normal_/random_/randint
    transformations restated) instead of one torch call per draw.  Before first use the
    restatement is checked bit for bit against torch itself on a CLONE of the global generator
    (values and final generator state, for every draw kind); if nothing matches on this host /
    torch build, the engines keep drawing through torch (slower, same results)."""
    _flavour = None          # None: not probed; False: unavailable; int: gm_host_replay_flavour

    @classmethod
    def available(cls):
        if cls._flavour is None:
            import os
            cls._flavour = cls._selfcheck()
            if cls._flavour is not False:
                from . import _lib
                _lib.call("gm_host_replay_flavour", cls._flavour)
                _lib.call("gm_host_replay_threads", max(1, int(os.environ.get("GM_HOST_THREADS", "1"))))
        return cls._flavour is not False

    @staticmethod
    def op(kind, n, dst, iter_stride, a=0, b=0, c=0, e0=0, e1=None):
        from ._lib import DrawOp
        return DrawOp(kind, n, a, b, c, dst.data_ptr(), iter_stride, e0, n if e1 is None else e1)

    @staticmethod
    def call(state, ops, n_iters):
        """Advance the serialized generator `state` (uint8 tensor) through n_iters iterations of the
        program.  Returns the C return code (0 ok, GM_EUNSUPPORTED: shape outside the restated paths)."""
        import ctypes
        from . import _lib
        arr = ops if isinstance(ops, ctypes.Array) else (_lib.DrawOp * len(ops))(*ops)
        return _lib.load().gm_host_replay(state.data_ptr(), state.numel(), arr, len(arr), n_iters)

    @classmethod
    def run(cls, ops, n_iters):
        """Replay on torch's GLOBAL CPU generator."""
        from . import _lib
        st = torch.get_rng_state()
        rc = cls.call(st, ops, n_iters)
        if rc == _lib.GM_EUNSUPPORTED:
            return False
        _lib.check(rc, "gm_host_replay")
        torch.set_rng_state(st)
        return True

    @classmethod
    def _selfcheck(cls):
        from . import _lib
        from ._lib import DRAW_INFO, DRAW_NORMAL, DRAW_SAMPLER, DRAW_UNIFORM
        g = torch.Generator()
        g.set_state(torch.get_rng_state())            # a clone: the global generator is not touched
        g.manual_seed(0x5EED0DD)
        torch.empty(7).normal_(generator=g)           # leave the twist boundary / cache mid-way
        s0 = g.get_state()
        N, B, zd, nd, nc = 1000, 24, 4, 10, 6
        # torch's side, draw by draw, on the private generator
        ref = {}
        torch.empty((), dtype=torch.int64).random_(generator=g)
        seed = int(torch.empty((), dtype=torch.int64).random_(generator=g).item())
        ref["idx"] = torch.randperm(N, generator=torch.Generator().manual_seed(seed))[:B]
        ref["n1"] = torch.empty(5120 + 17).normal_(generator=g)
        ref["u1"] = torch.empty(1003).uniform_(generator=g)
        zz = torch.empty(B, zd).normal_(generator=g)
        cat = torch.randint(0, nd, (B,), dtype=torch.long, generator=g)
        cc = torch.empty(B, nc).normal_(generator=g)
        info = torch.zeros(B, zd + nd + nc)
        info[:, :zd] = zz
        info[torch.arange(B), zd + cat] = 1
        info[:, zd + nd:] = cc
        ref["n2"] = torch.empty(64).normal_(generator=g)
        s1 = g.get_state()
        out = dict(idx=torch.empty(B, dtype=torch.int64), n1=torch.empty(5120 + 17),
                   u1=torch.empty(1003), info=torch.empty(B, zd + nd + nc), n2=torch.empty(64))
        ops = [cls.op(DRAW_SAMPLER, B, out["idx"], 0, a=N), cls.op(DRAW_NORMAL, 5137, out["n1"], 0),
               cls.op(DRAW_UNIFORM, 1003, out["u1"], 0),
               cls.op(DRAW_INFO, B, out["info"], 0, a=zd, b=nd, c=nc),
               cls.op(DRAW_NORMAL, 64, out["n2"], 0)]
        ref["info"] = info
        for flavour in (1, 2, 0):
            if _lib.load().gm_host_replay_flavour(flavour) != 0:
                continue
            st = s0.clone()
            if cls.call(st, ops, 1) != 0:
                continue
            if torch.equal(st, s1) and all(torch.equal(out[k], ref[k]) for k in out):
                return flavour
        return False


class _FillJob:
    """A sub-chunk of draws queued on the native fill worker (gm_fill_submit); the same surface as
    the prefetch pool's future."""
    __slots__ = ("id",)

    def __init__(self, job_id):
        self.id = job_id

    def done(self):
        from . import _lib
        return _lib.load().gm_fill_completed() >= self.id

    def result(self):
        from . import _lib
        rc = _lib.load().gm_fill_wait(self.id)
        if rc != 0:
            raise GMError("host draw replay failed on the fill worker (rc=%d)" % rc)


class NumpyReplay:
    """np.random.normal on numpy's LEGACY global generator (what bir_vae.py:92-94 draws from) replayed in C
    (csrc/gm_hostrng.cpp gm_numpy_legacy_normal_f32): same MT19937 stream, same polar method, same libm calls,
    float32 written straight into the pinned ring -- 10 240 samples cost numpy 188 us (+ 13 us for the .float()
    copy), this 60 us (candidate stage on AVX-512); the log / sqrt stage can be split over GM_NUMPY_THREADS
    (default 1: no gain measured).  Checked bit for bit (values and final state)
    against a private numpy RandomState before first use; if it does not match on this host the engine keeps
    calling numpy."""
    _ok = None

    @classmethod
    def available(cls):
        if cls._ok is None:
            import os
            cls._ok = cls._selfcheck()
        return cls._ok

    @staticmethod
    def threads():
        import os
        from . import _cpu_quota_cores
        q = _cpu_quota_cores()
        cap = int(q) if q else (os.cpu_count() or 1)
        # (measured on the GPU box, BIR-VAE epoch loop: 1 thread 172 us per iteration, 2: 179, 4: 177 -- the
        # per-chunk parts are too short for sleeping workers to pay off; numpy itself: 225)
        return max(1, min(int(os.environ.get("GM_NUMPY_THREADS", "1")), cap))

    @staticmethod
    def _call(state, loc, scale, n, dst_ptr, threads):
        """state: [key(uint32[624] array), pos, has_gauss, gauss] advanced in place."""
        import ctypes
        from . import _lib
        pos, hg, cg = ctypes.c_int32(state[1]), ctypes.c_int32(state[2]), ctypes.c_double(state[3])
        _lib.call("gm_numpy_legacy_normal_f32", state[0].ctypes.data, ctypes.addressof(pos), ctypes.addressof(hg),
                  ctypes.addressof(cg), float(loc), float(scale), int(n), dst_ptr, int(threads))
        state[1], state[2], state[3] = pos.value, hg.value, cg.value

    @staticmethod
    def _unpack(st):
        if st[0] != "MT19937":
            return None
        return [np.ascontiguousarray(st[1], dtype=np.uint32).copy(), int(st[2]), int(st[3]), float(st[4])]

    # numpy's own lock makes one np.random.normal call atomic; the replay reads the global state out, advances the
    # copy in C and writes it back, so everything in this package that reads or writes that state (the fills on the
    # prefetch thread, the BIR-VAE checkpoint) does it under this lock.  np.random calls made by OTHER code while a
    # BIR-VAE engine is drawing ahead are outside it: do not use the global numpy generator concurrently with one.
    STATE_LOCK = __import__("threading").RLock()

    @classmethod
    def fill(cls, scale, dst, B, Z, sizes):
        """dst[k].view(-1)[:b*Z] = float32(np.random.normal(0, scale, (b, Z))) for the chunk's batches, in order,
        on numpy's GLOBAL generator: the leading full batches as ONE call (they are contiguous in dst)."""
        with cls.STATE_LOCK:
            return cls._fill_locked(scale, dst, B, Z, sizes)

    @classmethod
    def _fill_locked(cls, scale, dst, B, Z, sizes):
        state = cls._unpack(np.random.get_state(legacy=True))
        if state is None:
            return False
        thr, stride, k = cls.threads(), B * Z * 4, 0
        while k < len(sizes):
            b, run = sizes[k], 1
            if b == B:
                while k + run < len(sizes) and sizes[k + run] == B:
                    run += 1
            cls._call(state, 0.0, scale, (run * B * Z) if b == B else b * Z, dst.data_ptr() + k * stride, thr)
            k += run
        np.random.set_state(("MT19937", state[0], state[1], state[2], state[3]))
        return True

    @classmethod
    def _selfcheck(cls):
        try:
            rs = np.random.RandomState(0x5EED)
            rs.normal(size=5)                         # leave a cached second value behind
            rs.random_sample(301)                     # and the position away from a multiple of 4
            state = cls._unpack(rs.get_state(legacy=True))
            if state is None:
                return False
            ok = True
            for n, thr in ((1, 1), (10240 * 3, 2), (7, 1), (4097, 3)):
                ref = torch.from_numpy(rs.normal(0.0, 0.37, n)).float()
                got = torch.empty(n)
                cls._call(state, 0.0, 0.37, n, got.data_ptr(), thr)
                ok = ok and torch.equal(ref.view(torch.int32), got.view(torch.int32))
            end = rs.get_state(legacy=True)
            return bool(ok and np.array_equal(end[1], state[0]) and (int(end[2]), int(end[3]), float(end[4])) ==
                        (state[1], state[2], state[3]))
        except Exception:                             # noqa: BLE001  (library without the symbol, ...)
            return False


class GANEngine(CriticStep, InfoQStep, GeneratorStep, PenaltySteps):
    """Graph-captured D_steps x train_D + train_G iteration for the score-based GAN variants
    (ns, mm, w, ls, ra, f, fisher, wgp, info, dra; BEGAN: began_engine.BEGANEngine).  This class is the CORE: buffers and
    switches, rings and host draws, graph capture, run(); what an iteration launches lives in gan_steps.py."""

    SUPPORTED = ("ns", "mm", "w", "ls", "ra", "f", "fisher", "wgp", "info", "be", "dra")

    def __init__(self, variant, model, data, B, device, method=None, use_graph=True,
                 world_size=1, rank=0, process_group=None, force_dp=False):
        """force_dp: run the data-parallel launch structure on one rank (diagnostics / tests)."""
        assert variant in self.SUPPORTED, variant
        from . import _respect_cpu_quota
        _respect_cpu_quota(force=False)             # once per process, when the first engine is built
        self.variant, self.model, self.device = variant, model, device
        self.method = method
        self.loss_key = ("f_" + method) if variant == "f" else \
            {"wgp": "w", "info": "ns", "dra": "ns"}.get(variant, variant)
        self.out_act = "relu" if variant == "wgp" else "sigmoid"
        self.B = B                         # GLOBAL batch (reference semantics)
        self.world, self.rank, self.pg = world_size, rank, process_group
        assert B % world_size == 0, "global batch must divide across ranks"
        self.Bl = B // world_size          # rows this rank computes
        # resident dataset [N, I]: 1 bit per pixel when it is binary (the reference's MNIST is,
        # utils.py:31; GM_PACKED=0 keeps fp32 rows), else fp32
        import os as _os
        if not isinstance(data, ops.PackedData) and _os.environ.get("GM_PACKED", "1") != "0" \
                and ops.PackedData.is_binary(data):
            data = ops.PackedData(data)
        self.data = data
        self.N, self.I = data.shape
        G, D = model.G, model.D
        import os
        # data-parallel gradient exchange: "peer" = kernels inside the iteration graph over hipIpc /
        # xGMI peer mappings (csrc/gm_comm.hip; falls back to "rccl" if its self-check fails);
        # "rccl" = host-launched torch.distributed all-reduces between per-segment graphs
        self.comm_mode = os.environ.get("GM_DP_COMM", "peer")
        self.force_segments = bool(force_dp)
        self._comms = None
        self.comm_memory, self.comm_fallback = "none", None
        if (world_size > 1 or force_dp) and self.comm_mode == "peer":
            sizes = {"D": sum(_align4(p.numel()) for p in D.parameters()),
                     "G": sum(_align4(p.numel()) for p in G.parameters())}
            if variant == "info":                    # third optimizer: G u Q (info_gan.py:146-148)
                sizes["Q"] = sum(_align4(p.numel()) for p in model.Q.parameters())
            self._setup_peer_comm(sizes)
        # the fallback exchange: RCCL all-reduces CAPTURED into the iteration's graph (round 5; GM_RCCL_IN_GRAPH=0 keeps
        # round 1's host-launched collectives between segment graphs)
        self._rccl = None
        if (world_size > 1 or force_dp) and self.comm_mode == "rccl" and use_graph and \
                os.environ.get("GM_RCCL_IN_GRAPH", "1") != "0":
            self._setup_rccl_graph_comm()
        alloc = (lambda net: (lambda n: self._comms[net].grad_buffer()[:n])) if self._comms else \
            (lambda net: None)
        self.fG = FlatParams(G.parameters(), device, grad_alloc=alloc("G"))
        self.fD = FlatParams(D.parameters(), device, grad_alloc=alloc("D"))
        g1, g2 = list(G.children())[:2]
        d1, d2 = list(D.children())[:2]
        self.G1, self.G2 = _Linear(self.fG, g1), _Linear(self.fG, g2)
        self.D1, self.D2 = _Linear(self.fD, d1), _Linear(self.fD, d2)
        self.Z = self.G1.W.shape[1]
        self.H = self.G1.W.shape[0]
        self.Hd_dim = self.D1.W.shape[0]
        assert self.D2.W.shape[0] == 1 or variant == "be", "score-based critics (BEGAN: autoencoder)"
        self.use_graph = use_graph
        # Launch-fusion forms of the step.  Rounds 1 - 5 kept an environment switch per form for same-box A/Bs; every
        # slower arm has its measurement committed (profiles/r01 .. r05_experiments.md) and round 6 removed the switches
        # (VERDICT r5 item 8).  The attributes stay: the unfused forms are still what the data-parallel, many-row and
        # penalty steps take where a fused one does not apply (_adam_in_epilogue, _fold_ok, _tick_in_head ...).
        self.fuse_head = self.fuse_adam = self.fold_tick = self.ride_head_dx = True
        self.group_head = self.ride_gather = self.pair_dw = self.batch_gen_env = True
        # folded critic head (round 3): no launch for the N = 1 layer -- partial dots in the hidden
        # layer's forward epilogue, scores / losses / dS rebuilt in the consumers' prologues, dH formed
        # in registers (csrc/gm_head.h).  "2": also where the forward would take the LDS macro-tile kernel
        self.fold_env = os.environ.get("GM_FOLD_HEAD", "1")
        # iterations per graph (largest captured size; powers of two below it are captured too).  A graph launch costs
        # 22 us of GPU time beyond its iterations (hand-off between graphs + the stage-in of its draws:
        # tools/piece_cost_probe.py, profiles/r05_experiments.md section 10).  GM_GRAPH_ITERS fixes it; otherwise
        # configure() picks 128 / 64 / 32 with the ring the draws of this variant afford (_ring_and_graph_size)
        self.graph_iters = max(1, int(os.environ.get("GM_GRAPH_ITERS", "32")))
        # launch graphs ahead of the host draws they consume; the stage-in kernel waits on the fill gate
        self.gated = True
        self._early_submit = True
        # (rounds 4 - 5 also issued every piece's stage-in on a side stream ahead of its graph: with the gate wait as its
        # own one-wave kernel in front of an ungated copy the in-graph stage-in is the faster form everywhere -- a launch
        # of k iterations 22 + 67.5 k us against 31 + 67.5 k us, profiles/r05_experiments.md section 10 -- removed)
        self._gate = None
        self._standalone_G = False
        Bl, I, H, Hd = self.Bl, self.I, self.H, self.Hd_dim
        dev = device
        z = lambda *s: torch.zeros(*s, device=dev)
        # rows [0,Bl): real batch, [Bl,2Bl): G(z) of the critic step, [2Bl,3Bl): G(z) of the
        # generator step -- contiguous so that both generator forwards can be ONE 2Bl-row launch
        # WGAN-GP: the penalty's gamma rows sit right BEFORE [x ; G(zD)] (and u right before dH) so
        # that dW1 = [u ; dH]^T [gamma ; X2] is ONE weight-gradient GEMM over 3B rows
        self.XX4 = z(4 * Bl, I)
        self.XX = self.XX4[Bl:]
        self.X2, self.Xg2 = self.XX[:2 * Bl], self.XX[2 * Bl:]
        self.HG = z(2 * Bl, H)
        self.Hg, self.Hg2 = self.HG[:Bl], self.HG[Bl:]
        self.Hd = z(2 * Bl, Hd)
        self.S2 = z(2 * Bl)                # scores
        self.dS = z(2 * Bl)                # d loss / d pre-activation score
        self.DU = z(3 * Bl, Hd)
        self.dHd = self.DU[Bl:]
        self.dXg = z(Bl, I)
        self.dHg = z(Bl, H)
        self.rowloss = z(2 * Bl)
        self.fold = ops.HeadFold(2 * Bl, Hd, dev) if (self.D2.W.shape[0] == 1 and Hd <= 512) else None
        self.aux = z(8)                    # Fisher lambda + moments
        self.pre = z(16)                   # data parallel: scalars exchanged between the loss phases
        if variant in ("wgp", "dra"):
            self.Xh, self.Hh, self.Sh = z(Bl, I), z(Bl, Hd), z(Bl)
            self.merge_fwd3 = True
            if self.merge_fwd3:
                # D's hidden layer on [x_hat ; x ; G(z)] as ONE 3B-row launch: x_hat lives in the first row block
                # of XX4 (WGAN-GP writes gamma there LATER, when x_hat has been consumed), its hidden activations
                # right in front of the other two blocks'
                self.HH3 = z(3 * Bl, Hd)
                self.Hh, self.Hd = self.HH3[:Bl], self.HH3[Bl:]
                self.Xh = self.XX4[:Bl]
            self.Gr, self.T = z(Bl, I), z(Bl, Hd)
            if variant == "wgp":
                self.U, self.Gam = self.DU[:Bl], self.XX4[:Bl]
                self.gw2_pen = z(Hd)
                self.pen_in_head = os.environ.get("GM_WGP_PEN_IN_HEAD", "1") != "0"
            else:
                self.U, self.Gam = z(Bl, Hd), z(Bl, I)
            self.pen = z(Bl)
        if variant == "dra":
            self.da2, self.dA1, self.stdv = z(Bl), z(Bl, Hd), z(1)
            self.dra_stack = self.merge_fwd3 and os.environ.get("GM_DRA_STACK", "1") != "0"
            if self.dra_stack:
                # round 4: the critic's three layer-1 weight gradients as ONE GEMM over 4B rows,
                #   dW1 = [u ; dA1 ; dH_x ; dH_g]^T [dv ; x_hat ; x ; G(zD)]      (dra_gan.py:207-223)
                # (u's rows do not reach db1: `ones_from` = B).  Both operands contiguous: dv and x_hat in front of
                # [x ; G(zD) ; G(zG)] (the last two stay adjacent: both generator forwards are one launch), u and
                # dA1 in front of dH.  The merged forward reads rows [B, 4B) = [x_hat ; x ; G(zD)].
                self.XX5 = z(5 * Bl, I)
                self.Gam, self.Xh = self.XX5[:Bl], self.XX5[Bl:2 * Bl]
                self.XX = self.XX5[2 * Bl:]
                self.XX4 = self.XX5[Bl:]                       # [x_hat ; x ; G(zD) ; G(zG)]: what the merged forward indexes
                self.X2, self.Xg2 = self.XX[:2 * Bl], self.XX[2 * Bl:]
                self.DU4 = z(4 * Bl, Hd)
                self.U, self.dA1, self.dHd = self.DU4[:Bl], self.DU4[Bl:2 * Bl], self.DU4[2 * Bl:]
                self.gw2_pen, self.gb2_pen = z(Hd), z(1)
            from . import ops_fused as _of
            self.std_ws = _of.std_workspace(dev)
        if variant == "info":
            # InfoGAN (info_gan.py:78-148): auxiliary net Q and a third optimizer over G u Q that
            # keeps its OWN Adam moments for G's parameters
            Q = model.Q
            self.fQ = FlatParams(Q.parameters(), device, grad_alloc=alloc("Q"))
            q1, q2 = list(Q.children())[:2]
            self.Q1, self.Q2 = _Linear(self.fQ, q1), _Linear(self.fQ, q2)
            self.mi_m, self.mi_v = torch.zeros_like(self.fG.m), torch.zeros_like(self.fG.v)
            self.G1mi = _Linear(self.fG, g1, m=self.mi_m, v=self.mi_v)
            self.G2mi = _Linear(self.fG, g2, m=self.mi_m, v=self.mi_v)
            self.zd, self.nd, self.nc = model.z_dim, model.disc_dim, model.cont_dim
            nq = self.Q2.W.shape[0]
            self.Hq, self.Qo, self.dQo, self.dHq = z(Bl, self.Q1.W.shape[0]), z(Bl, nq), z(Bl, nq), \
                z(Bl, self.Q1.W.shape[0])
        self.ctr = torch.zeros(1, dtype=torch.int64, device=dev)
        self.graph = None
        self._graph_key = None

    # -- slots --------------------------------------------------------------------------------
    # -- which fusions apply to this run ---------------------------------------------------------
    def _single(self):
        return self.world == 1 and not self.force_segments

    def _peer(self):
        """Data parallel with the in-graph peer exchange (one hipGraph per iteration, Adam applied by
        the all-gather kernel)."""
        return not self._single() and self.comm_mode == "peer"

    def _rccl_in_graph(self):
        return not self._single() and self.comm_mode == "rccl" and self._rccl is not None

    def _one_graph(self):
        return self._single() or self._peer() or self._rccl_in_graph()

    def _tick_in_head(self):
        """The per-graph tick rides in the generator step's head_bwd kernel."""
        return self.use_graph and self.fold_tick and self.fuse_head

    def _adam_in_epilogue(self, net):
        """Adam folded into the gradient-producing kernels (single GPU; not when later kernels
        still accumulate into the gradients, i.e. WGAN-GP's critic)."""
        if not (self.fuse_adam and self._single()):
            return False
        if net == "D":
            if self.variant == "dra":
                return self._dra_stacked()
            if self.variant in ("ra", "fisher"):
                return self._fold_head()
            return self.fuse_head and (self.variant != "wgp" or self._wgp_stacked())
        return True

    def _fold_ok(self, rows):
        """The critic's N = 1 head without a launch of its own (ns_gan.py:57-60 + the loss lines):
        the riding kernels, 16-byte aligned layer widths, row counts the consumers can keep in LDS;
        by default only where the hidden layer's forward is a split-reduction launch anyway (rows
        below the LDS macro-tile kernel's threshold; GM_FOLD_HEAD=2 lifts that)."""
        if self.fold_env == "0" or self.fold is None or self.variant == "be":
            return False
        if not (self.fuse_head and self.group_head and self.ride_head_dx):
            return False
        if self.Hd_dim % 4 or self.I % 4 or rows > 2048 or self.Hd_dim > 512:
            return False
        return self.fold_env == "2" or rows < ops.lds_min_m()

    def _fold_head(self):
        """Critic step: the separable losses whose whole step runs on the fused head (not the
        penalty variants' stacked / accumulating steps); round 4: RaGAN's and Fisher's too, on one GPU -- every
        consumer workgroup holds all rows' scores, so the batch means are block reductions in its prologue (under
        data parallelism their phases sit around scalar exchanges: gm_gan_loss_phase)."""
        import os
        if self.variant in ("ra", "fisher"):
            return self._single() and not self.force_segments and self._fold_ok(2 * self.Bl) and \
                os.environ.get("GM_FOLD_HEAD_TP", "1") != "0"
        return self.variant in ("ns", "mm", "w", "ls", "f", "info") and self._fold_ok(2 * self.Bl)

    def _fold_head_G(self):
        """Generator step: every variant's D(G(z)) pass is the plain fused head in generator mode."""
        import os
        return self._fold_ok(self.Bl)

    def _wgp_stacked(self):
        """WGAN-GP critic step with the second backward folded into the first-order launches: the
        layer-1 weight gradient as one stacked GEMM (+Adam), the w2 share added in the head's
        backward, D(x_hat)'s head + u as one kernel."""
        import os
        return self.variant == "wgp" and self.fuse_head and self.group_head and \
            os.environ.get("GM_WGP_STACK", "1") != "0"

    def _dra_stacked(self):
        """DRAGAN critic step with its three layer-1 weight gradients as one stacked GEMM (+ Adam) and the sigma''
        path's share of the head's gradient added inside the head's backward (round 4)."""
        return self.variant == "dra" and getattr(self, "dra_stack", False) and self.fuse_head and \
            self.group_head

    def _slot(self, it, mul, add, ring, stride, post=False):
        """Graph mode: resolved on device from the counter; eager mode: resolved here.
        post=True: the consumer runs after the folded tick of this iteration (ctr already +1)."""
        if self.use_graph:
            if post and self._tick_in_head():
                add -= mul
            return ops.slot(self.ctr.data_ptr(), mul, add, ring, stride)
        i = it * mul + add
        if ring > 0:
            i %= ring
        return ops.slot(0, 0, i, 0, stride)

    # -- one iteration = D_steps critic steps + one generator step -----------------------------
    def _segments(self):
        """The iteration as launch segments separated by the gradient all-reduces (SURVEY.md 8e:
        exactly two collectives per D+G step at D_steps=1):
            [D fwd+bwd] AR(D) [Adam D | next D fwd+bwd] AR(D) ... [Adam D | G fwd+bwd] AR(G) [Adam G | tick]
        Returns [(fn(st, it), flat_grad_to_allreduce_after_or_None), ...]."""
        d = self.D_steps
        segs = []
        def seg(parts, ar):
            def run(st, it, parts=parts):
                for f in parts:
                    f(st, it)
            segs.append((run, ar))
        pending = []
        for j in range(d):
            pending.append(lambda st, it, j=j: self._issue_D_pre(st, it, j))
            seg(pending, ("D", j))
            pending = [lambda st, it, j=j: self._issue_D_post(st, it, j)]
        pending.append(lambda st, it: self._issue_G_pre(st, it))
        seg(pending, ("G", 0))
        tail = [lambda st, it: self._issue_G_post(st, it)]
        if self.variant == "info":
            tail.append(lambda st, it: self._issue_Q(st, it))
        if self.variant == "be":
            tail.append(lambda st, it: self._issue_end(st, it))       # K / schedulers / tick
        elif self.use_graph and not self._tick_in_head():
            tail.append(lambda st, it: ops.tick(self.ctr, 1, stream=st))
        seg(tail, None)
        return segs

    def _issue_iteration(self, st, it):
        """All segments back to back (one hipGraph when use_graph): single GPU, or data parallel with
        the in-graph peer exchange."""
        for run, ar in self._segments():
            run(st, it)
            if ar is not None:
                self._allreduce(ar, st, it)

    def _allreduce(self, ar, st=None, it=0):
        """Gradient exchange after a backward segment.  ar = ("D", j) | ("G", 0)."""
        if self._single():
            return
        net, j = ar
        fp = self.fD if net == "D" else self.fG
        if self._peer():
            # all-reduce + optim.Adam.step in the all-gather kernel (ns_gan.py:139,156)
            if net == "D":
                slot, clamp = self._slot(it, self.D_steps, j, 0, 1), self.clip
                sched, scale = self.schedD, self._lr_scale("D")
            else:
                slot, clamp = self._G_sched_slot(it), 0.0
                sched, scale = self.schedG, self._lr_scale("G")
            self._comms[net].allreduce_adam(fp.grad, fp.flat, fp.m, fp.v, sched, slot, clamp=clamp,
                                            lr_scale=scale, stream=st)
            return
        if self._rccl_in_graph():
            self._rccl.allreduce(fp.grad, stream=st)  # a node of the iteration's graph; Adam follows in _issue_*_post
            return
        from . import dp
        dp.allreduce_sum_(fp.grad, self.pg)

    def _lr_scale(self, net):
        return None                                 # BEGAN: the plateau schedulers' device-side scale

    def _dp(self):
        return not self._single()

    def _exchange_scalars(self, st, vals, k):
        """SUM over ranks of vals[0..k) inside the graph (the pre-reductions of the losses that are not
        a mean of per-sample terms)."""
        self._comms["D"].allreduce_scalars(vals, k, stream=st)

    def _issue_D(self, st, it, j):
        self._issue_D_pre(st, it, j)
        self._allreduce(("D", j), st, it)
        self._issue_D_post(st, it, j)

    def _issue_G(self, st, it):
        self._issue_G_pre(st, it)
        self._allreduce(("G", 0), st, it)
        self._issue_G_post(st, it)

    # -- host prefetch of one chunk of iterations ---------------------------------------------
    AHEAD = 3           # x SUB iterations of host draws may be submitted and unfinished
    RAMP = (1, 1, 2, 4, 8, 16)   # sub-chunk sizes of the first fills of a cold run
    FIRST_PIECE = 2     # iterations in the first graph of a cold run (see _plan)
    GATE_TIMEOUT_S = 20.0

    def _ring_and_graph_size(self, D_steps):
        """(ring slots, iterations of the largest graph).  Longer graphs amortise the 22 us a graph launch costs beyond
        its iterations (NSGAN bs=256, 4096 steps x 3, same-call alternation: 32 per graph / 128 slots 68.0 us per step,
        64 / 256 67.5 - 67.8, 128 / 512 67.4 - 67.6; profiles/r05_experiments.md section 10); the ring holds four of
        the largest graphs so that the host draws, the copy and the GPU never wait for one another's slots.  But a
        graph waits for ALL its iterations' draws, and the host draws only as fast as the batch is small: at bs=1024
        (172 KB of draws per iteration, 60 us of host time against 143 us on the GPU) 128-iteration graphs made a
        cold 600-step run 2 % SLOWER (147 against 144.4 us per step).  So: 128 per graph up to 64 KB of draws per
        iteration (the GLOBAL batch's indices and noise; NSGAN bs=256: 42 KB), 64 up to 128 KB, else 32.
        GM_RING / GM_GRAPH_ITERS fix either.
        DRAGAN stages 0.8 MB per critic step at B = 256 and keeps <= 64 slots (16 slots, round 1's choice, serialised
        host fills and GPU graphs: a 16-iteration graph had to FINISH before its slots could be refilled)."""
        import os
        B, Z, d = self.B, self.Z, D_steps
        per_it = d * B * (8 + 4 * Z) + B * 4 * Z                      # indices + zD per critic step, zG
        if self.variant == "wgp":
            per_it += d * B * 4
        if self.variant == "info":
            per_it += B * 4 * Z
        if self.variant == "dra":
            per_it += d * B * 4 * (1 + self.I)
        env_ring, env_gi = os.environ.get("GM_RING"), os.environ.get("GM_GRAPH_ITERS")
        if env_ring:
            ring = int(env_ring)
        else:
            ring = 512 if per_it <= (64 << 10) else 256 if per_it <= (128 << 10) else GAN_RING
        R = max(1, min(ring, 64) if self.variant == "dra" else ring)
        gi = max(1, int(env_gi)) if env_gi else max(32, min(128, R // 4))
        return R, gi

    def _alloc_rings(self, R):
        """Device rings of R iterations and PINNED host rings of the same layout.  The host replay
        writes sub-chunks straight into the host rings; the first kernel of every graph
        (gm_stage_in) pulls its own iterations' slots into the device rings."""
        d, B, Z, dev = self.D_steps, self.B, self.Z, self.device
        self.R = R
        # iterations per fill job: the fill gate advances per JOB, so this is the granularity at which graphs see their
        # draws arrive -- independent of the graph size (as one 128-iteration job a 4-iteration graph at an epoch's start
        # waited 1.9 ms for draws it did not need: Trainer.train() 69.5 -> 75.3 us per step)
        self.SUB = max(1, min(32, self.graph_iters, R))
        self.z_joint = self._batch_gen()
        # (leading dims per ring, pieces per iteration, trailing dims): every ring is a sequence of
        # [B, *tail] draws, `pieces` of them per iteration
        layout = {"idx": ((R * d,), d, (), torch.int64)}
        if self.z_joint:          # one ring of [zD; zG] pairs: slot stride 2*B*Z
            layout["z"] = ((R, 2), 2, (Z,), torch.float32)
        else:
            layout["zD"] = ((R * d,), d, (Z,), torch.float32)
            layout["zG"] = ((R,), 1, (Z,), torch.float32)
        if self.variant == "wgp":
            layout["eps"] = ((R * d,), d, (), torch.float32)
        if self.variant == "info":
            layout["zQ"] = ((R,), 1, (Z,), torch.float32)
        if self.variant == "dra":
            layout["delta"] = ((R * d,), d, (), torch.float32)
            layout["U"] = ((R * d,), d, (self.I,), torch.float32)
        # Data parallel: the HOST rings hold every draw of the GLOBAL batch (the replay advances the one
        # global generator; it materialises this rank's rows at their global offsets), the DEVICE rings
        # only this rank's rows -- the stage-in pulls Bl of every B rows over PCIe, so its cost per
        # iteration does not grow with the number of ranks (at 8 x 256 rows the global slot is 344 KB =
        # 15.6 us of a ~85 us step; the rank's share is 43 KB).  DRAGAN's uniforms go through the copy
        # engine (_copy_U) in whole slots and keep the global layout.
        self.local_rings = self._local_rings()
        self.ring_B = self.Bl if self.local_rings else B          # rows per draw in the device rings
        self.ring_r0 = 0 if self.local_rings else self.rank * self.Bl   # my first row inside them
        shapes = {k: (lead + (B,) + tail, dt) for k, (lead, m, tail, dt) in layout.items()}
        dshapes = {k: (lead + (self.ring_B,) + tail, dt) for k, (lead, m, tail, dt) in layout.items()}
        self.dring = {k: torch.zeros(*sh, dtype=dt, device=dev) for k, (sh, dt) in dshapes.items()}
        self.hring = {k: torch.zeros(*sh, dtype=dt).pin_memory() for k, (sh, dt) in shapes.items()}
        self.idx_ring = self.dring["idx"]
        rB = self.ring_B
        if self.z_joint:
            self.z_ring = self.dring["z"]
            self.zD_ring, self.zG_ring = self.z_ring[:, 0], self.z_ring[:, 1]
            flat = self.z_ring.view(-1)
            self.zD_base, self.zG_base = flat, flat[rB * Z:]
            self.zD_stride = self.zG_stride = 2 * rB * Z
        else:
            self.zD_ring, self.zG_ring = self.dring["zD"], self.dring["zG"]
            self.zD_base, self.zG_base = self.zD_ring.view(-1), self.zG_ring.view(-1)
            self.zD_stride = self.zG_stride = rB * Z
        self.eps_ring, self.zQ_ring = self.dring.get("eps"), self.dring.get("zQ")
        self.delta_ring, self.U_ring = self.dring.get("delta"), self.dring.get("U")
        # stage-in segments: (device-visible address of the host ring, device ring, bytes / iteration)
        import ctypes
        from . import _lib
        segs = []
        # DRAGAN's uniforms (B x 784 per critic step, 0.8 MB at B = 256) do not go through the in-graph
        # stage-in -- 36 us of serial GPU-initiated PCIe reads per iteration -- but through the copy
        # engine on a side stream, overlapping the previous piece's kernels (_copy_U)
        import os
        self._u_copy = "U" in shapes
        self._u_copied, self._u_stream = 0, None
        for k in shapes:
            if k == "U" and self._u_copy:
                continue
            h, dv = self.hring[k], self.dring[k]
            devp = ctypes.c_void_p()
            _lib.call("gm_host_device_ptr", h.data_ptr(), ctypes.byref(devp))
            if not self.local_rings:
                segs.append(_lib.StageSeg(devp.value, dv.data_ptr(), h.numel() * h.element_size() // R))
                continue
            _, m, tail, _dt = layout[k]
            wb = int(np.prod(tail, dtype=np.int64)) * h.element_size()     # bytes per row of a draw
            segs.append(_lib.StageSeg(devp.value + self.rank * self.Bl * wb, dv.data_ptr(), m * self.Bl * wb,
                                      m, 0, B * wb, self.Bl * wb))
        self._segs = (_lib.StageSeg * len(segs))(*segs)
        if getattr(self, "_gate", None) is None:
            # fill gate (gm_stage_in_gated): [0] = iterations written into the host rings since
            # configure(), [1] = raised by a stage-in kernel whose wait timed out.  Allocated once per
            # engine: captured graphs hold its address.
            self._gate = torch.zeros(2, dtype=torch.int64).pin_memory()
            self._gate_np = self._gate.numpy()
            gp = ctypes.c_void_p()
            _lib.call("gm_host_device_ptr", self._gate.data_ptr(), ctypes.byref(gp))
            self._gate_dev = gp.value
        self._views = {}
        for r in range(R):                            # every slot a sub-chunk can start at: built once
            self._host_views(r)
        self._replay_ok = HostReplay.available()
        # native fill worker (gm_fill_submit): usable when the C replay covers this variant's draws --
        # probed by running one iteration's program on a COPY of the generator state (the ring slots it
        # scribbles on are rewritten before anything reads them)
        import os
        self._native_fill = False
        if self._replay_ok:
            probe = torch.get_rng_state().clone()
            self._native_fill = HostReplay.call(probe, self._host_views(0)["program"], 1) == 0
        self._rng_state, self._rng_owned = None, False

    def _host_views(self, r):
        """Views of the host rings starting at ring slot r (what one sub-chunk's draws write), with
        the gm_draw_op program of one iteration addressed at them.  Cached per slot."""
        v = self._views.get(r)
        if v is None:
            d, h = self.D_steps, self.hring
            v = dict(idx=h["idx"][r * d:])
            if self.z_joint:
                v["z"] = h["z"][r:]
                v["zD"], v["zG"] = v["z"][:, 0], v["z"][:, 1]
            else:
                v["zD"], v["zG"] = h["zD"][r * d:], h["zG"][r:]
            for k, per in (("eps", d), ("zQ", 1), ("delta", d), ("U", d)):
                if k in h:
                    v[k] = h[k][r * per:]
            v["idx_np"] = v["idx"].numpy()
            from ._lib import DrawOp
            prog = self._program(v)
            v["program"] = (DrawOp * len(prog))(*prog)          # built once: one C call per sub-chunk
            self._views[r] = v
        return v

    def _program(self, s):
        """The draws of ONE iteration in reference order as a gm_draw_op
        list writing into the ring views `s`; HostReplay runs it for a whole sub-chunk in one C call."""
        from ._lib import DRAW_INFO, DRAW_NORMAL, DRAW_SAMPLER, DRAW_UNIFORM
        d, B, Z, I = self.D_steps, self.B, self.Z, self.I
        op = HostReplay.op
        # a data-parallel rank materialises only its own rows of the noise (16-aligned ranges of
        # the normal_ stream are self-contained, see draw_rows); the stream advances in full
        r0, r1 = self.rank * self.Bl, (self.rank + 1) * self.Bl
        rows = lambda w: dict(e0=r0 * w, e1=r1 * w) if (
            self.world > 1 and (r0 * w) % 16 == 0 and (r1 * w) % 16 == 0 and (B * w) % 16 == 0) else {}
        urows = lambda w: dict(e0=r0 * w, e1=r1 * w) if self.world > 1 else {}
        zs = (2 if self.z_joint else d) * B * Z * 4          # bytes between iterations of zD
        gs = (2 if self.z_joint else 1) * B * Z * 4
        prog = []
        for j in range(d):
            prog.append(op(DRAW_SAMPLER, B, s["idx"][j], d * B * 8, a=self.N))
            if self.variant == "info":
                prog.append(op(DRAW_INFO, B, s["zD"][j], zs, a=self.zd, b=self.nd, c=self.nc))
                continue
            prog.append(op(DRAW_NORMAL, B * Z, s["zD"][j], zs, **rows(Z)))    # ns_gan.py:183,220
            if self.variant == "wgp":
                prog.append(op(DRAW_UNIFORM, B, s["eps"][j], d * B * 4, **urows(1)))   # w_gp_gan.py:197
            if self.variant == "dra":
                prog.append(op(DRAW_UNIFORM, B, s["delta"][j], d * B * 4))             # dra_gan.py:200
                prog.append(op(DRAW_UNIFORM, B * I, s["U"][j], d * B * I * 4))         # dra_gan.py:205
        if self.variant == "info":
            prog.append(op(DRAW_INFO, B, s["zG"][0], gs, a=self.zd, b=self.nd, c=self.nc))
            prog.append(op(DRAW_INFO, B, s["zQ"][0], B * Z * 4, a=self.zd, b=self.nd, c=self.nc))
        else:
            prog.append(op(DRAW_NORMAL, B * Z, s["zG"][0], gs, **rows(Z)))    # ns_gan.py:208
        return prog

    def _issue_stage_in(self, st, it, k, max_blocks=256):
        """Stage-in of iterations [it, it+k): host ring -> device ring (first launch of a graph)."""
        from . import _lib
        ring_slot, it_slot = self._slot(it, 1, 0, self.R, 1), self._slot(it, 1, 0, 0, 1)
        if self.gated:
            _lib.call("gm_stage_in_gated", st, self._segs, len(self._segs), ring_slot, k, self._gate_dev,
                      it_slot, self.GATE_TIMEOUT_S, None, max_blocks)
        else:
            _lib.call("gm_stage_in", st, self._segs, len(self._segs), ring_slot, k)

    def _draw_info_noise(self, dst):
        """info_gan.py:306-325: [randn(B,z) | one_hot(randint(0,nd,(B,))) | randn(B,nc)].  The
        three draws are made on contiguous tensors exactly like the reference's (normal_ on a
        strided view would consume the generator differently), then packed into dst [B, z+nd+nc]."""
        B, zd, nd, nc = self.B, self.zd, self.nd, self.nc
        zz = torch.randn(B, zd)
        cat = torch.randint(0, nd, (B,), dtype=torch.long)
        cc = torch.randn(B, nc)
        dst[:, :zd] = zz
        dst[:, zd:zd + nd] = 0
        dst[torch.arange(B), zd + cat] = 1
        dst[:, zd + nd:] = cc

    def _draw_D(self, s, k):
        """Draws of one critic step in reference order (appendix A.4)."""
        draw_sampler_indices(self.N, self.B, s["idx_np"][k])
        if self.variant == "info":
            self._draw_info_noise(s["zD"][k])
            return
        self._noise(s["zD"][k], "normal")            # torch.randn(B, Z)   ns_gan.py:183,220
        if self.variant == "wgp":
            self._noise(s["eps"][k].view(-1, 1), "uniform")   # torch.rand(B, 1)  w_gp_gan.py:197
        if self.variant == "dra":
            s["delta"][k].uniform_()                 # torch.rand(B, 1)    dra_gan.py:200
            s["U"][k].uniform_()                     # torch.rand(B, 784)  dra_gan.py:205

    def _draw_G(self, s, k):
        if self.variant == "info":
            self._draw_info_noise(s["zG"][k])        # train_G: info_gan.py:258-260
            self._draw_info_noise(s["zQ"][k])        # train_Q: info_gan.py:283-284
            return
        self._noise(s["zG"][k], "normal")            # ns_gan.py:208

    def _noise(self, dst, kind):
        """One noise tensor of the GLOBAL batch.  A single process draws it whole; a data-parallel
        rank materialises only its own rows and skips the rest of the stream (draw_rows): the
        per-iteration host cost stays ~constant instead of growing with the number of ranks."""
        if self.world == 1:
            getattr(dst, kind + "_")()
        else:
            r0 = self.rank * self.Bl
            draw_rows(dst, r0, r0 + self.Bl, kind)

    def _fill(self, c0, n_it):
        """HOST: replay the reference's draw order for iterations [c0, c0+n_it) into the pinned host
        ring.  Runs on the prefetch thread while the main thread launches earlier sub-chunks (the C
        replay and torch's RNG kernels release the GIL; the draws stay strictly in order: single
        worker, sub-chunks submitted in order).  The ring slots are free: _pump only submits a fill
        once the launch that last staged them in has completed (_slots_free_now)."""
        s = self._host_views(c0 % self.R)
        if self._replay_ok:
            if HostReplay.run(s["program"], n_it):
                self._gate_np[0] = c0 + n_it         # plain 8-byte store after the ring writes (x86 TSO)
                if self._trace is not None:
                    import time
                    self._trace.append(("filled", n_it, time.perf_counter()))
                return s
            self._replay_ok = False                  # shape outside the restated paths: torch draws
        d = self.D_steps
        for i in range(n_it):
            for j in range(d):
                self._draw_D(s, i * d + j)
            self._draw_G(s, i)
        self._gate_np[0] = c0 + n_it
        return s

    # -- public: run `n_iters` iterations starting a fresh train() ----------------------------
    def configure(self, n_iters, G_lr, D_lr, D_steps, clip=0.0, hyper=(), g_init=0,
                  gp_lambda=10.0, resume=None, extra_config=None):
        """Called once per train(): fresh Adam state (optimizers are locals of the reference's
        train()), schedules, loss buffers, rings, graph.
        resume: optimizer state of a checkpoint (optim_state()) -- the Adam moments are restored
        and the bias-correction schedules continue from the saved step counts instead (an extension:
        the reference saves weights only, ns_gan.py:283-290)."""
        dev = self.device
        self.D_steps, self.clip, self.hyper = D_steps, float(clip), tuple(hyper)
        self.g_off = g_init
        self.gp_lambda = float(gp_lambda)
        self.inv_b = float(np.float32(1.0) / np.float32(self.B))
        self.fG.rebind(); self.fD.rebind()
        self.fG.reset_state(); self.fD.reset_state()
        self.fG.grad.zero_(); self.fD.grad.zero_()
        self.step0 = {"G": 0, "D": 0, "MI": 0}
        self.run_config = {"variant": self.variant, "B": int(self.B), "D_steps": int(D_steps),
                           "G_lr": float(G_lr), "D_lr": float(D_lr), "clip": float(clip),
                           "hyper": [float(h) for h in hyper]}
        # settings of a subclass that the restored state depends on (BEGAN: GAMMA, LAMBDA, patience):
        # part of the strict comparison below, not added after it
        self.run_config.update(extra_config or {})
        if resume is not None:
            saved = resume.get("config")
            if saved is not None and not resume.get("lenient", False):
                diff = {k: (saved[k], self.run_config[k]) for k in self.run_config
                        if k in saved and saved[k] != self.run_config[k]}
                if diff:
                    raise GMError("checkpoint was written by a run with different settings (saved, now): "
                                  "%s -- Adam moments and the bias-correction schedule would not "
                                  "continue that run; load_checkpoint(path, strict=False) overrides" % diff)
            for net, fp in (("G", self.fG), ("D", self.fD)):
                st = resume[net]
                if st["m"].numel() != fp.m.numel():
                    raise GMError("checkpoint optimizer state does not match this model")
                fp.m.copy_(st["m"]); fp.v.copy_(st["v"])
                self.step0[net] = int(st["step"])
        self.steps_planned = {"G": n_iters + g_init, "D": n_iters * D_steps}
        # schedule / loss buffers keep their ADDRESSES across train() calls (grow-only capacity): a
        # second train() with the same settings replays the captured graphs instead of re-capturing
        # them (6 graph sizes x 11..35 kernels: 15-40 ms per call, measured through Trainer.train)
        self._moved = False
        self.schedD = self._pbuf("schedD", ops.adam_schedule(D_lr, max(1, n_iters * D_steps),
                                                             start=self.step0["D"] + 1))
        self.schedG = self._pbuf("schedG", ops.adam_schedule(G_lr, n_iters + g_init,
                                                             start=self.step0["G"] + 1))
        self.lossD = self._pbuf("lossD", max(1, n_iters * D_steps))
        self.lossG = self._pbuf("lossG", n_iters + g_init)
        if self.variant == "info":
            self.fQ.rebind(); self.fQ.reset_state(); self.fQ.grad.zero_()
            self.mi_m.zero_(); self.mi_v.zero_()
            if resume is not None:
                # info_gan.py:146-148: the MI optimizer keeps its OWN moments for G's parameters and Q's
                mi = resume["MI"]
                self.mi_m.copy_(mi["m_G"]); self.mi_v.copy_(mi["v_G"])
                self.fQ.m.copy_(mi["m_Q"]); self.fQ.v.copy_(mi["v_Q"])
                self.step0["MI"] = int(mi["step"])
            self.schedMI = self._pbuf("schedMI", ops.adam_schedule(G_lr, max(1, n_iters),
                                                                   start=self.step0["MI"] + 1))
            self.lossMI = self._pbuf("lossMI", max(1, n_iters))
        self.aux.zero_()
        if resume is not None and self.variant == "fisher":
            self.aux.copy_(resume["fisher_aux"])     # lambda (fisher_gan.py:117-118,155-156) + moments
        self.ctr.zero_()
        self._drain()
        from collections import deque
        self.n_planned = n_iters
        self._fills, self._cursor, self._next_it = deque(), 0, 0
        self._u_copied = 0
        self._launched, self._ramp = deque(), []
        if self._gate is not None:
            torch.cuda.synchronize(self.device)      # no stage-in of an earlier run may still be waiting
            self._gate_np[:] = 0
        torch.cuda.synchronize(self.device)
        import os
        self._trace = [] if os.environ.get("GM_TRACE_RUN") == "1" else None
        self._event_pool = []
        import os
        self.D_steps = D_steps
        R, self.graph_iters = self._ring_and_graph_size(D_steps)
        # graphs larger than the run are never launched: not captured either (a later, longer train() recaptures)
        gmax = 1
        while gmax * 2 <= min(self.graph_iters, max(1, n_iters)):
            gmax *= 2
        self._graph_max = self.graph_iters if self.graph_iters <= n_iters else gmax
        key = (D_steps, R, self.clip, self.hyper, g_init, self.gp_lambda, self.use_graph,
               self.fuse_head, self.fuse_adam, self.fold_tick, self._batch_gen(), self.gated, self._graph_max)
        if getattr(self, "_ring_key", None) != (D_steps, R, self._batch_gen()):
            self._alloc_rings(R)
            self._ring_key = (D_steps, R, self._batch_gen())
            self._graph_key = None
        if self._moved:
            self._graph_key = None                   # a buffer was (re)allocated: captured addresses are stale
        self._key = key

    def _pbuf(self, name, init):
        """Persistent device buffer `name`: init is a host float array (copied in) or an element count
        (zeroed).  The allocation is reused while it is large enough -- its address is captured in the
        graphs -- and `_moved` is raised when it had to be (re)allocated."""
        bufs = self.__dict__.setdefault("_pbufs", {})
        host = None if isinstance(init, int) else torch.from_numpy(np.ascontiguousarray(init, dtype=np.float32))
        n = init if host is None else host.numel()
        cur = bufs.get(name)
        if cur is None or cur.numel() < n:
            # generous floor (256 KB): a later, longer train() on the same engine usually still fits
            cur = torch.zeros(max(n, 1 << 16, 2 * (cur.numel() if cur is not None else 0)), device=self.device)
            bufs[name] = cur
            self._moved = True
        view = cur[:n]
        if host is None:
            view.zero_()
        else:
            view.copy_(host.view(-1))
            view = view.view(host.shape)
        return view

    def _setup_peer_comm(self, sizes):
        """One communicator per optimizer bucket (own flags and sequence numbers); the flat gradient
        buffers are placed inside them.  Every rank runs the self-check; unless ALL ranks pass,
        every rank falls back to RCCL."""
        from . import dp
        comms, why = {}, None
        try:
            for k, n in sizes.items():
                comms[k] = dp.PeerComm(n, self.world, self.rank, self.pg)
            # EVERY communicator's self-check runs on EVERY rank (they are collective kernels: a rank
            # that skipped one would leave its peers in a 10 s bounded wait)
            checks = [c.selfcheck(self.device) for c in comms.values()]
            ok = all(checks)
            self.comm_selfcheck = dict(seconds=round(sum(c.selfcheck_seconds for c in comms.values()), 4),
                                       wait_bound_s=max(c.selfcheck_bound_s for c in comms.values()),
                                       passed={k: bool(v) for k, v in zip(comms, checks)})
            if not ok:
                why = "self-check failed for %s" % [k for k, c in zip(comms, checks) if not c]
        except Exception as e:                       # noqa: BLE001  (no IPC / no peer access here)
            ok, why = False, "%s: %s" % (type(e).__name__, e)
        if self.world > 1:
            import torch.distributed as dist
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
            if dist.get_backend(self.pg) == "nccl":
                flag = flag.to(self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)
            if ok and not bool(flag.item()):
                why = "another rank could not use the peer exchange"
            ok = bool(flag.item())
        # what the exchange regions are made of (bench.py reports it in config.launch)
        self.comm_memory = "fine-grained" if comms and all(c.fine_grained for c in comms.values()) else \
            ("coarse-grained (one shared device)" if comms else "none")
        if ok:
            self._comms = comms
        else:
            import warnings
            for c in comms.values():                 # nothing stays mapped on the fallback path
                try:
                    c.close()
                except Exception:                    # noqa: BLE001
                    pass
            self.comm_mode = "rccl"
            self.comm_fallback = why or "unknown"
            warnings.warn("generative_models_amd: in-graph peer gradient exchange unavailable (%s); "
                          "falling back to host-launched RCCL all-reduces" % self.comm_fallback)

    def _setup_rccl_graph_comm(self):
        """A communicator of the library's own whose all-reduce can be captured; any failure (no RCCL in the process,
        a rank that cannot initialise) leaves every rank on the host-launched path -- agreed on collectively."""
        from . import dp
        ok, why = True, None
        try:
            self._rccl = dp.RcclGraphComm(self.world, self.rank, self.pg)
        except Exception as e:                       # noqa: BLE001
            ok, why, self._rccl = False, "%s: %s" % (type(e).__name__, e), None
        if self.world > 1:
            import torch.distributed as dist
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
            if dist.get_backend(self.pg) == "nccl":
                flag = flag.to(self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)
            if not bool(flag.item()) and self._rccl is not None:
                self._rccl.close()
                self._rccl = None
        if self._rccl is None and why:
            import warnings
            warnings.warn("generative_models_amd: RCCL all-reduce inside the graph unavailable (%s); host-launched "
                          "collectives between segment graphs" % why)

    def exchange_form(self):
        """Which gradient exchange a step takes: 'none' (one rank), 'one_kernel' / 'one_kernel_push' / 'two_kernels'
        (in-graph peer exchange, csrc/gm_comm.hip), 'rccl_in_graph' or 'rccl' (host-launched) for the fallback."""
        if not self._dp():
            return "none"
        if not self._peer():
            return "rccl_in_graph" if self._rccl_in_graph() else "rccl"
        comms = getattr(self, "_comms", None) or {}
        if any(c.two_kernels for c in comms.values()):
            return "two_kernels"
        return "one_kernel_push" if any(getattr(c, "push", False) for c in comms.values()) else "one_kernel"

    def optim_state(self):
        """Everything the optimizers and controllers carry across steps, after the train() call that
        just finished (checkpointing): Adam moments + step counts per optimizer, InfoGAN's third
        optimizer, Fisher's lambda, BEGAN's K and plateau schedulers, and the run's settings."""
        torch.cuda.synchronize()
        cpu = lambda t: t.detach().cpu().clone()
        st = {net: {"m": cpu(fp.m), "v": cpu(fp.v), "step": self.step0[net] + self.steps_planned[net]}
              for net, fp in (("G", self.fG), ("D", self.fD))}
        st["config"] = dict(self.run_config)
        if self.variant == "info":
            st["MI"] = {"m_G": cpu(self.mi_m), "v_G": cpu(self.mi_v), "m_Q": cpu(self.fQ.m),
                        "v_Q": cpu(self.fQ.v), "step": self.step0["MI"] + self.n_planned}
        if self.variant == "fisher":
            st["fisher_aux"] = cpu(self.aux)
        return st

    def _ensure_graph(self):
        if not self.use_graph or self._graph_key == self._key:
            return
        torch.cuda.synchronize()
        if self._one_graph():
            # graphs of graph_iters, ..., 4, 2, 1 iterations (the device counter advances inside every
            # iteration): any run length is a handful of launches.  Each starts with the stage-in of its own
            # iterations (a one-wave wait on the fill gate + the copy: gm_stage_in_gated).  Two forms
            # that were measured and removed in round 4's clean-up: the iteration as a multi-stream DAG inside the
            # graph (-35 %, profiles/r01_experiments.md) and the stage-in's tail on a forked branch of a long graph
            # (72.4 -> 81.7 us per iteration, r02) -- any parallel branch leaves the runtime's linear-chain path.
            def body(k):
                def fn(st):
                    self._issue_stage_in(st, 0, k)
                    for _ in range(k):
                        self._issue_iteration(st, 0)
                return fn
            self.graph = ops.Graph().capture(body(1))
            self.graphs_by_size = [(1, self.graph)]
            k = 2
            while k <= self._graph_max:
                self.graphs_by_size.insert(0, (k, ops.Graph().capture(body(k))))
                k *= 2
            if self.graphs_by_size[0][0] != self._graph_max and self._graph_max > 1:
                self.graphs_by_size.insert(0, (self._graph_max, ops.Graph().capture(body(self._graph_max))))
            self.seg_graphs = None
        else:
            # data parallel: one hipGraph per segment, RCCL all-reduces launched between them
            segs = self._segments()
            first = segs[0][0]
            segs[0] = (lambda st, it: (self._issue_stage_in(st, it, 1), first(st, it)), segs[0][1])
            self.seg_graphs = [(ops.Graph().capture(lambda st, run=run: run(st, 0)), ar)
                               for run, ar in segs]
        self._graph_key = self._key

    def _drain(self):
        """Wait for host fills still in flight (configure / error paths)."""
        pend = getattr(self, "_fills", None)
        failed = False
        while pend:
            _, _, fut = pend.popleft()
            try:
                fut.result()
            except Exception:                         # noqa: BLE001  (a failed fill of a dead run)
                failed = True
        if getattr(self, "_rng_owned", False):
            self._release_rng()
        if failed and getattr(self, "_native_fill", False):
            from . import _lib
            _lib.load().gm_fill_reset()

    def _pump(self, limit, upto=None):
        """Submit sub-chunks of host draws, in order, never past `limit`, while fewer than
        AHEAD * SUB iterations are submitted-and-unfinished (upto: stop once iteration `upto` is
        covered; those sub-chunks are submitted even if that means waiting for ring slots).  The first
        sub-chunks of a run that starts cold are short (`_ramp`) so that the fill gate of the first
        graph opens after one or two iterations' worth of draws.  Native fill worker: one C call per
        sub-chunk, no Python thread involved; otherwise the prefetch thread runs _fill."""
        unfinished = sum(n for _, n, f in self._fills if not f.done())     # (once: pessimistic inside the loop)
        while self._cursor < limit and (upto is None or self._cursor < upto):
            if unfinished >= self.AHEAD * self.SUB:
                break
            it = self._cursor
            want = self._ramp[0] if self._ramp else self.SUB
            n = min(want, self.SUB, self.R - it % self.R, limit - it)
            # never submit a fill whose ring slots belong to iterations that are not launched /
            # finished yet (a draw-ahead past `end` would otherwise sit in the worker until the next
            # run() -- however long the caller takes between epochs)
            if not self._slots_free_now(it, n, wait=upto is not None):
                break
            if self._native_fill:
                fut = self._submit_native(it, n)
            else:
                fut = _prefetch_pool().submit(self._fill, it, n)
            if self._ramp:
                self._ramp.pop(0)
            self._fills.append((it, n, fut))
            self._cursor += n
            unfinished += n

    def _slots_free_now(self, c0, n, wait):
        """True when the ring slots of [c0, c0+n) can be rewritten (the launch that last staged them in has completed); wait=True blocks for it."""
        need = c0 + n - self.R
        if need <= 0:
            return True
        for it_end, e in self._launched:
            if it_end >= need:
                if not e.query():
                    if not wait:
                        return False
                    e.synchronize()
                while self._launched and self._launched[0][0] < need:
                    self._event_pool.append(self._launched.popleft()[1])
                return True
        if wait:                                      # cannot happen: iterations < c0 + n - R precede every piece being launched
            raise GMError("host ring: iteration %d was never launched" % (need - 1))
        return False

    def _submit_native(self, c0, n):
        from . import _lib
        if not self._rng_owned:
            # the generator state lives in this buffer while jobs are in flight; torch's global
            # generator gets it back in _release_rng
            self._rng_state = torch.get_rng_state()
            self._rng_owned = True
        prog = self._host_views(c0 % self.R)["program"]
        job = _lib.load().gm_fill_submit(self._rng_state.data_ptr(), self._rng_state.numel(), prog, len(prog),
                                         n, self._gate.data_ptr(), c0 + n)
        if job <= 0:
            _lib.check(int(job), "gm_fill_submit")
        return _FillJob(job)

    def _release_rng(self):
        """Hand the generator state back to torch once no draw job is outstanding."""
        if self._rng_owned and not self._fills:
            torch.set_rng_state(self._rng_state)
            self._rng_owned = False

    def _reap(self, block=False, upto=None):
        """Retire finished fills (re-raises what a worker hit).  block: wait for the oldest one;
        upto: wait for every fill that covers iterations < upto."""
        fl = self._fills
        while fl and (block or fl[0][2].done() or (upto is not None and fl[0][0] < upto)):
            fl.popleft()[2].result()
            block = False

    def _plan(self, it, n, cold):
        """Split iterations [it, it+n) into graph launches: powers of two up to graph_iters, never
        across the end of the ring (a stage-in copies contiguous slots), ascending inside each ring
        segment.  cold (nothing drawn ahead): the first piece is at most FIRST_PIECE iterations --
        the GPU idles until its draws exist -- and no piece is more than 4x what precedes it (the
        host draws ~5x faster than the GPU consumes, so later gates are open on arrival)."""
        cap = 1
        while cap * 2 <= min(self.graph_iters, self.R):
            cap *= 2
        out, done = [], 0
        while n > 0:
            seg = min(n, self.R - it % self.R)
            q, r = divmod(seg, cap)
            pieces = [1 << i for i in range(cap.bit_length()) if (r >> i) & 1] + [cap] * q
            if cold:
                fixed = []
                for s_ in pieces:
                    stack = [s_]
                    while stack:
                        x = stack.pop()
                        bound = self.FIRST_PIECE if done == 0 else 4 * done
                        if x > bound and x > 1:
                            stack += [x // 2, x // 2]
                        else:
                            fixed.append(x)
                            done += x
                pieces = fixed
            out += pieces
            it += seg
            n -= seg
        return out

    def _copy_U(self, upto):
        """DRAGAN: host ring -> device ring for the uniforms of iterations [_u_copied, upto) on the copy
        engine (side stream); the launch stream waits for it.  Their draws are complete (the caller
        waited), and their device slots are free: a slot's draws are only made after the launch that
        last read it has completed (_slots_free_now)."""
        if self._u_stream is None:
            self._u_stream = torch.cuda.Stream(device=self.device)
        d, R = self.D_steps, self.R
        with torch.cuda.stream(self._u_stream):
            while self._u_copied < upto:
                r = self._u_copied % R
                m = min(upto - self._u_copied, R - r)
                self.dring["U"][r * d:(r + m) * d].copy_(self.hring["U"][r * d:(r + m) * d], non_blocking=True)
                self._u_copied += m
        ev = torch.cuda.Event()
        ev.record(self._u_stream)
        torch.cuda.current_stream(self.device).wait_event(ev)

    def _check_gate(self):
        if self._gate is not None and self._gate_np[1] != 0:
            raise GMError("a stage-in kernel gave up waiting for the host draws of its iterations "
                          "(%g s): results of this run are invalid" % self.GATE_TIMEOUT_S)

    def _launch(self, it, k):
        """Enqueue iterations [it, it+k) (their ring slots are uploaded)."""
        if self.use_graph and self._one_graph():
            for size, g in self.graphs_by_size:           # largest first: 32, 16, ..., 1 iterations
                while k >= size:
                    g.launch()
                    k -= size
        elif self.use_graph:
            for _ in range(k):
                for g, ar in self.seg_graphs:
                    g.launch()
                    if ar is not None:
                        self._allreduce(ar)           # rccl: host-launched between the segment graphs
        else:
            st = ops.stream_ptr()
            self._issue_stage_in(st, it, k)
            for i in range(k):
                self._issue_iteration(st, it + i)

    def run(self, n_iters, it_start=0, horizon=None):
        """Run iterations [it_start, it_start+n_iters): the run is cut into power-of-two graph launches
        (_plan); the host draws of every piece are SUBMITTED (native fill worker, or the prefetch
        thread) before its graph is enqueued and the graph's stage-in kernel waits for them on the fill
        gate, so launches run ahead of the draws.  The GPU starts after the first one or two
        iterations' worth of draws, not after a whole ring.
        horizon: the host may draw ahead up to this iteration (exclusive) while the caller does
        something else after run() returns (train(): the epoch-end loss read-back); never past
        what configure() planned.  Without it the host draws exactly what this call consumes, so a
        caller that times run() times its draws too."""
        if self._dp() and self.variant in ("ra", "fisher", "dra", "be", "info") and not self._peer():
            raise GMError("%s needs exchanges inside the step (scalar pre-reductions over the global "
                          "batch / a third gradient bucket, SURVEY.md 8e); they run on the in-graph peer "
                          "communicator (GM_DP_COMM=peer), which is not active here" % self.variant)
        if it_start != self._next_it:
            raise GMError("run(): iterations are consumed in order (next is %d, got %d)"
                          % (self._next_it, it_start))
        self._ensure_graph()
        end = it_start + n_iters
        if end > self.n_planned:
            raise GMError("run(): configure() planned %d iterations" % self.n_planned)
        limit = min(self.n_planned, max(end, horizon or 0))
        it = it_start
        cold = not self._fills and self._cursor == it_start       # nothing drawn ahead
        gated = self.gated
        # (a gate that timed out in an earlier run raises HERE, before any fill job of this run owns the generator state:
        # ADVICE r5 -- the early submit below used to sit in front of this check and outside the try block)
        self._check_gate()
        trace = self._trace
        try:
            if cold:
                self._ramp = list(self.RAMP)
                # the first piece's draws go to the fill worker before anything else happens here: the GPU's first gate
                # opens ~30 us after THIS point, everything below (plan, first launch: ~50 us) runs beside the draws
                if gated and self._native_fill and self._early_submit:
                    self._pump(limit, upto=it_start + min(self.FIRST_PIECE, n_iters))
        except BaseException:
            self._drain()
            raise
        if trace is not None:
            import time
            trace.append(("run", it_start, time.perf_counter()))
            tev = [torch.cuda.Event(enable_timing=True)]     # GPU-side completion time of every piece (trace mode only)
            tev[0].record()
        try:
            plan = self._plan(it_start, n_iters, cold)
            if trace is not None:
                trace.append(("plan", len(plan), time.perf_counter()))
            for k in plan:
                # the draws of [it, it+k) must be SUBMITTED before their graph is enqueued; gated: the
                # graph's stage-in kernel waits for them on the fill gate, so the launch (~50 us of
                # host time) overlaps the draws; ungated: wait for them here
                self._reap()
                if trace is not None:
                    trace.append(("reaped", it, time.perf_counter()))
                self._pump(limit, upto=it + k)
                if trace is not None:
                    trace.append(("pumped", it, time.perf_counter()))
                while self._cursor < it + k:
                    self._reap(block=True)
                    self._pump(limit, upto=it + k)
                if not gated:
                    self._reap(upto=it + k)
                    self._pump(limit)                 # further draws overlap the launch below
                if self._u_copy:
                    self._reap(upto=it + k)           # the uniforms must exist before the copy engine moves them
                    self._copy_U(it + k)
                if trace is not None:
                    trace.append(("got", it, time.perf_counter()))
                self._launch(it, k)
                if trace is not None:
                    trace.append(("graph", it, time.perf_counter()))
                ev = self._event_pool.pop() if self._event_pool else torch.cuda.Event()
                ev.record()
                self._launched.append((it + k, ev))
                if gated:
                    self._pump(limit)                 # (gated: the launch itself overlaps this piece's draws)
                if trace is not None:
                    trace.append(("pump2", it, time.perf_counter()))
                if trace is not None:
                    tev.append(torch.cuda.Event(enable_timing=True))
                    tev[-1].record()
                while len(self._launched) > 4 * self.R:          # recycle what nobody waits for
                    self._event_pool.append(self._launched.popleft()[1])
                if trace is not None:
                    trace.append(("launched", it + k, time.perf_counter()))
                it += k
            self._reap(upto=end)                      # a failed draw surfaces here, not as a GPU time-out
            self._pump(limit)
            self._reap()
            self._release_rng()                       # (draws still ahead of `end`: the state stays with them)
            if trace is not None:
                import os
                if os.environ.get("GM_TRACE_NOSYNC") == "1":      # the caller evaluates the events once the GPU is through
                    trace.append(("gpu_piece_events", tev, time.perf_counter()))
                else:
                    tev[-1].synchronize()
                    trace.append(("gpu_piece_ends_us", [round(tev[0].elapsed_time(e) * 1e3, 1) for e in tev[1:]],
                                  time.perf_counter()))
        except BaseException:
            if self._gate is not None:
                self._gate_np[0] = 1 << 62            # open every gate: nothing on the GPU waits for draws that will not come
            self._drain()
            raise
        self._next_it = end

    def g_init_steps(self, n):
        """MMGAN pre-training (mm_gan.py:121-136): process_batch draws + G step, eager."""
        s = self._host_views(0)
        saved_graph, saved_off = self.use_graph, self.g_off
        self.use_graph = False
        self._standalone_G = True
        st = ops.stream_ptr()
        for k in range(n):
            draw_sampler_indices(self.N, self.B, s["idx_np"][0])    # images are unused by train_G
            s["zG"][0].normal_()
            src = s["zG"][0]
            if self.local_rings:                       # the device ring holds this rank's rows only
                src = src[self.rank * self.Bl:(self.rank + 1) * self.Bl]
            self.zG_ring[0].copy_(src)
            self.g_off = k
            self._issue_G(st, 0)
            torch.cuda.synchronize()                 # the host slot is rewritten by the next pre-step
        self.use_graph, self.g_off = saved_graph, saved_off
        self._standalone_G = False

    def mark(self):
        """An event behind everything run() has enqueued so far: losses(after=mark) waits for THAT point only, so a
        caller may enqueue the next epoch before it reads the previous epoch's losses (trainers._train)."""
        ev = torch.cuda.Event()
        ev.record()
        return ev

    def _read_back(self, tensors, after):
        """Device loss slices -> numpy, on a side stream behind `after` (the launch stream is not synchronised: whatever
        was enqueued after the mark keeps running)."""
        if getattr(self, "_rb_stream", None) is None:
            self._rb_stream = torch.cuda.Stream(device=self.device)
            self._rb_event = torch.cuda.Event()
            self._rb_pinned = {}
        out = []
        # The HOST waits for the mark; only then is anything enqueued on the side stream.  (A stream-side wait on an event
        # that is an epoch away sat in the second hardware queue as a pending barrier, and while it did EVERY kernel of
        # the launch stream took ~1 us longer: 128-iteration graphs 8 620 -> 9 670 us, Trainer.train() 68.9 -> 75.7 us
        # per step -- tools/trainer_pipeline_probe.py, profiles/r05_experiments.md section 10.)
        after.synchronize()
        with torch.cuda.stream(self._rb_stream):
            for i, t in enumerate(tensors):
                buf = self._rb_pinned.get(i)
                if buf is None or buf.numel() < t.numel():
                    buf = self._rb_pinned[i] = torch.empty(max(t.numel(), 1024), dtype=t.dtype).pin_memory()
                buf[:t.numel()].copy_(t.reshape(-1), non_blocking=True)
                out.append((buf, t.numel()))
            self._rb_event.record(self._rb_stream)
        self._rb_event.synchronize()
        return [b[:n].numpy().copy() for b, n in out]

    def losses(self, it0, it1, after=None):
        """Per-iteration (G loss, mean D loss over D_steps) like ns_gan.py:142-154.  after: an event from mark() --
        wait for that point instead of synchronising the launch stream (one GPU only)."""
        d = self.D_steps
        lg_t = self.lossG[self.g_off + it0:self.g_off + it1]
        ld_t = self.lossD[it0 * d:it1 * d]
        if after is not None and self.world == 1:
            lg, ld = self._read_back([lg_t, ld_t], after)
            ld = ld.reshape(-1, d)
            self._check_gate()
            G = lg.astype(np.float64).tolist()
            D = ld.astype(np.float64).mean(axis=1).tolist() if d > 1 else ld.astype(np.float64)[:, 0].tolist()
            return G, D
        if self.world > 1:       # per-rank partial means (1/B_global scaling) -> global means
            import torch.distributed as dist
            from . import dp
            if self._comms is not None:
                for c in self._comms.values():        # D, G and InfoGAN's Q bucket
                    c.check()
            cpu = dist.get_backend(self.pg) != "nccl"           # gloo control plane: host tensors
            lg_t, ld_t = (lg_t.cpu() if cpu else lg_t.clone()), (ld_t.cpu() if cpu else ld_t.clone())
            dp.allreduce_sum_(lg_t, self.pg)
            dp.allreduce_sum_(ld_t, self.pg)
        lg = lg_t.cpu().numpy()
        ld = ld_t.cpu().numpy().reshape(-1, d)
        self._check_gate()
        # Python floats as the reference collects them (.item() per step, np.mean over the D steps in
        # float64, ns_gan.py:142-154) -- vectorised: a per-row np.mean cost 1 ms per 196-step epoch
        G = lg.astype(np.float64).tolist()
        D = ld.astype(np.float64).mean(axis=1).tolist() if d > 1 else ld.astype(np.float64)[:, 0].tolist()
        return G, D

    def mi_losses(self, it0, it1, after=None):
        t = self.lossMI[it0:it1]
        if after is not None and self.world == 1:
            return [float(x) for x in self._read_back([t], after)[0]]
        if self.world > 1:                           # per-rank partial means -> global
            import torch.distributed as dist
            from . import dp
            t = t.cpu() if dist.get_backend(self.pg) != "nccl" else t.clone()
            dp.allreduce_sum_(t, self.pg)
        return [float(x) for x in t.cpu().numpy()]


# the engines built on this module's pieces (imported last: they import FlatParams / GANEngine ... from here)
from .vae_engine import AEEngine, BIRVAEEngine, VAEEngine      # noqa: E402,F401
from .began_engine import BEGANEngine                          # noqa: E402,F401
