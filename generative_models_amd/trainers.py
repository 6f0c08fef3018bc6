"""Host-side mirror of the reference's class surface: Generator /
Discriminator / <Name> / <Name>Trainer with the reference's constructor and `train` signatures,
attribute names and state_dict keys -- so `from ns_gan import *` scripts and README-style
subclasses (override train_D / train_G, README.md:29-65) keep working.

Two execution paths, both on the HIP kernels:
  * FAST: stock train_D/train_G/process_batch/compute_noise  -> engine.GANEngine (hipGraph of
    fused kernels, host RNG protocol prefetch, no per-step sync);
  * GENERAL: anything overridden -> the reference's step loop restated over autograd Functions
    backed by the same GEMM kernels (ops.fused_linear) and the flat HIP Adam.
There is no CPU execution path: compute on a non-CUDA tensor raises.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import GMError

EPS = 1e-8


def to_cuda(x):
    """utils.py:10-14."""
    if torch.cuda.is_available():
        x = x.cuda()
    return x


def to_var(x):
    """utils.py:6-8."""
    return to_cuda(x).requires_grad_()


def get_data(BATCH_SIZE=100, root="./data/", n_train=50000, n_val=10000, n_test=10000):
    """utils.py:16-53 without the network download: loads MNIST IDX files from `root` if the user
    supplies them, else a synthetic Bernoulli stand-in with the same shapes/seed protocol."""
    import os
    torch.manual_seed(3435)
    raw = os.path.join(root, "MNIST", "raw", "train-images-idx3-ubyte")
    if os.path.isfile(raw):
        def idx(path, off):
            a = np.fromfile(path, dtype=np.uint8)[off:]
            return a
        tr = idx(raw, 16).reshape(-1, 1, 28, 28).astype(np.float32) / 255.0
        te = idx(os.path.join(root, "MNIST", "raw", "t10k-images-idx3-ubyte"), 16) \
            .reshape(-1, 1, 28, 28).astype(np.float32) / 255.0
        trl = idx(os.path.join(root, "MNIST", "raw", "train-labels-idx1-ubyte"), 8).astype(np.int64)
        tel = idx(os.path.join(root, "MNIST", "raw", "t10k-labels-idx1-ubyte"), 8).astype(np.int64)
        train_img = torch.stack([torch.bernoulli(d) for d in torch.from_numpy(tr)])
        train_label = torch.from_numpy(trl)
        test_img = torch.stack([torch.bernoulli(d) for d in torch.from_numpy(te)])
        test_label = torch.from_numpy(tel)
        val_img, val_label = train_img[-10000:].clone(), train_label[-10000:].clone()
        train_img, train_label = train_img[:-10000], train_label[:-10000]
    else:
        mk = lambda n: torch.bernoulli(torch.full((n, 1, 28, 28), 0.1307))
        train_img, val_img, test_img = mk(n_train), mk(n_val), mk(n_test)
        train_label = torch.zeros(n_train, dtype=torch.int64)
        val_label = torch.zeros(n_val, dtype=torch.int64)
        test_label = torch.zeros(n_test, dtype=torch.int64)
    ds = torch.utils.data.TensorDataset
    dl = lambda d: torch.utils.data.DataLoader(d, batch_size=BATCH_SIZE, shuffle=True)
    return dl(ds(train_img, train_label)), dl(ds(val_img, val_label)), dl(ds(test_img, test_label))


def _lin(layer, x, act):
    if not x.is_cuda:
        raise GMError("generative_models_amd computes on MI355X only: got a %s tensor and there is "
                      "no CPU fallback (move the model and inputs with to_cuda)" % x.device)
    return ops.fused_linear(x, layer.weight, layer.bias, act)


def _stock_module(mod, n_linear=2):
    """True iff `mod` is EXACTLY one of the module classes this package ships (not a user subclass
    with its own forward / extra layers) with the expected nn.Linear children.  The fused engines
    read the first Linear layers and hard-code relu-hidden MLPs; anything the user has changed in
    the model (README.md:29-31 invites editing Generator / Discriminator) must take the general
    autograd path instead of being silently trained as a 2-layer MLP."""
    cls = type(mod)
    if not cls.__dict__.get("_gm_stock_model", False):
        return False
    kids = list(mod.children())
    return len(kids) == n_linear and all(type(k) is nn.Linear for k in kids)


def stock_model(cls):
    """Marks a module class shipped by this package (checked on the class itself, so a subclass
    does not inherit the mark)."""
    cls._gm_stock_model = True
    return cls


class _TwoLayer(nn.Module):
    """relu(first) -> out_act(second); attribute names are the reference's state_dict keys."""
    _names = ("linear", "second")
    _out_act = "sigmoid"

    def _build(self, n_in, n_hidden, n_out):
        setattr(self, self._names[0], nn.Linear(n_in, n_hidden))
        setattr(self, self._names[1], nn.Linear(n_hidden, n_out))

    def forward(self, x):
        h = _lin(getattr(self, self._names[0]), x, "relu")
        return _lin(getattr(self, self._names[1]), h, self._out_act)


@stock_model
class Generator(_TwoLayer):
    """ns_gan.py:35-46."""
    _names = ("linear", "generate")

    def __init__(self, image_size, hidden_dim, z_dim):
        super().__init__()
        self._build(z_dim, hidden_dim, image_size)


@stock_model
class Discriminator(_TwoLayer):
    """ns_gan.py:49-60 (sigmoid output)."""
    _names = ("linear", "discriminate")

    def __init__(self, image_size, hidden_dim, output_dim):
        super().__init__()
        self._build(image_size, hidden_dim, output_dim)


@stock_model
class CriticReLU(Discriminator):
    """w_gp_gan.py:49-62 (ReLU output)."""
    _out_act = "relu"


class GANModel(nn.Module):
    """ns_gan.py:63-74: .G .D .z_dim .shape (+ the constructor arguments as attributes)."""
    _D = Discriminator

    def __init__(self, image_size, hidden_dim, z_dim, output_dim=1):
        super().__init__()
        self.image_size, self.hidden_dim, self.z_dim, self.output_dim = \
            image_size, hidden_dim, z_dim, output_dim
        self.G = Generator(image_size, hidden_dim, z_dim)
        self.D = self._D(image_size, hidden_dim, output_dim)
        self.shape = int(image_size ** 0.5)


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam stand-in for the general path: same math (ops.adam = SURVEY.md 3.5), one
    launch over a private flat copy of the parameters.  It is a torch Optimizer so that
    lr schedulers (BEGAN's ReduceLROnPlateau, be_gan.py:133-136) can drive param_groups[0]['lr'].
    Parameters whose grad is None are skipped like torch does (their moments do not advance):
    handled by keeping one (flat, step) pair per "has grad" pattern is overkill here -- the
    reference never mixes patterns within one optimizer, so a None grad means "all None"."""

    def __init__(self, params, lr, weight_decay=0.0, clamp=0.0):
        params = [p for p in params if p.requires_grad]
        super().__init__(params, dict(lr=lr))
        self.wd, self.clamp = weight_decay, clamp
        self._ps = params
        dev = params[0].device
        n = sum((p.numel() + 3) // 4 * 4 for p in params)
        z = lambda: torch.zeros(n, device=dev)
        self.flat, self.grad, self.m, self.v = z(), z(), z(), z()
        self.offs, o = [], 0
        for p in params:
            self.offs.append(o)
            o += (p.numel() + 3) // 4 * 4
        self.t = 0
        self.steps = [0] * len(params)          # per-parameter step counts (torch keeps one per parameter)

    def zero_grad(self, set_to_none=True):
        for p in self._ps:
            p.grad = None

    @torch.no_grad()
    def step(self, closure=None):
        """One Adam step for every parameter that HAS a gradient; a parameter whose grad is None is
        skipped entirely -- value, moments and its own step count stay put -- like
        torch.optim.Adam does (frozen or unused parameters in user-written train_D / train_G).
        Parameters are updated in runs of neighbours that share a step count: one launch in the
        usual case (every parameter has a gradient on every step)."""
        active = [p.grad is not None for p in self._ps]
        if not any(active):
            return
        for i, (p, o) in enumerate(zip(self._ps, self.offs)):
            if active[i]:
                k = p.numel()
                self.flat[o:o + k].copy_(p.data.reshape(-1))
                self.grad[o:o + k].copy_(p.grad.reshape(-1))
                self.steps[i] += 1
        lr = self.param_groups[0]["lr"]
        ends = self.offs[1:] + [self.flat.numel()]
        i, n = 0, len(self._ps)
        while i < n:
            if not active[i]:
                i += 1
                continue
            j = i
            while j + 1 < n and active[j + 1] and self.steps[j + 1] == self.steps[i]:
                j += 1
            lo, hi = self.offs[i], ends[j]
            sched = torch.from_numpy(ops.adam_schedule(lr, 1, start=self.steps[i])).to(self.flat.device)
            ops.adam(self.flat[lo:hi], self.grad[lo:hi], self.m[lo:hi], self.v[lo:hi], sched,
                     weight_decay=self.wd, clamp=self.clamp)
            i = j + 1
        self.t = max(self.steps)
        for k, (p, o) in enumerate(zip(self._ps, self.offs)):
            if active[k]:
                p.data.copy_(self.flat[o:o + p.numel()].view(p.shape))


class GANTrainer:
    """ns_gan.py:77-226 and siblings.  Subclasses set `variant` and the train() signature."""
    variant = "ns"
    defaults = (2e-4, 2e-4, 1)
    method = None
    _STOCK = ("train_D", "train_G", "process_batch", "compute_noise")

    def __init__(self, model, train_iter, val_iter, test_iter, viz=False):
        self.model = to_cuda(model)
        self.name = model.__class__.__name__
        self.train_iter, self.val_iter, self.test_iter = train_iter, val_iter, test_iter
        self.Glosses, self.Dlosses = [], []
        self.viz = viz
        self.num_epochs = 0
        self._engine = None
        self.use_graph = True

    # ---- reference-visible hooks (general path implementations) --------------------------
    def compute_noise(self, batch_size, z_dim):
        """ns_gan.py:218-220."""
        return to_cuda(torch.randn(batch_size, z_dim))

    def process_batch(self, iterator):
        """ns_gan.py:222-226."""
        images, _ = next(iter(iterator))
        return to_cuda(images.view(images.shape[0], -1))

    def _scores(self, images):
        """(D(x), D(G(z)), G(z)) with the reference's draw/evaluation order per variant."""
        m = self.model
        if self.variant in ("ns", "w", "ls", "wgp"):
            g_out = m.G(self.compute_noise(images.shape[0], m.z_dim))
            return m.D(images), m.D(g_out), g_out
        sx = m.D(images)
        g_out = m.G(self.compute_noise(images.shape[0], m.z_dim))
        return sx, m.D(g_out), g_out

    def train_D(self, images, **kw):
        v = self.variant
        sx, sg, g_out = self._scores(images)
        if v in ("ns", "mm"):
            return torch.sum(-torch.mean(torch.log(sx + EPS) + torch.log(1 - sg + EPS)))
        if v == "w":
            return -1 * torch.mean(sx) + torch.mean(sg)
        if v == "ls":
            a, b = kw.get("a", 0), kw.get("b", 1)
            return 0.50 * torch.mean((sx - b) ** 2) + 0.50 * torch.mean((sg - a) ** 2)
        if v == "ra":
            return -torch.mean(torch.log(torch.sigmoid(sx - sg.mean()) + EPS)
                               + torch.log(torch.sigmoid(1 - sg) + EPS)) / 2
        if v == "f":
            return self.loss_fnc.D_loss(sx, sg)
        if v == "fisher":
            m1x, m1g = sx.mean(), sg.mean()
            m2x, m2g = (sx ** 2).mean(), (sg ** 2).mean()
            omega = 1 - (0.5 * m2x + 0.5 * m2g)
            return -((m1x - m1g) + self.LAMBDA * omega - (self.RHO / 2) * (omega ** 2))
        if v == "dra":
            lam, K, C = kw.get("LAMBDA", 10), kw.get("K", 1), kw.get("C", 1)
            loss = -torch.mean(torch.log(sx + EPS) + torch.log(1 - sg + EPS))
            delta = to_cuda(torch.rand(images.shape[0], 1).expand(images.size()))
            x_hat = to_var(delta * images.data + (1 - delta) *
                           (images.data + C * images.data.std() * to_cuda(torch.rand(images.size()))))
            d_hat = self.model.D(x_hat)
            grads = torch.autograd.grad(outputs=d_hat, inputs=x_hat,
                                        grad_outputs=to_cuda(torch.ones(d_hat.shape)),
                                        create_graph=True, retain_graph=True, only_inputs=True)[0]
            return loss + lam * torch.mean((grads.norm(2, dim=1) - K) ** 2)
        if v == "wgp":
            lam = kw.get("LAMBDA", 10)
            eps = to_var(torch.rand(images.shape[0], 1).expand(images.size()))
            x_hat = eps * images + (1 - eps) * g_out
            d_hat = self.model.D(x_hat)
            grads = torch.autograd.grad(outputs=d_hat, inputs=x_hat,
                                        grad_outputs=to_cuda(torch.ones(d_hat.size())),
                                        create_graph=True, retain_graph=True, only_inputs=True)[0]
            return torch.mean(sg) - torch.mean(sx) + lam * torch.mean((grads.norm(2, dim=1) - 1) ** 2)
        raise NotImplementedError(v)

    def train_G(self, images, **kw):
        v, m = self.variant, self.model
        sg = m.D(m.G(self.compute_noise(images.shape[0], m.z_dim)))
        if v in ("ns", "ra", "dra"):
            return -torch.mean(torch.log(sg + EPS))
        if v == "f":
            return self.loss_fnc.G_loss(sg)
        if v == "fisher":
            return -sg.mean()
        if v == "mm":
            return torch.mean(torch.log((1 - sg) + EPS))
        if v in ("w", "wgp"):
            return -1 * torch.mean(sg)
        if v == "ls":
            return 0.50 * torch.mean((sg - kw.get("c", 1)) ** 2)
        raise NotImplementedError(v)

    # ---- path selection -------------------------------------------------------------------
    def _hook_is_stock(self, name):
        """The hook `name` is the one this package ships (not overridden on the instance or in a
        user subclass)."""
        if name in self.__dict__:
            return False
        for base in type(self).__mro__:
            if name in base.__dict__:              # the class that actually defines the hook
                return bool(base.__dict__.get("_gm_stock_class", False))
        return True

    def _stock(self):
        from .engine import GANEngine
        if self.variant not in GANEngine.SUPPORTED:
            return False
        hooks = self._STOCK + (("clip_D_weights",) if self.variant == "w" else ())
        if not all(self._hook_is_stock(name) for name in hooks):
            return False
        m = self.model
        if not (_stock_module(getattr(m, "G", None)) and _stock_module(getattr(m, "D", None))):
            return False                               # edited / subclassed networks: general path
        if self.variant == "info" and not _stock_module(getattr(m, "Q", None)):
            return False
        it = self.train_iter
        ok = (isinstance(it, torch.utils.data.DataLoader)
              and isinstance(it.dataset, torch.utils.data.TensorDataset)
              and isinstance(it.sampler, torch.utils.data.RandomSampler)
              and it.sampler.generator is None and it.generator is None
              and not it.sampler.replacement and it.num_workers == 0
              and it.batch_size is not None and it.batch_size <= len(it.dataset))
        return bool(ok)

    def _captured_general_ok(self):
        """Only train_D / train_G differ from the stock trainer (README.md:29-65): stock process_batch / compute_noise
        (their draws are what the host replays), stock networks, the reference's loader, one GPU, and a variant whose
        stock hooks draw nothing but compute_noise's randn (captured.VARIANTS)."""
        import os
        from . import captured, dp
        if os.environ.get("GM_CAPTURED_GENERAL", "1") == "0" or not torch.cuda.is_available():
            return False
        if self.variant not in captured.VARIANTS or dp.current()[0] != 1:
            return False
        if not (self._hook_is_stock("process_batch") and self._hook_is_stock("compute_noise")
                and self._hook_is_stock("_after_D_backward")):
            return False
        m = self.model
        if not (_stock_module(getattr(m, "G", None)) and _stock_module(getattr(m, "D", None))):
            return False
        from .engine import HostReplay
        it = self.train_iter
        if not HostReplay.available() or (it.batch_size or 0) * m.z_dim < 16:
            return False
        return bool(isinstance(it, torch.utils.data.DataLoader)
                    and isinstance(it.dataset, torch.utils.data.TensorDataset)
                    and isinstance(it.sampler, torch.utils.data.RandomSampler)
                    and it.sampler.generator is None and it.generator is None
                    and not it.sampler.replacement and it.num_workers == 0
                    and it.batch_size is not None and it.batch_size <= len(it.dataset))

    def _get_captured(self):
        from .captured import CapturedLoop
        it = self.train_iter
        key = (id(it.dataset), it.batch_size)
        if getattr(self, "_captured", None) is None or self._captured_key != key:
            import os
            dev = next(self.model.parameters()).device
            imgs = it.dataset.tensors[0]
            data = imgs.reshape(imgs.shape[0], -1).to(dev, torch.float32).contiguous()
            if os.environ.get("GM_PACKED", "1") != "0" and ops.PackedData.is_binary(data):
                data = ops.PackedData(data)
            self._captured = CapturedLoop(self, data, it.batch_size, dev)
            self._captured_key = key
        return self._captured

    def _get_engine(self):
        from .engine import GANEngine
        it = self.train_iter
        key = (id(it.dataset), it.batch_size, self.method)
        if self._engine is None or self._engine_key != key:
            if not torch.cuda.is_available():
                raise GMError("no MI355X visible: the fused step engine has no CPU fallback")
            dev = next(self.model.parameters()).device
            imgs = it.dataset.tensors[0]
            data = imgs.reshape(imgs.shape[0], -1).to(dev, torch.float32).contiguous()
            from . import dp
            world, rank, group = dp.current()
            self._engine = GANEngine(self.variant, self.model, data, it.batch_size, dev,
                                     method=self.method, use_graph=self.use_graph,
                                     world_size=world, rank=rank, process_group=group,
                                     force_dp=getattr(self, "force_dp", False))
            self._engine_key = key
        return self._engine

    # ---- the step loop (ns_gan.py:94-170) --------------------------------------------------
    def _train(self, num_epochs, G_lr, D_lr, D_steps, clip=0.0, hyper=(), G_init=0, quiet=False,
               train_D_kw=None, train_G_kw=None):
        epoch_steps = int(np.ceil(len(self.train_iter) / D_steps))
        if self._stock():
            eng = self._get_engine()
            eng.configure(num_epochs * epoch_steps, G_lr, D_lr, D_steps, clip=clip, hyper=hyper,
                          g_init=G_init, resume=self.__dict__.pop("_resume_optim", None))
            if G_init > 0:
                eng.g_init_steps(G_init)
            # One GPU, no viz: epoch e+1 is ENQUEUED before epoch e's losses are read back (on a side stream, behind an
            # event at epoch e's end), so the GPU does not idle through the read-back, the progress line and the next
            # run()'s start -- ~0.5 ms per epoch, 3.5 % of a 196-step NSGAN epoch at bs=256.  Same values, same order of
            # the progress lines; nothing between the reference's epochs reads the model (ns_gan.py:158-170 with
            # viz=False).  GM_PIPELINE_EPOCHS=0: read back before the next epoch is enqueued.
            import os
            # A subclass that overrides _end_epoch / _viz_epoch (e.g. to save or read the model per epoch) must see the
            # parameters of THAT epoch's end, not of an epoch already running ahead: no pipelining then.
            pipelined = ((not self.viz) and eng.world == 1 and os.environ.get("GM_PIPELINE_EPOCHS", "1") != "0"
                         and self._hook_is_stock("_end_epoch") and self._hook_is_stock("_viz_epoch"))

            def finish(epoch, it0, mark):
                G_losses, D_losses = eng.losses(it0, it0 + epoch_steps, after=mark)     # one sync per epoch
                if self.variant == "info":
                    self.MIlosses.extend(eng.mi_losses(it0, it0 + epoch_steps, after=mark))
                self._end_epoch(epoch, num_epochs, G_losses, D_losses, quiet)
                self._viz_epoch(epoch)

            pending = None
            for epoch in range(1, num_epochs + 1):
                self.model.train()
                it0 = (epoch - 1) * epoch_steps
                # viz draws from the global generator at every epoch end (ns_gan.py:168,234): the host
                # replay must then not run ahead into the next epoch's draws
                try:
                    eng.run(epoch_steps, it_start=it0,
                            horizon=None if self.viz else num_epochs * epoch_steps)
                except BaseException:
                    if pending is not None:                # the finished epoch's losses are recorded before the error surfaces
                        try:
                            finish(*pending)
                        except Exception:                  # noqa: BLE001
                            pass
                    raise
                mark = eng.mark() if pipelined else None
                if pending is not None:
                    finish(*pending)
                    pending = None
                if pipelined and epoch < num_epochs:
                    pending = (epoch, it0, mark)
                else:
                    finish(epoch, it0, mark)
            return
        # GENERAL path: user-overridden hooks, same loop as the reference
        if self.__dict__.get("_resume_optim") is not None:
            raise GMError("load_checkpoint() restored optimizer state, but this trainer runs the general "
                          "path (overridden hooks / edited networks), whose optimizers start fresh: "
                          "resume is only defined on the fused engine")
        m = self.model
        # WGAN's clamp (w_gan.py:158,241-243) is folded into the Adam kernel unless the user
        # overrides clip_D_weights: then theirs is called after every critic step
        user_clip = clip > 0 and hasattr(self, "clip_D_weights") and not self._hook_is_stock("clip_D_weights")
        # README.md:29-65 -- ONLY train_D / train_G overridden: the stock loop around them keeps the device data path
        # and the two hooks are replayed as one captured graph per iteration (captured.py)
        if G_init == 0 and not user_clip and self._captured_general_ok():
            from .captured import NotCapturable
            cap = self._get_captured()
            cap.configure(num_epochs * epoch_steps, G_lr, D_lr, D_steps, clip, train_D_kw or {}, train_G_kw or {})
            try:
                for epoch in range(1, num_epochs + 1):
                    m.train()
                    it0 = (epoch - 1) * epoch_steps
                    cap.run(epoch_steps, use_graph=self.use_graph)
                    G_losses, D_losses = cap.losses(it0, it0 + epoch_steps)
                    self._end_epoch(epoch, num_epochs, G_losses, D_losses, quiet)
                    self._viz_epoch(epoch)
                return
            except NotCapturable as e:          # raised by the FIRST iteration's transaction only: nothing has changed
                if cap.done:
                    raise
                self._captured_refused = str(e)
        G_opt = FlatAdam(m.G.parameters(), G_lr)
        D_opt = FlatAdam(m.D.parameters(), D_lr, clamp=0.0 if user_clip else clip)
        kwD, kwG = train_D_kw or {}, train_G_kw or {}
        for _ in range(G_init):
            images = self.process_batch(self.train_iter)
            G_opt.zero_grad()
            self.train_G(images, **kwG).backward()
            G_opt.step()
        for epoch in range(1, num_epochs + 1):
            m.train()
            G_losses, D_losses = [], []
            for _ in range(epoch_steps):
                step = []
                for _ in range(D_steps):
                    images = self.process_batch(self.train_iter)
                    D_opt.zero_grad()
                    D_loss = self.train_D(images, **kwD)
                    if isinstance(D_loss, tuple):       # Fisher returns (D_loss, IPM_ratio)
                        D_loss = D_loss[0]
                    D_loss.backward()
                    self._after_D_backward()
                    D_opt.step()
                    if user_clip:
                        self.clip_D_weights(clip)
                    step.append(D_loss.item())
                D_losses.append(np.mean(step))
                G_opt.zero_grad()
                G_loss = self.train_G(images, **kwG)
                G_losses.append(G_loss.item())
                G_loss.backward()
                G_opt.step()
            self._end_epoch(epoch, num_epochs, G_losses, D_losses, quiet)
            self._viz_epoch(epoch)

    # ---- visualisation (ns_gan.py:166-170, 228-281) ---------------------
    viz_dir = None          # default: ../viz/<name>/ like the reference (scripts run from src/)

    def _viz_epoch(self, epoch):
        if self.viz:
            self.generate_images(epoch)
            try:
                import matplotlib.pyplot as plt
                plt.show()
            except Exception:                                # noqa: BLE001
                pass

    def generate_images(self, epoch, num_outputs=36, save=True):
        """ns_gan.py:228-262: a grid of generator samples, saved as ../viz/<name>/reconst_<epoch>.png."""
        from . import viz
        return viz.generate_images(self, epoch, num_outputs, save, self.viz_dir)

    def viz_loss(self):
        """ns_gan.py:264-281."""
        from . import viz
        viz.viz_loss(self)

    def _after_D_backward(self):
        """Fisher GAN's hand-rolled lambda ascent (fisher_gan.py:155-156); no-op otherwise."""
        if self.variant == "fisher":
            self.LAMBDA = self.LAMBDA + self.RHO * self.LAMBDA.grad
            self.LAMBDA = to_var(self.LAMBDA.detach())

    def _end_epoch(self, epoch, num_epochs, G_losses, D_losses, quiet=False):
        self.Glosses.extend(G_losses)
        self.Dlosses.extend(D_losses)
        if not quiet:
            print("Epoch[%d/%d], G Loss: %.4f, D Loss: %.4f"
                  % (epoch, num_epochs, np.mean(G_losses), np.mean(D_losses)))
        self.num_epochs += 1

    def train(self, num_epochs, G_lr=2e-4, D_lr=2e-4, D_steps=1):
        """ns_gan.py:94."""
        self._train(num_epochs, G_lr, D_lr, D_steps)

    # ---- checkpoint surface (ns_gan.py:283-290) ---------------------------------------------
    def save_model(self, savepath):
        torch.save(self.model.state_dict(), savepath)

    # ---- full checkpoint / resume (the reference saves weights only) ----
    def save_checkpoint(self, savepath, collective=True):
        """Weights (same state_dict keys as save_model, loadable by the reference), Adam moments and
        step counts of the last train() call, the global CPU generator's state (= the cursor of the
        sampling / noise protocol) and the loss history.  After load_checkpoint() the next train()
        continues as if the run had never stopped.

        Under data parallelism the call is COLLECTIVE (every rank makes it; rank 0 writes, a failed write
        raises on every rank); `collective=False` is for the `if rank == 0: save_checkpoint(p)` pattern:
        rank 0 writes without synchronising, the other ranks' calls do nothing."""
        _save_checkpoint(self, savepath, tuple(n for n in ("Glosses", "Dlosses", "MIlosses", "K", "num_epochs")
                                               if hasattr(self, n)), collective=collective)

    def load_checkpoint(self, loadpath, strict=True):
        """strict: refuse a checkpoint whose run settings (batch size, D_steps, learning rates ...)
        differ from the next train() call's."""
        _load_checkpoint(self, loadpath, strict)

    def load_model(self, loadpath):
        state = torch.load(loadpath)
        self.model.load_state_dict(state)


GANTrainer._gm_stock_class = True


def stock(cls):
    """Marks a trainer class shipped by this package (fast-path eligible when not overridden)."""
    cls._gm_stock_class = True
    return cls


# ============================================================================================
# Checkpoint / resume shared by all trainers
# ============================================================================================
CHECKPOINT_VERSION = 2


def _plain(v):
    """History values as plain Python numbers / lists (so the file loads with weights_only=True)."""
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    if torch.is_tensor(v) and v.dim() == 0:
        return v.item()
    return v


def _save_checkpoint(trainer, savepath, history, numpy_rng=False, collective=True):
    eng = getattr(trainer, "_engine", None)
    if eng is None or not hasattr(eng, "steps_planned"):
        raise GMError("save_checkpoint needs a finished train() call on the fused engine")
    state = {"version": CHECKPOINT_VERSION, "name": trainer.name,
             "model": {k: v.detach().cpu() for k, v in trainer.model.state_dict().items()},
             "optim": eng.optim_state(), "rng": torch.get_rng_state(),
             "history": {n: _plain(getattr(trainer, n)) for n in history}}
    if numpy_rng:
        # bir_vae.py:92-94 draws its reparameterisation noise from NUMPY's global generator: its
        # state is part of the protocol cursor (stored as plain tensors / numbers: weights_only load)
        from .engine import NumpyReplay
        with NumpyReplay.STATE_LOCK:                 # never in the middle of a draw-ahead's read-advance-write
            kind, keys, pos, has_gauss, cached = np.random.get_state()
        if kind != "MT19937":
            raise GMError("numpy's global generator is not the legacy MT19937 one")
        state["numpy_rng"] = {"keys": torch.from_numpy(keys.astype(np.int64)), "pos": int(pos),
                              "has_gauss": int(has_gauss), "cached": float(cached)}
    from . import dp
    world, rank, group = dp.current()
    if world == 1 or not collective:
        if rank == 0:                            # data parallel: replicas are identical, rank 0 writes
            torch.save(state, savepath)
        return
    # COLLECTIVE under data parallelism: every rank of the trainer's group must make this call.  Rank 0 writes; its
    # outcome is broadcast, so nobody loads a half-written file, a failed write (disk full, bad path) raises on
    # EVERY rank instead of leaving the others in a barrier, and `if rank == 0: save_checkpoint(p)` -- a call only
    # rank 0 makes -- has the documented escape `collective=False`.
    import torch.distributed as dist
    outcome = [None]
    if rank == 0:
        try:
            torch.save(state, savepath)
        except Exception as e:                   # noqa: BLE001 -- reported on every rank below
            outcome[0] = "%s: %s" % (type(e).__name__, e)
    dist.broadcast_object_list(outcome, src=0, group=group)
    if outcome[0] is not None:
        raise GMError("save_checkpoint(%r) failed on rank 0: %s" % (savepath, outcome[0]))


def _load_checkpoint(trainer, loadpath, strict=True):
    # plain tensors / numbers / containers only: no pickled code is executed
    ck = torch.load(loadpath, weights_only=True)
    if not isinstance(ck, dict) or "name" not in ck or "optim" not in ck:
        raise GMError("%s is not a checkpoint written by save_checkpoint()" % loadpath)
    if ck.get("version") != CHECKPOINT_VERSION:
        raise GMError("checkpoint format version %r, this build reads version %d -- re-save it with the "
                      "build that wrote it, or keep only the weights (ck['model'])"
                      % (ck.get("version"), CHECKPOINT_VERSION))
    if ck.get("name") != trainer.name:
        raise GMError("checkpoint of a %s trainer, not of a %s trainer" % (ck.get("name"), trainer.name))
    trainer.model.load_state_dict(ck["model"])
    for n, v in ck["history"].items():
        setattr(trainer, n, list(v) if isinstance(v, list) else v)
    trainer._resume_optim = dict(ck["optim"], lenient=not strict)   # consumed by the next train()
    if "numpy_rng" in ck:
        n = ck["numpy_rng"]
        np.random.set_state(("MT19937", n["keys"].numpy().astype(np.uint32), int(n["pos"]),
                             int(n["has_gauss"]), float(n["cached"])))
    torch.set_rng_state(ck["rng"])               # LAST: construction / loading drew nothing after


# ============================================================================================
# VAE (vae.py:47-223)
# ============================================================================================
@stock_model
class Encoder(nn.Module):
    """vae.py:47-61."""

    def __init__(self, image_size, hidden_dim, z_dim):
        super().__init__()
        self.linear = nn.Linear(image_size, hidden_dim)
        self.mu = nn.Linear(hidden_dim, z_dim)
        self.log_var = nn.Linear(hidden_dim, z_dim)

    def forward(self, x):
        h = _lin(self.linear, x, "relu")
        return _lin(self.mu, h, "id"), _lin(self.log_var, h, "id")


@stock_model
class Decoder(_TwoLayer):
    """vae.py:64-77."""
    _names = ("linear", "recon")

    def __init__(self, z_dim, hidden_dim, image_size):
        super().__init__()
        self._build(z_dim, hidden_dim, image_size)

    def forward(self, z):
        return super().forward(z)


@stock_model
class VAE(nn.Module):
    """vae.py:80-106."""

    def __init__(self, image_size=784, hidden_dim=400, z_dim=20):
        super().__init__()
        self.image_size, self.hidden_dim, self.z_dim = image_size, hidden_dim, z_dim
        self.encoder = Encoder(image_size=image_size, hidden_dim=hidden_dim, z_dim=z_dim)
        self.decoder = Decoder(z_dim=z_dim, hidden_dim=hidden_dim, image_size=image_size)
        self.shape = int(image_size ** 0.5)

    def forward(self, x):
        mu, log_var = self.encoder(x)
        z = self.reparameterize(mu, log_var)
        return self.decoder(z), mu, log_var

    def reparameterize(self, mu, log_var):
        epsilon = to_cuda(torch.randn(mu.shape))
        return mu + epsilon * torch.exp(log_var / 2)


def _epoch_order(loader):
    """The permutation one `for batch in loader` pass uses, drawn exactly like the reference's
    DataLoader does: base seed at iterator creation (dataloader.py:706-710), sampler seed at the
    first next() (sampler.py:163-165), then randperm(n) on a private generator."""
    torch.empty((), dtype=torch.int64).random_()
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(len(loader.dataset), generator=g)


@stock
class VAETrainer:
    """vae.py:109-223."""
    _gm_stock_class = True
    _hook_names = ("compute_batch", "kl_divergence", "evaluate")

    def __init__(self, model, train_iter, val_iter, test_iter, viz=False):
        self.model = to_cuda(model)
        self.name = model.__class__.__name__
        self.train_iter, self.val_iter, self.test_iter = train_iter, val_iter, test_iter
        self.best_val_loss = 1e10
        self.debugging_image, _ = next(iter(test_iter))          # vae.py:120 (consumes RNG)
        self.viz = viz
        self.kl_loss, self.recon_loss = [], []
        self.num_epochs = 0
        self._engine = None
        self.use_graph = True

    def compute_batch(self, batch):
        """vae.py:193-208 (general path: autograd over the fused linear kernels)."""
        images, _ = batch
        images = to_cuda(images.view(images.shape[0], -1))
        outputs, mu, log_var = self.model(images)
        recon_loss = torch.sum((images - outputs) ** 2)
        return recon_loss, self.kl_divergence(mu, log_var)

    def kl_divergence(self, mu, log_var):
        """vae.py:210-212."""
        return torch.sum(0.5 * (mu ** 2 + torch.exp(log_var) - log_var - 1))

    def evaluate(self, iterator):
        """vae.py:214-223."""
        loss = []
        for batch in iterator:
            r, k = self.compute_batch(batch)
            loss.append((r + k).item())
        return np.mean(loss)

    def _loader_ok(self, it):
        return (isinstance(it, torch.utils.data.DataLoader)
                and isinstance(it.dataset, torch.utils.data.TensorDataset)
                and isinstance(it.sampler, torch.utils.data.RandomSampler)
                and it.sampler.generator is None and it.generator is None
                and not it.sampler.replacement and it.num_workers == 0 and not it.drop_last
                and it.batch_size is not None)

    def _stock(self):
        cls = type(self)
        for name in self._hook_names:
            if name in self.__dict__:
                return False
            for base in cls.__mro__:
                if name in base.__dict__:
                    if not base.__dict__.get("_gm_stock_class", False):
                        return False
                    break
        m = self.model
        enc, dec = getattr(m, "encoder", None), getattr(m, "decoder", None)
        n_enc = 3 if isinstance(enc, Encoder) else 1       # VAE: linear, mu, log_var; AE: one layer
        n_dec = 2 if isinstance(dec, Decoder) else 1
        if not (_stock_module(enc, n_enc) and _stock_module(dec, n_dec)):
            return False                               # edited / subclassed networks: general path
        if not type(m).__dict__.get("_gm_stock_model", False):
            return False                               # a subclass may have changed forward / reparameterize
        return (self._loader_ok(self.train_iter) and self._loader_ok(self.val_iter)
                and self.train_iter.batch_size == self.val_iter.batch_size)

    def _device_data(self, loader):
        cache = self.__dict__.setdefault("_data_cache", {})
        key = id(loader.dataset)
        if key not in cache:
            import os
            imgs = loader.dataset.tensors[0]
            dev = next(self.model.parameters()).device
            flat = imgs.reshape(imgs.shape[0], -1)
            if os.environ.get("GM_PACKED", "1") != "0" and ops.PackedData.is_binary(flat):
                cache[key] = ops.PackedData(flat.to(dev))        # 1 bit / pixel
            else:
                cache[key] = flat.to(dev, torch.float32).contiguous()
        return cache[key]

    def train(self, num_epochs, lr=1e-3, weight_decay=1e-5, quiet=False):
        """vae.py:127-191."""
        from copy import deepcopy
        if self._stock():
            if not torch.cuda.is_available():
                raise GMError("no MI355X visible: the fused step engine has no CPU fallback")
            from .engine import VAEEngine
            dev = next(self.model.parameters()).device
            if self._engine is None:
                from . import dp
                world, rank, group = dp.current()
                self._engine = VAEEngine(self.model, dev, use_graph=self.use_graph, world_size=world,
                                         rank=rank, process_group=group,
                                         force_dp=getattr(self, "force_dp", False))
            eng = self._engine
            eng.use_graph = self.use_graph
            B = self.train_iter.batch_size
            steps = len(self.train_iter)
            eng.configure(B, num_epochs * steps, lr, weight_decay,
                          resume=self.__dict__.pop("_resume_optim", None))
            tdata, vdata = self._device_data(self.train_iter), self._device_data(self.val_iter)
            nval = len(self.val_iter)
            eng.alloc_val(nval)
            for epoch in range(1, num_epochs + 1):
                self.model.train()
                t0 = (epoch - 1) * steps
                eng.run_pass(tdata, _epoch_order(self.train_iter), True, t0)
                self.model.eval()
                eng.run_pass(vdata, _epoch_order(self.val_iter), False, 0)
                recon = [float(x) for x in eng.read_losses(eng.recon, t0, steps)]     # one sync
                kl = [float(x) for x in eng.read_losses(eng.kl, t0, steps)]
                vr, vk = eng.read_losses(eng.vrecon, 0, nval), eng.read_losses(eng.vkl, 0, nval)
                val_loss = np.mean([float(a + b) for a, b in zip(vr, vk)])
                self._end_epoch(epoch, num_epochs, recon, kl, val_loss, deepcopy, quiet)
            return
        # GENERAL path (compute_batch / evaluate overridden)
        opt = FlatAdam(self.model.parameters(), lr, weight_decay=weight_decay)
        for epoch in range(1, num_epochs + 1):
            self.model.train()
            recon, kl = [], []
            for batch in self.train_iter:
                opt.zero_grad()
                r, k = self.compute_batch(batch)
                (r + k).backward()
                opt.step()
                recon.append(r.item())
                kl.append(k.item())
            self.model.eval()
            val_loss = self.evaluate(self.val_iter)
            self._end_epoch(epoch, num_epochs, recon, kl, val_loss, deepcopy, quiet)

    def _end_epoch(self, epoch, num_epochs, recon, kl, val_loss, deepcopy, quiet):
        self.kl_loss.extend(kl)
        self.recon_loss.extend(recon)
        if val_loss < self.best_val_loss:
            self.best_model = deepcopy(self.model)
            self.best_val_loss = val_loss
        if not quiet:
            tot = [float(np.float32(a) + np.float32(b)) for a, b in zip(recon, kl)]
            print("Epoch[%d/%d], Total Loss: %.4f, Reconst Loss: %.4f, KL Div: %.7f, Val Loss: %.4f"
                  % (epoch, num_epochs, np.mean(tot), np.mean(recon), np.mean(kl), val_loss))
        self.num_epochs += 1
        self._viz_epoch(epoch)

    # ---- visualisation hooks (vae.py:189-191, 225-362) ------------------
    viz_dir = None          # default: ../viz/<name>/ like the reference (scripts run from src/)

    def _viz_epoch(self, epoch):
        if self.viz:
            self.sample_images(epoch)               # vae.py:190: one randn(36, z_dim) from the global generator

    def sample_images(self, epoch=-100, num_images=36, save=True):
        from . import viz
        return viz.vae_sample_images(self, epoch, num_images, save, self.viz_dir)

    def reconstruct_images(self, images, epoch, save=True):
        from . import viz
        return viz.vae_reconstruct_images(self, images, epoch, save, self.viz_dir)

    def sample_interpolated_images(self):
        from . import viz
        return viz.vae_sample_interpolated_images(self)

    def explore_latent_space(self, num_epochs=3):
        """vae.py:295-334: train a 2-latent model of the same family on get_data(), return it (its
        variational means / a 10 x 10 decoded grid are what the reference plots)."""
        train_iter, val_iter, test_iter = get_data()
        latent_model = type(self.model)(image_size=784, hidden_dim=400, z_dim=2)
        latent_space = type(self)(latent_model, train_iter, val_iter, test_iter)
        latent_space.train(num_epochs)
        latent_model = latent_space.best_model
        mu = torch.stack([torch.FloatTensor([m1, m2]) for m1 in np.linspace(-2, 2, 10)
                          for m2 in np.linspace(-2, 2, 10)])
        with torch.no_grad():
            self.latent_grid = latent_model.decoder(to_cuda(mu)).detach().cpu()
        return latent_model

    def make_all(self):
        """vae.py:336-346."""
        print("Sampled images from latent space:")
        self.sample_images(save=False)
        print("Interpolating between two randomly sampled")
        self.sample_interpolated_images()
        print("Exploring latent representations")
        _ = self.explore_latent_space()

    def viz_loss(self):
        from . import viz
        viz.vae_viz_loss(self, "kl_loss")

    def save_model(self, savepath):
        torch.save(self.model.state_dict(), savepath)

    def load_model(self, loadpath):
        self.model.load_state_dict(torch.load(loadpath))

    def save_checkpoint(self, savepath, collective=True):
        """See GANTrainer.save_checkpoint; collective under data parallelism."""
        hist = tuple(n for n in ("recon_loss", "kl_loss", "num_epochs", "best_val_loss")
                     if hasattr(self, n))
        _save_checkpoint(self, savepath, hist, collective=collective)

    def load_checkpoint(self, loadpath, strict=True):
        """strict: refuse a checkpoint whose run settings (batch size, D_steps, learning rates ...)
        differ from the next train() call's."""
        _load_checkpoint(self, loadpath, strict)


# ============================================================================================
# BIR-VAE (bir_vae.py:37-232; SURVEY.md 8f item 2, second half) -- exported by src/bir_vae.py as
# Encoder / Decoder / BIRVAE / BIRVAETrainer.  Fused path: engine.BIRVAEEngine.
# ============================================================================================
@stock_model
class BIREncoder(nn.Module):
    """bir_vae.py:37-51: 784 -> 400 (relu) -> mu."""

    def __init__(self, image_size, hidden_dim, z_dim):
        super().__init__()
        self.linear = nn.Linear(image_size, hidden_dim)
        self.mu = nn.Linear(hidden_dim, z_dim)

    def forward(self, x):
        return _lin(self.mu, _lin(self.linear, x, "relu"), "id")


@stock_model
class BIRDecoder(_TwoLayer):
    """bir_vae.py:54-66."""
    _names = ("linear", "recon")

    def __init__(self, z_dim, hidden_dim, image_size):
        super().__init__()
        self._build(z_dim, hidden_dim, image_size)

    def forward(self, z):
        return super().forward(z)


@stock_model
class BIRVAE(nn.Module):
    """bir_vae.py:69-97.  I: how many bits are let through; set_var = 1 / 4**(I / z_dim)."""

    def __init__(self, image_size=784, hidden_dim=400, z_dim=20, I=13.3):
        super().__init__()
        self.image_size, self.hidden_dim, self.z_dim, self.I = image_size, hidden_dim, z_dim, I
        self.encoder = BIREncoder(image_size=image_size, hidden_dim=hidden_dim, z_dim=z_dim)
        self.decoder = BIRDecoder(z_dim=z_dim, hidden_dim=hidden_dim, image_size=image_size)
        self.shape = int(image_size ** 0.5)
        self.set_var = 1 / (4 ** (I / z_dim))

    def forward(self, x):
        mu = self.encoder(x)
        z = self.reparameterize(mu)
        return self.decoder(z), z

    def reparameterize(self, mu):
        """bir_vae.py:86-97: eps from NUMPY's global RNG with scale = set_var (the reference passes a
        variance as the standard deviation; part of the contract)."""
        eps = to_cuda(torch.from_numpy(np.random.normal(loc=0.0, scale=self.set_var,
                                                        size=tuple(mu.shape))).float())
        return mu + eps


@stock
class BIRVAETrainer(VAETrainer):
    """bir_vae.py:99-232: the VAE loop with loss = sum (x - x_hat)^2 + 1000 * MMD(z)."""
    _gm_stock_class = True
    _hook_names = ("compute_batch", "evaluate", "maximum_mean_discrepancy", "compute_kernel")

    def __init__(self, model, train_iter, val_iter, test_iter, viz=False):
        self.model = to_cuda(model)
        self.name = model.__class__.__name__
        self.train_iter, self.val_iter, self.test_iter = train_iter, val_iter, test_iter
        self.best_val_loss = 1e10
        self.debugging_image, _ = next(iter(test_iter))          # bir_vae.py:110 (consumes RNG)
        self.viz = viz
        self.mmd_loss, self.recon_loss = [], []
        self.num_epochs = 0
        self._engine = None
        self.use_graph = True

    def _stock(self):
        m = self.model
        if not (_stock_module(getattr(m, "encoder", None), 2) and _stock_module(getattr(m, "decoder", None), 2)
                and type(m).__dict__.get("_gm_stock_model", False)):
            return False
        cls = type(self)
        for name in self._hook_names:
            if name in self.__dict__:
                return False
            for base in cls.__mro__:
                if name in base.__dict__:
                    if not base.__dict__.get("_gm_stock_class", False):
                        return False
                    break
        return (self._loader_ok(self.train_iter) and self._loader_ok(self.val_iter)
                and self.train_iter.batch_size == self.val_iter.batch_size)

    # ---- reference-visible hooks (general path) ---------------------------------------------
    def compute_kernel(self, x, y):
        """bir_vae.py:210-221."""
        x_size, y_size, dim = x.size(0), y.size(0), x.size(1)
        tx = x.unsqueeze(1).expand(x_size, y_size, dim)
        ty = y.unsqueeze(0).expand(x_size, y_size, dim)
        return torch.exp(-torch.div(torch.mean(torch.pow(tx - ty, 2), dim=2), dim))

    def maximum_mean_discrepancy(self, z):
        """bir_vae.py:201-208 (the prior sample is drawn on the CPU generator, then moved: the
        reference's own code mixes a CPU tensor with z and only runs on CPU)."""
        x = to_cuda(torch.randn(z.shape))
        return self.compute_kernel(x, x).sum() + self.compute_kernel(z, z).sum() \
            - 2 * self.compute_kernel(x, z).sum()

    def compute_batch(self, batch, LAMBDA=1000.):
        """bir_vae.py:180-199."""
        images, _ = batch
        images = to_cuda(images.view(images.shape[0], -1))
        outputs, z = self.model(images)
        return torch.sum((images - outputs) ** 2), LAMBDA * self.maximum_mean_discrepancy(z)

    def evaluate(self, iterator):
        """bir_vae.py:223-232."""
        loss = []
        for batch in iterator:
            mse, mmd = self.compute_batch(batch)
            loss.append((mse + mmd).item())
        return np.mean(loss)

    def train(self, num_epochs, lr=1e-3, weight_decay=1e-5, quiet=False):
        """bir_vae.py:119-178.  (The reference never increments num_epochs here; neither do we.)"""
        from copy import deepcopy
        if self._stock():
            if not torch.cuda.is_available():
                raise GMError("no MI355X visible: the fused step engine has no CPU fallback")
            from .engine import BIRVAEEngine
            dev = next(self.model.parameters()).device
            if self._engine is None:
                from . import dp
                world, rank, group = dp.current()
                self._engine = BIRVAEEngine(self.model, dev, use_graph=self.use_graph, world_size=world,
                                            rank=rank, process_group=group)
            eng = self._engine
            eng.use_graph = self.use_graph
            steps = len(self.train_iter)
            eng.configure(self.train_iter.batch_size, num_epochs * steps, lr, weight_decay,
                          resume=self.__dict__.pop("_resume_optim", None))
            tdata, vdata = self._device_data(self.train_iter), self._device_data(self.val_iter)
            nval = len(self.val_iter)
            eng.alloc_val(nval)
            for epoch in range(1, num_epochs + 1):
                self.model.train()
                t0 = (epoch - 1) * steps
                eng.run_pass(tdata, _epoch_order(self.train_iter), True, t0)
                self.model.eval()
                eng.run_pass(vdata, _epoch_order(self.val_iter), False, 0)
                recon = [float(x) for x in eng.read_losses(eng.recon, t0, steps)]     # one sync
                mmd = [float(x) for x in eng.read_losses(eng.kl, t0, steps)]
                vr, vm = eng.read_losses(eng.vrecon, 0, nval), eng.read_losses(eng.vkl, 0, nval)
                val_loss = np.mean([float(a + b) for a, b in zip(vr, vm)])
                self._end_epoch_bir(epoch, num_epochs, recon, mmd, val_loss, deepcopy, quiet)
            return
        opt = FlatAdam([p for p in self.model.parameters() if p.requires_grad], lr,
                       weight_decay=weight_decay)
        for epoch in range(1, num_epochs + 1):
            self.model.train()
            recon, mmd = [], []
            for batch in self.train_iter:
                opt.zero_grad()
                a, b = self.compute_batch(batch)
                (a + b).backward()
                opt.step()
                recon.append(a.item())
                mmd.append(b.item())
            self.model.eval()
            val_loss = self.evaluate(self.val_iter)
            self._end_epoch_bir(epoch, num_epochs, recon, mmd, val_loss, deepcopy, quiet)

    def _end_epoch_bir(self, epoch, num_epochs, recon, mmd, val_loss, deepcopy, quiet):
        self.mmd_loss.extend(mmd)
        self.recon_loss.extend(recon)
        if val_loss < self.best_val_loss:
            self.best_model = deepcopy(self.model)
            self.best_val_loss = val_loss
        if not quiet:
            tot = [float(np.float32(a) + np.float32(b)) for a, b in zip(recon, mmd)]
            print("Epoch[%d/%d], Total Loss: %.4f, MSE Loss: %.4f, MMD Loss: %.4f, Val Loss: %.4f"
                  % (epoch, num_epochs, np.mean(tot), np.mean(recon), np.mean(mmd), val_loss))
        self._viz_epoch(epoch)                      # bir_vae.py:176-178

    def viz_loss(self):
        from . import viz
        viz.vae_viz_loss(self, "mmd_loss")

    def save_checkpoint(self, savepath, collective=True):
        """See GANTrainer.save_checkpoint; also carries numpy's global generator state (the
        reparameterisation noise of bir_vae.py:92-94 comes from it)."""
        hist = tuple(n for n in ("recon_loss", "mmd_loss", "num_epochs", "best_val_loss") if hasattr(self, n))
        _save_checkpoint(self, savepath, hist, numpy_rng=True, collective=collective)


# ============================================================================================
# Autoencoder (ae.py:29-205) -- exported by src/ae.py as Encoder / Decoder /
# Autoencoder / AutoencoderTrainer.  Runs on the VAE engine's machinery (engine.AEEngine).
# ============================================================================================
@stock_model
class AEEncoder(nn.Module):
    """ae.py:29-39."""

    def __init__(self, image_size, hidden_dim):
        super().__init__()
        self.linear = nn.Linear(image_size, hidden_dim)

    def forward(self, x):
        return _lin(self.linear, x, "relu")


@stock_model
class AEDecoder(nn.Module):
    """ae.py:42-52."""

    def __init__(self, hidden_dim, image_size):
        super().__init__()
        self.linear = nn.Linear(hidden_dim, image_size)

    def forward(self, encoder_output):
        return _lin(self.linear, encoder_output, "sigmoid")


@stock_model
class Autoencoder(nn.Module):
    """ae.py:55-67."""

    def __init__(self, image_size=784, hidden_dim=32):
        super().__init__()
        self.image_size, self.hidden_dim = image_size, hidden_dim
        self.encoder = AEEncoder(image_size=image_size, hidden_dim=hidden_dim)
        self.decoder = AEDecoder(hidden_dim=hidden_dim, image_size=image_size)

    def forward(self, x):
        return self.decoder(self.encoder(x))


class AutoencoderTrainer(VAETrainer):
    """ae.py:69-205: same loop as the VAE trainer with one loss list (`recon_loss`)."""
    _gm_stock_class = True
    _hook_names = ("compute_batch", "evaluate")

    def __init__(self, model, train_iter, val_iter, test_iter, viz=False):
        self.model = to_cuda(model)
        self.name = model.__class__.__name__
        self.train_iter, self.val_iter, self.test_iter = train_iter, val_iter, test_iter
        self.best_val_loss = 1e10
        self.debugging_image, _ = next(iter(test_iter))          # ae.py:80 (consumes RNG)
        self.viz = viz
        self.recon_loss = []
        self.num_epochs = 0
        self._engine = None
        self.use_graph = True

    def compute_batch(self, batch):
        """ae.py:147-160 (general path: autograd over the fused linear kernels)."""
        images, _ = batch
        images = to_cuda(images.view(images.shape[0], -1))
        return torch.sum((images - self.model(images)) ** 2)

    def evaluate(self, iterator):
        """ae.py:162-164."""
        return np.mean([self.compute_batch(batch).item() for batch in iterator])

    def train(self, num_epochs, lr=1e-3, weight_decay=1e-5, quiet=False):
        """ae.py:87-145."""
        from copy import deepcopy
        if self._stock():
            if not torch.cuda.is_available():
                raise GMError("no MI355X visible: the fused step engine has no CPU fallback")
            from .engine import AEEngine
            dev = next(self.model.parameters()).device
            if self._engine is None:
                from . import dp
                world, rank, group = dp.current()
                self._engine = AEEngine(self.model, dev, use_graph=self.use_graph, world_size=world,
                                        rank=rank, process_group=group,
                                        force_dp=getattr(self, "force_dp", False))
            eng = self._engine
            eng.use_graph = self.use_graph
            steps = len(self.train_iter)
            eng.configure(self.train_iter.batch_size, num_epochs * steps, lr, weight_decay,
                          resume=self.__dict__.pop("_resume_optim", None))
            tdata, vdata = self._device_data(self.train_iter), self._device_data(self.val_iter)
            nval = len(self.val_iter)
            eng.alloc_val(nval)
            for epoch in range(1, num_epochs + 1):
                self.model.train()
                t0 = (epoch - 1) * steps
                eng.run_pass(tdata, _epoch_order(self.train_iter), True, t0)
                self.model.eval()
                eng.run_pass(vdata, _epoch_order(self.val_iter), False, 0)
                recon = [float(x) for x in eng.read_losses(eng.recon, t0, steps)]     # one sync
                val_loss = np.mean([float(x) for x in eng.read_losses(eng.vrecon, 0, nval)])
                self._end_epoch_ae(epoch, num_epochs, recon, val_loss, deepcopy, quiet)
            return
        # GENERAL path (compute_batch / evaluate overridden)
        opt = FlatAdam([p for p in self.model.parameters() if p.requires_grad], lr,
                       weight_decay=weight_decay)
        for epoch in range(1, num_epochs + 1):
            self.model.train()
            recon = []
            for batch in self.train_iter:
                opt.zero_grad()
                loss = self.compute_batch(batch)
                loss.backward()
                opt.step()
                recon.append(loss.item())
            self.model.eval()
            val_loss = self.evaluate(self.val_iter)
            self._end_epoch_ae(epoch, num_epochs, recon, val_loss, deepcopy, quiet)

    def _end_epoch_ae(self, epoch, num_epochs, recon, val_loss, deepcopy, quiet):
        self.recon_loss.extend(recon)
        if val_loss < self.best_val_loss:
            self.best_model = deepcopy(self.model)
            self.best_val_loss = val_loss
        if not quiet:
            print("Epoch[%d/%d], Train Loss: %.4f, Val Loss: %.4f"
                  % (epoch, num_epochs, np.mean(recon), val_loss))
        self.num_epochs += 1
        self._viz_epoch(epoch)

    def _viz_epoch(self, epoch):
        if self.viz:
            self.reconstruct_images(self.debugging_image, epoch)     # ae.py:143-144 (draws nothing)

    def viz_loss(self):
        from . import viz
        viz.vae_viz_loss(self, "none")


# ============================================================================================
# f-GAN divergences (f_gan.py:85-142) -- used by the general path and exposed as a drop-in name
# ============================================================================================
class Divergence:
    METHODS = ("total_variation", "forward_kl", "reverse_kl", "pearson", "hellinger",
               "jensen_shannon")

    def __init__(self, method):
        self.method = method.lower().strip()
        assert self.method in self.METHODS, "Invalid divergence."

    def D_loss(self, DX_score, DG_score):
        m, x, g = self.method, DX_score, DG_score
        if m == "total_variation":
            return -(torch.mean(0.5 * torch.tanh(x)) - torch.mean(0.5 * torch.tanh(g)))
        if m == "forward_kl":
            return -(torch.mean(x) - torch.mean(torch.exp(g - 1)))
        if m == "reverse_kl":
            return -(torch.mean(-torch.exp(x)) - torch.mean(-1 - g))
        if m == "pearson":
            return -(torch.mean(x) - torch.mean(0.25 * g ** 2 + g))
        if m == "hellinger":
            return -(torch.mean(1 - torch.exp(x)) - torch.mean((1 - torch.exp(g)) / torch.exp(g)))
        return -(torch.mean(torch.tensor(2.) - (1 + torch.exp(-x)))
                 - torch.mean(-(torch.tensor(2.) - torch.exp(g))))

    def G_loss(self, DG_score):
        m, g = self.method, DG_score
        if m == "total_variation":
            return -torch.mean(0.5 * torch.tanh(g))
        if m == "forward_kl":
            return -torch.mean(torch.exp(g - 1))
        if m == "reverse_kl":
            return -torch.mean(-1 - g)
        if m == "pearson":
            return -torch.mean(0.25 * g ** 2 + g)
        if m == "hellinger":
            return -torch.mean((1 - torch.exp(g)) / torch.exp(g))
        return -torch.mean(-(torch.tensor(2.) - torch.exp(g)))


# ============================================================================================
# BEGAN (be_gan.py:48-258): autoencoder critic, proportional control of K, plateau schedulers.
# General path only: K and the schedulers are host-side scalars updated from per-step losses
# (be_gan.py:189-195), which is a host sync per step by construction of the algorithm.
# ============================================================================================
@stock_model
class AEDiscriminator(_TwoLayer):
    """be_gan.py:63-76."""
    _names = ("encoder", "decoder")
    _out_act = "id"

    def __init__(self, image_size, hidden_dim):
        super().__init__()
        self._build(image_size, hidden_dim, image_size)


class BEGANModel(nn.Module):
    """be_gan.py:79-90."""

    def __init__(self, image_size, hidden_dim, z_dim):
        super().__init__()
        self.image_size, self.hidden_dim, self.z_dim = image_size, hidden_dim, z_dim
        self.G = Generator(image_size, hidden_dim, z_dim)
        self.D = AEDiscriminator(image_size, hidden_dim)
        self.shape = int(image_size ** 0.5)


class BEGANTrainerBase(GANTrainer):
    variant = "be"
    _gm_stock_class = True

    def _get_engine(self):
        from .engine import BEGANEngine
        it = self.train_iter
        key = (id(it.dataset), it.batch_size)
        if self._engine is None or self._engine_key != key:
            if not torch.cuda.is_available():
                raise GMError("no MI355X visible: the fused step engine has no CPU fallback")
            from . import dp
            world, rank, group = dp.current()
            dev = next(self.model.parameters()).device
            imgs = it.dataset.tensors[0]
            data = imgs.reshape(imgs.shape[0], -1).to(dev, torch.float32).contiguous()
            self._engine = BEGANEngine(self.model, data, it.batch_size, dev, use_graph=self.use_graph,
                                       world_size=world, rank=rank, process_group=group,
                                       force_dp=getattr(self, "force_dp", False))
            self._engine_key = key
        return self._engine

    def train_D(self, images, K):
        """be_gan.py:212-238."""
        m = self.model
        DX_loss = torch.mean(torch.sum(torch.abs(m.D(images) - images), dim=1))
        G_output = m.G(self.compute_noise(images.shape[0], m.z_dim))
        DG_loss = torch.mean(torch.sum(torch.abs(m.D(G_output) - G_output), dim=1))
        return DX_loss - (K * DG_loss), DX_loss, DG_loss

    def train_G(self, images):
        """be_gan.py:240-258."""
        m = self.model
        G_output = m.G(self.compute_noise(images.shape[0], m.z_dim))
        return torch.mean(torch.sum(torch.abs(m.D(G_output) - G_output), dim=1))

    def train(self, num_epochs, G_lr=1e-4, D_lr=1e-4, D_steps=1, GAMMA=0.50, LAMBDA=1e-3, K=0.00,
              quiet=False):
        """be_gan.py:109-210."""
        if self._stock():
            eng = self._get_engine()
            eng.use_graph = self.use_graph
            epoch_steps = int(np.ceil(len(self.train_iter) / D_steps))
            eng.configure(num_epochs * epoch_steps, G_lr, D_lr, D_steps, GAMMA=GAMMA, LAMBDA=LAMBDA,
                          K=K, patience=5 * len(self.train_iter),
                          resume=self.__dict__.pop("_resume_optim", None))
            for epoch in range(1, num_epochs + 1):
                self.model.train()
                it0 = (epoch - 1) * epoch_steps
                eng.run(epoch_steps, it_start=it0,
                        horizon=None if self.viz else num_epochs * epoch_steps)
                G_losses, D_losses = eng.losses(it0, it0 + epoch_steps)
                self.K = eng.K_value()
                self._end_epoch(epoch, num_epochs, G_losses, D_losses, quiet)
                self._viz_epoch(epoch)
            return
        from torch.optim.lr_scheduler import ReduceLROnPlateau
        m = self.model
        G_opt, D_opt = FlatAdam(m.G.parameters(), G_lr), FlatAdam(m.D.parameters(), D_lr)
        pat = 5 * len(self.train_iter)
        G_sch = ReduceLROnPlateau(G_opt, factor=0.50, threshold=0.01, patience=pat)
        D_sch = ReduceLROnPlateau(D_opt, factor=0.50, threshold=0.01, patience=pat)
        epoch_steps = int(np.ceil(len(self.train_iter) / D_steps))
        for epoch in range(1, num_epochs + 1):
            m.train()
            G_losses, D_losses = [], []
            for _ in range(epoch_steps):
                step = []
                for _ in range(D_steps):
                    images = self.process_batch(self.train_iter)
                    D_opt.zero_grad()
                    D_loss, DX_loss, DG_loss = self.train_D(images, K)
                    D_loss.backward()
                    D_opt.step()
                    step.append(D_loss.item())
                D_losses.append(np.mean(step))
                G_opt.zero_grad()
                G_loss = self.train_G(images)
                G_loss.backward()
                G_opt.step()
                G_losses.append(G_loss.item())
                convergence = (DX_loss + torch.abs(GAMMA * DX_loss - DG_loss)).item()
                K_update = (K + LAMBDA * (GAMMA * DX_loss - DG_loss)).item()
                K = min(max(0, K_update), 1)
                D_sch.step(convergence)
                G_sch.step(convergence)
            self.K = K
            self._end_epoch(epoch, num_epochs, G_losses, D_losses, quiet)


# ============================================================================================
# InfoGAN (info_gan.py:45-325).  General path (three optimizers, G updated by two of them).
# ============================================================================================
@stock_model
class InfoGenerator(_TwoLayer):
    _names = ("linear", "generate")

    def __init__(self, image_size, hidden_dim, z_dim, disc_dim, cont_dim):
        super().__init__()
        self._build(z_dim + disc_dim + cont_dim, hidden_dim, image_size)


@stock_model
class InfoDiscriminator(_TwoLayer):
    _names = ("linear", "discriminator")

    def __init__(self, image_size, hidden_dim, output_dim):
        super().__init__()
        self.image_size, self.hidden_dim, self.output_dim = image_size, hidden_dim, output_dim
        self._build(image_size, hidden_dim, output_dim)


@stock_model
class InfoQ(_TwoLayer):
    _names = ("linear", "inference")
    _out_act = "id"

    def __init__(self, image_size, hidden_dim, disc_dim, cont_dim):
        super().__init__()
        self.image_size, self.hidden_dim = image_size, hidden_dim
        self.disc_dim, self.cont_dim = disc_dim, cont_dim
        self._build(image_size, hidden_dim, disc_dim + cont_dim)

    def forward(self, x):
        out = super().forward(x)
        return out[:, :self.disc_dim], out[:, self.disc_dim:]


class InfoGANModel(nn.Module):
    """info_gan.py:97-110."""

    def __init__(self, image_size, hidden_dim, z_dim, disc_dim, cont_dim, output_dim=1):
        super().__init__()
        self.image_size, self.hidden_dim, self.z_dim = image_size, hidden_dim, z_dim
        self.disc_dim, self.cont_dim, self.output_dim = disc_dim, cont_dim, output_dim
        self.G = InfoGenerator(image_size, hidden_dim, z_dim, disc_dim, cont_dim)
        self.D = InfoDiscriminator(image_size, hidden_dim, output_dim)
        self.Q = InfoQ(image_size, hidden_dim, disc_dim, cont_dim)
        self.shape = int(image_size ** 0.5)


class InfoGANTrainerBase(GANTrainer):
    variant = "info"
    _gm_stock_class = True
    _STOCK = GANTrainer._STOCK + ("train_Q",)

    def __init__(self, model, train_iter, val_iter, test_iter, viz=False):
        super().__init__(model, train_iter, val_iter, test_iter, viz)
        self.MIlosses = []

    def compute_noise(self, batch_size, z_dim, disc_dim, cont_dim, c=None):
        """info_gan.py:306-325."""
        z = torch.randn(batch_size, z_dim)
        disc_c = torch.zeros((batch_size, disc_dim))
        if c is not None:
            categorical = int(c) * torch.ones((batch_size,), dtype=torch.long)
        else:
            categorical = torch.randint(0, disc_dim, (batch_size,), dtype=torch.long)
        disc_c[range(batch_size), categorical] = 1
        cont_c = torch.randn(batch_size, cont_dim)
        return to_cuda(torch.cat((z, disc_c, cont_c), dim=1))

    def _noise(self, images):
        m = self.model
        return self.compute_noise(images.shape[0], m.z_dim, m.disc_dim, m.cont_dim)

    def generate_images(self, epoch, num_outputs=36, save=True, c=None):
        """info_gan.py:333-365: c fixes the categorical code of every sample (latent exploration)."""
        from . import viz
        m = self.model
        noise = self.compute_noise(num_outputs, m.z_dim, m.disc_dim, m.cont_dim, c=c)
        return viz.generate_images(self, epoch, num_outputs, save, self.viz_dir, noise=noise)

    def train_D(self, images):
        m = self.model
        sx = m.D(images)
        sg = m.D(m.G(self._noise(images)))
        return torch.sum(-torch.mean(torch.log(sx + EPS) + torch.log(1 - sg + EPS)))

    def train_G(self, images):
        m = self.model
        return -torch.mean(torch.log(m.D(m.G(self._noise(images))) + EPS))

    def train_Q(self, images, LAMBDA=1):
        """info_gan.py:269-304."""
        import torch.nn.functional as F
        m = self.model
        noise = self._noise(images)
        q_disc, q_cont = m.Q(m.G(noise))
        target = noise[:, m.z_dim:m.z_dim + m.disc_dim]
        disc_loss = F.cross_entropy(q_disc, torch.max(target, 1)[1])
        cont_loss = F.mse_loss(q_cont, noise[:, m.z_dim + m.disc_dim:])
        return LAMBDA * (disc_loss + cont_loss)

    def train(self, num_epochs, G_lr=2e-4, D_lr=2e-4, D_steps=1, quiet=False):
        """info_gan.py:130-221."""
        if self._stock():
            return self._train(num_epochs, G_lr, D_lr, D_steps, quiet=quiet)     # fused engine
        m = self.model
        pD, pG, pQ = list(m.D.parameters()), list(m.G.parameters()), list(m.Q.parameters())
        D_opt, G_opt, MI_opt = FlatAdam(pD, D_lr), FlatAdam(pG, G_lr), FlatAdam(pG + pQ, G_lr)
        epoch_steps = int(np.ceil(len(self.train_iter) / D_steps))
        for epoch in range(1, num_epochs + 1):
            m.train()
            G_losses, D_losses, MI_losses = [], [], []
            for _ in range(epoch_steps):
                step = []
                for _ in range(D_steps):
                    images = self.process_batch(self.train_iter)
                    D_opt.zero_grad()
                    D_loss = self.train_D(images)
                    D_loss.backward()
                    D_opt.step()
                    step.append(D_loss.item())
                D_losses.append(np.mean(step))
                G_opt.zero_grad()
                G_loss = self.train_G(images)
                G_losses.append(G_loss.item())
                G_loss.backward()
                G_opt.step()
                MI_opt.zero_grad()
                MI_loss = self.train_Q(images)
                MI_losses.append(MI_loss.item())
                MI_loss.backward()
                MI_opt.step()
            self.MIlosses.extend(MI_losses)
            self._end_epoch(epoch, num_epochs, G_losses, D_losses, quiet)
