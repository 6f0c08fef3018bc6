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


class Generator(_TwoLayer):
    """ns_gan.py:35-46."""
    _names = ("linear", "generate")

    def __init__(self, image_size, hidden_dim, z_dim):
        super().__init__()
        self._build(z_dim, hidden_dim, image_size)


class Discriminator(_TwoLayer):
    """ns_gan.py:49-60 (sigmoid output)."""
    _names = ("linear", "discriminate")

    def __init__(self, image_size, hidden_dim, output_dim):
        super().__init__()
        self._build(image_size, hidden_dim, output_dim)


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


class FlatAdam:
    """torch.optim.Adam stand-in for the general path: same math (ops.adam = SURVEY.md 3.5), one
    launch over a flat buffer.  zero_grad() sets grads to None like torch >= 2.0."""

    def __init__(self, params, lr, weight_decay=0.0, clamp=0.0):
        from .engine import FlatParams
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.wd, self.clamp = lr, weight_decay, clamp
        dev = self.params[0].device
        self.fp = FlatParams(self.params, dev)
        self.t = 0
        self._sched = None

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def step(self):
        fp = self.fp
        fp.rebind()
        for p, g in zip(self.params, fp.gviews):
            if p.grad is None:
                g.zero_()
            elif p.grad.data_ptr() != g.data_ptr():
                g.copy_(p.grad)
        self.t += 1
        sched = torch.from_numpy(ops.adam_schedule(self.lr, 1, start=self.t)).to(fp.flat.device)
        ops.adam(fp.flat, fp.grad, fp.m, fp.v, sched, weight_decay=self.wd, clamp=self.clamp)


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
        if v in ("ns", "ra"):
            return -torch.mean(torch.log(sg + EPS))
        if v == "mm":
            return torch.mean(torch.log((1 - sg) + EPS))
        if v in ("w", "wgp"):
            return -1 * torch.mean(sg)
        if v == "ls":
            return 0.50 * torch.mean((sg - kw.get("c", 1)) ** 2)
        raise NotImplementedError(v)

    # ---- path selection -------------------------------------------------------------------
    def _stock(self):
        from .engine import GANEngine
        if self.variant not in GANEngine.SUPPORTED:
            return False
        cls = type(self)
        for name in self._STOCK:
            if name in self.__dict__:
                return False
            for base in cls.__mro__:
                if name in base.__dict__:          # the class that actually defines the hook
                    if not base.__dict__.get("_gm_stock_class", False):
                        return False
                    break
        it = self.train_iter
        ok = (isinstance(it, torch.utils.data.DataLoader)
              and isinstance(it.dataset, torch.utils.data.TensorDataset)
              and isinstance(it.sampler, torch.utils.data.RandomSampler)
              and it.sampler.generator is None and it.generator is None
              and not it.sampler.replacement and it.num_workers == 0
              and it.batch_size is not None and it.batch_size <= len(it.dataset))
        return bool(ok)

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
            self._engine = GANEngine(self.variant, self.model, data, it.batch_size, dev,
                                     method=self.method, use_graph=self.use_graph)
            self._engine_key = key
        return self._engine

    # ---- the step loop (ns_gan.py:94-170) --------------------------------------------------
    def _train(self, num_epochs, G_lr, D_lr, D_steps, clip=0.0, hyper=(), G_init=0, quiet=False,
               train_D_kw=None, train_G_kw=None):
        epoch_steps = int(np.ceil(len(self.train_iter) / D_steps))
        if self._stock():
            eng = self._get_engine()
            eng.configure(num_epochs * epoch_steps, G_lr, D_lr, D_steps, clip=clip, hyper=hyper,
                          g_init=G_init)
            if G_init > 0:
                eng.g_init_steps(G_init)
            for epoch in range(1, num_epochs + 1):
                self.model.train()
                it0 = (epoch - 1) * epoch_steps
                eng.run(epoch_steps, it_start=it0)
                G_losses, D_losses = eng.losses(it0, it0 + epoch_steps)     # one sync per epoch
                self._end_epoch(epoch, num_epochs, G_losses, D_losses, quiet)
            return
        # GENERAL path: user-overridden hooks, same loop as the reference
        m = self.model
        G_opt = FlatAdam(m.G.parameters(), G_lr)
        D_opt = FlatAdam(m.D.parameters(), D_lr, clamp=clip)
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
                    D_loss.backward()
                    D_opt.step()
                    step.append(D_loss.item())
                D_losses.append(np.mean(step))
                G_opt.zero_grad()
                G_loss = self.train_G(images, **kwG)
                G_losses.append(G_loss.item())
                G_loss.backward()
                G_opt.step()
            self._end_epoch(epoch, num_epochs, G_losses, D_losses, quiet)

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

    def load_model(self, loadpath):
        state = torch.load(loadpath)
        self.model.load_state_dict(state)


GANTrainer._gm_stock_class = True


def stock(cls):
    """Marks a trainer class shipped by this package (fast-path eligible when not overridden)."""
    cls._gm_stock_class = True
    return cls
