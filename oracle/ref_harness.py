"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Loads the *unmodified* reference (shayneobrien/generative-models, mounted read-only at
/root/reference) so that its own Trainer classes can be executed on CPU.  It is used for two
things and nothing else:

  * `oracle/gen_golden.py` runs it to produce the committed fixtures in `tests/golden/`;
  * `tests/test_oracle_pin.py` runs it (only where /root/reference exists, i.e. in the build
    container -- never on the GPU box) to pin `oracle/port.py` bit-for-bit.

Shims (none touches arithmetic):
  1. fake `torchvision{,.datasets,.transforms,.utils}` modules (imported at ns_gan.py:19,
     utils.py:2-3, vae.py:33-34; never called on the hot path);
  2. `matplotlib.use('Agg')`;
  3. `/root/reference/src` on sys.path (every file does `from utils import *`, ns_gan.py:32);
  4. w_gan only: `/root/reference` on sys.path (`from src import utils`, w_gan.py:40) and
     `to_cuda`/`get_data` injected into its namespace (NameError at w_gan.py:92 otherwise);
  5. a synthetic TensorDataset instead of `utils.get_data()` (needs a network download).
"""
import contextlib
import importlib
import io
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("GM_REFERENCE_ROOT", "/root/reference")
REFERENCE_SRC = os.path.join(REFERENCE_ROOT, "src")

MODULES = ("ns_gan", "mm_gan", "w_gan", "w_gp_gan", "ls_gan", "dra_gan", "be_gan", "ra_gan",
           "f_gan", "fisher_gan", "info_gan", "vae", "ae", "bir_vae")


def available():
    return os.path.isfile(os.path.join(REFERENCE_SRC, "ns_gan.py"))


def _install_stubs():
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        for sub in ("datasets", "transforms", "utils"):
            m = types.ModuleType("torchvision." + sub)
            setattr(tv, sub, m)
            sys.modules["torchvision." + sub] = m
        tv.transforms.ToTensor = lambda *a, **k: None
        tv.transforms.ToPILImage = lambda *a, **k: None
        tv.utils.make_grid = lambda *a, **k: None
        tv.utils.save_image = lambda *a, **k: None
        sys.modules["torchvision"] = tv
    import matplotlib
    matplotlib.use("Agg")


_loaded = {}


def load(name):
    """Import reference module `name` (e.g. 'ns_gan') unmodified and return it."""
    if name in _loaded:
        return _loaded[name]
    if not available():
        raise RuntimeError("reference not mounted at %s" % REFERENCE_ROOT)
    _install_stubs()
    saved_path = list(sys.path)
    # The reference's module names (utils, vae, ...) are generic: import them under a private
    # alias so they never shadow the product's drop-in modules of the same name.
    saved_mods = {k: sys.modules.get(k) for k in ("utils", "src", "src.utils", name)}
    for k in saved_mods:
        sys.modules.pop(k, None)
    try:
        sys.path.insert(0, REFERENCE_ROOT)
        sys.path.insert(0, REFERENCE_SRC)
        mod = importlib.import_module(name)
        if name == "w_gan":
            ref_utils = importlib.import_module("utils")
            mod.to_cuda = ref_utils.to_cuda
            mod.get_data = ref_utils.get_data
            mod.to_var = ref_utils.to_var
    finally:
        sys.path[:] = saved_path
        for k, v in saved_mods.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v
    _loaded[name] = mod
    return mod


@contextlib.contextmanager
def quiet():
    """Silence the reference's print()/tqdm output."""
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        yield


def synthetic_loaders(batch_size, n_train=50000, n_val=10000, n_test=10000, image_shape=(1, 28, 28),
                      p=0.1307, seed=3435):
    """Synthetic stand-in for utils.get_data (utils.py:16-53): Bernoulli({0,1}) images with the
    MNIST mean intensity, zero labels, three shuffled DataLoaders, seeded like utils.py:18."""
    torch.manual_seed(seed)
    def make(n):
        img = torch.bernoulli(torch.full((n,) + tuple(image_shape), p))
        return torch.utils.data.TensorDataset(img, torch.zeros(n, dtype=torch.int64))
    mk = lambda ds: torch.utils.data.DataLoader(ds, batch_size=batch_size, shuffle=True)
    return mk(make(n_train)), mk(make(n_val)), mk(make(n_test))
