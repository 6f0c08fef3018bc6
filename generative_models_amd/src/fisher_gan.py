"""Drop-in for the reference's src/fisher_gan.py: same module-level names, constructor and train()
signatures and state_dict keys (fisher_gan.py:41-248); compute runs on the gfx950 kernels of generative_models_amd."""
import _bootstrap  # noqa: F401
import numpy as np  # noqa: F401
import torch  # noqa: F401
import torch.nn as nn  # noqa: F401

from utils import *  # noqa: F401,F403
from generative_models_amd import trainers as _t
from generative_models_amd.trainers import Generator, Discriminator  # noqa: F401



class FisherGAN(_t.GANModel):
    """fisher_gan.py:41-248"""
    pass

@_t.stock
class FisherGANTrainer(_t.GANTrainer):
    """fisher_gan.py:41-248"""
    variant = "fisher"

    def train(self, num_epochs, G_lr=1e-4, D_lr=1e-4, D_steps=1, RHO=1e-6):
        """fisher_gan.py:101; lambda lives on the device and is updated by the loss kernel
        (:155-156); self.LAMBDA mirrors it after train()."""
        self.LAMBDA = to_var(torch.zeros(1))          # fisher_gan.py:117-118
        self.RHO = to_var(torch.tensor(RHO))
        self._train(num_epochs, G_lr, D_lr, D_steps, hyper=(RHO,))
        if self._engine is not None:
            self.LAMBDA = self._engine.aux[0:1].clone()
