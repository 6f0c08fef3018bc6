"""Drop-in for the reference's src/f_gan.py (f_gan.py:42-284): Generator, Discriminator, fGAN,
Divergence, fGANTrainer.train(num_epochs, method, G_lr, D_lr, D_steps)."""
import _bootstrap  # noqa: F401
import numpy as np  # noqa: F401
import torch  # noqa: F401
import torch.nn as nn  # noqa: F401

from utils import *  # noqa: F401,F403
from generative_models_amd import trainers as _t
from generative_models_amd.trainers import Divergence, Generator, Discriminator  # noqa: F401


class fGAN(_t.GANModel):
    """f_gan.py:71-82"""


@_t.stock
class fGANTrainer(_t.GANTrainer):
    """f_gan.py:145-284"""
    variant = "f"

    def train(self, num_epochs, method, G_lr=1e-4, D_lr=1e-4, D_steps=1):
        """f_gan.py:162."""
        self.loss_fnc = Divergence(method)
        self.method = self.loss_fnc.method
        self._train(num_epochs, G_lr, D_lr, D_steps)
