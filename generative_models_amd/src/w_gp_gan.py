"""Drop-in for the reference's src/w_gp_gan.py: same module-level names, constructor and train()
signatures and state_dict keys (w_gp_gan.py:34-239); compute runs on the gfx950 kernels of generative_models_amd."""
import _bootstrap  # noqa: F401
import numpy as np  # noqa: F401
import torch  # noqa: F401
import torch.nn as nn  # noqa: F401

from utils import *  # noqa: F401,F403
from generative_models_amd import trainers as _t
from generative_models_amd.trainers import Generator, Discriminator  # noqa: F401

Discriminator = _t.CriticReLU      # ReLU-output critic (w_gp_gan.py:59-62)


class WGPGAN(_t.GANModel):
    """w_gp_gan.py:34-239"""
    _D = _t.CriticReLU

@_t.stock
class WGPGANTrainer(_t.GANTrainer):
    """w_gp_gan.py:34-239"""
    variant = "wgp"

    def train_D(self, images, LAMBDA=10):
        """w_gp_gan.py:177-220."""
        return super().train_D(images, LAMBDA=LAMBDA)

    def train(self, num_epochs, G_lr=1e-4, D_lr=1e-4, D_steps=5):
        """w_gp_gan.py:96 (LAMBDA=10 is the train_D default, :177)."""
        self._train(num_epochs, G_lr, D_lr, D_steps)
