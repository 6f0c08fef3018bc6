"""Drop-in for the reference's src/mm_gan.py: same module-level names, constructor and train()
signatures and state_dict keys (mm_gan.py:35-237); compute runs on the gfx950 kernels of generative_models_amd."""
import _bootstrap  # noqa: F401
import numpy as np  # noqa: F401
import torch  # noqa: F401
import torch.nn as nn  # noqa: F401

from utils import *  # noqa: F401,F403
from generative_models_amd import trainers as _t
from generative_models_amd.trainers import Generator, Discriminator  # noqa: F401



class MMGAN(_t.GANModel):
    """mm_gan.py:35-237"""
    pass

@_t.stock
class MMGANTrainer(_t.GANTrainer):
    """mm_gan.py:35-237"""
    variant = "mm"

    def train(self, num_epochs, G_lr=2e-4, D_lr=2e-4, D_steps=1, G_init=5):
        """mm_gan.py:97 (G_init generator-only pre-steps, :121-136)."""
        self._train(num_epochs, G_lr, D_lr, D_steps, G_init=G_init)
