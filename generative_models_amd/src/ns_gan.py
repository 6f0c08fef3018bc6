"""Drop-in for the reference's src/ns_gan.py: same module-level names, constructor and train()
signatures and state_dict keys (ns_gan.py:35-226); compute runs on the gfx950 kernels of generative_models_amd."""
import _bootstrap  # noqa: F401
import numpy as np  # noqa: F401
import torch  # noqa: F401
import torch.nn as nn  # noqa: F401

from utils import *  # noqa: F401,F403
from generative_models_amd import trainers as _t
from generative_models_amd.trainers import Generator, Discriminator  # noqa: F401



class NSGAN(_t.GANModel):
    """ns_gan.py:35-226"""
    pass

@_t.stock
class NSGANTrainer(_t.GANTrainer):
    """ns_gan.py:35-226"""
    variant = "ns"

    def train(self, num_epochs, G_lr=2e-4, D_lr=2e-4, D_steps=1):
        """ns_gan.py:94."""
        self._train(num_epochs, G_lr, D_lr, D_steps)
