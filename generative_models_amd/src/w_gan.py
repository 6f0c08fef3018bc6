"""Drop-in for the reference's src/w_gan.py: same module-level names, constructor and train()
signatures and state_dict keys (w_gan.py:43-243); compute runs on the gfx950 kernels of generative_models_amd."""
import _bootstrap  # noqa: F401
import numpy as np  # noqa: F401
import torch  # noqa: F401
import torch.nn as nn  # noqa: F401

from utils import *  # noqa: F401,F403
from generative_models_amd import trainers as _t
from generative_models_amd.trainers import Generator, Discriminator  # noqa: F401



class WGAN(_t.GANModel):
    """w_gan.py:43-243"""
    pass

@_t.stock
class WGANTrainer(_t.GANTrainer):
    """w_gan.py:43-243"""
    variant = "w"

    def train(self, num_epochs, G_lr=5e-5, D_lr=5e-5, D_steps=5, clip=0.01):
        """w_gan.py:105 (Adam, not RMSprop: the code wins over the docstring; clamp :158)."""
        self._train(num_epochs, G_lr, D_lr, D_steps, clip=clip)

    def clip_D_weights(self, clip):
        """w_gan.py:241-243."""
        for parameter in self.model.D.parameters():
            parameter.data.clamp_(-clip, clip)
