"""Drop-in for the reference's src/ls_gan.py: same module-level names, constructor and train()
signatures and state_dict keys (ls_gan.py:33-215); compute runs on the gfx950 kernels of generative_models_amd."""
import _bootstrap  # noqa: F401
import numpy as np  # noqa: F401
import torch  # noqa: F401
import torch.nn as nn  # noqa: F401

from utils import *  # noqa: F401,F403
from generative_models_amd import trainers as _t
from generative_models_amd.trainers import Generator, Discriminator  # noqa: F401



class LSGAN(_t.GANModel):
    """ls_gan.py:33-215"""
    pass

@_t.stock
class LSGANTrainer(_t.GANTrainer):
    """ls_gan.py:33-215"""
    variant = "ls"

    def train_D(self, images, a=0, b=1):
        """ls_gan.py:159-180."""
        return super().train_D(images, a=a, b=b)

    def train_G(self, images, c=1):
        """ls_gan.py:182-201."""
        return super().train_G(images, c=c)

    def train(self, num_epochs, G_lr=1e-4, D_lr=1e-4, D_steps=1):
        """ls_gan.py:95 (labels a=0, b=1, c=1: :173,:197)."""
        self._train(num_epochs, G_lr, D_lr, D_steps, hyper=(0.0, 1.0, 1.0))
