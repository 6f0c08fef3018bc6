"""Drop-in for the reference's src/info_gan.py (info_gan.py:45-325)."""
import _bootstrap  # noqa: F401
import numpy as np  # noqa: F401
import torch  # noqa: F401
import torch.nn as nn  # noqa: F401

from utils import *  # noqa: F401,F403
from generative_models_amd import trainers as _t
from generative_models_amd.trainers import InfoGenerator as Generator  # noqa: F401
from generative_models_amd.trainers import InfoDiscriminator as Discriminator  # noqa: F401
from generative_models_amd.trainers import InfoQ as Q  # noqa: F401


class InfoGAN(_t.InfoGANModel):
    """info_gan.py:97-110"""


@_t.stock
class InfoGANTrainer(_t.InfoGANTrainerBase):
    """info_gan.py:112-325"""
