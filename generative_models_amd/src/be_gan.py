"""Drop-in for the reference's src/be_gan.py (be_gan.py:48-258)."""
import _bootstrap  # noqa: F401
import numpy as np  # noqa: F401
import torch  # noqa: F401
import torch.nn as nn  # noqa: F401

from utils import *  # noqa: F401,F403
from generative_models_amd import trainers as _t
from generative_models_amd.trainers import Generator  # noqa: F401
from generative_models_amd.trainers import AEDiscriminator as Discriminator  # noqa: F401


class BEGAN(_t.BEGANModel):
    """be_gan.py:79-90"""


@_t.stock
class BEGANTrainer(_t.BEGANTrainerBase):
    """be_gan.py:93-258"""
