"""Drop-in for the reference's src/bir_vae.py (bir_vae.py:37-232): Encoder, Decoder, BIRVAE,
BIRVAETrainer with the same constructor / train() signatures and state_dict keys (encoder.linear/mu.*,
decoder.linear/recon.*); compute runs on the gfx950 kernels of generative_models_amd
(second half: Gaussian-kernel MMD kernel + numpy-RNG reparameterisation)."""
import _bootstrap  # noqa: F401
import numpy as np  # noqa: F401
import torch  # noqa: F401
import torch.nn as nn  # noqa: F401

from utils import *  # noqa: F401,F403
from generative_models_amd.trainers import BIRDecoder as Decoder  # noqa: F401
from generative_models_amd.trainers import BIREncoder as Encoder  # noqa: F401
from generative_models_amd.trainers import BIRVAE, BIRVAETrainer  # noqa: F401
