"""Drop-in for the reference's src/vae.py (vae.py:47-223): Encoder, Decoder, VAE, VAETrainer with
the same constructor / train() signatures and state_dict keys (encoder.linear/mu/log_var.*,
decoder.linear/recon.*); compute runs on the gfx950 kernels of generative_models_amd."""
import _bootstrap  # noqa: F401
import numpy as np  # noqa: F401
import torch  # noqa: F401
import torch.nn as nn  # noqa: F401

from utils import *  # noqa: F401,F403
from generative_models_amd.trainers import Decoder, Encoder, VAE, VAETrainer  # noqa: F401
