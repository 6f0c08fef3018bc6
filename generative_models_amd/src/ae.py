"""Drop-in for the reference's src/ae.py (ae.py:29-205): Encoder, Decoder, Autoencoder,
AutoencoderTrainer with the same constructor / train() signatures and state_dict keys
(encoder.linear.*, decoder.linear.*); compute runs on the gfx950 kernels of generative_models_amd
(the first "next" row, on the VAE step's kernels)."""
import _bootstrap  # noqa: F401
import numpy as np  # noqa: F401
import torch  # noqa: F401
import torch.nn as nn  # noqa: F401

from utils import *  # noqa: F401,F403
from generative_models_amd.trainers import AEDecoder as Decoder  # noqa: F401
from generative_models_amd.trainers import AEEncoder as Encoder  # noqa: F401
from generative_models_amd.trainers import Autoencoder, AutoencoderTrainer  # noqa: F401
