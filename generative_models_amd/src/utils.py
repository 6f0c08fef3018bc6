"""Drop-in for the reference's src/utils.py (utils.py:6-53): to_var, to_cuda, get_data."""
import _bootstrap  # noqa: F401
import torch  # noqa: F401

from generative_models_amd.trainers import get_data, to_cuda, to_var  # noqa: F401

__all__ = ["to_var", "to_cuda", "get_data", "torch"]
