"""Makes `generative_models_amd` importable when only this `src/` directory is on sys.path
(the reference's scripts/notebooks do `sys.path.append('../src'); from ns_gan import *`)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
