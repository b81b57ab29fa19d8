"""srgan_amd -- MI355X-native Style-Restricted GAN train-step path.

Host-side mirror of the reference's operator API (pyfiles/model.py, pyfiles/util.py:455-553,
pyfiles/util_notebook.py:419-734) over the hand-written gfx950 kernels of libsrgan_hip.so.
"""
from . import _lib, ops  # noqa: F401

__all__ = ["_lib", "ops"]
