"""boofcv_amd -- MI355X (gfx950) provider for BoofCV's detect -> describe -> associate hot path.

The product is boofcv_amd/libboofhip.so (hand-written HIP kernels behind the C ABI of include/boofhip.h);
`boofcv_amd.api` mirrors the reference's Java interfaces on top of it.  Nothing here falls back to the CPU.
"""
from . import _lib  # noqa: F401
from .api import *  # noqa: F401,F403

__version__ = "0.1"
