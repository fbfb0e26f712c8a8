"""nerf_sr_amd: MI355X-native supersampled volumetric render hot path of NeRF-SR.

The package is a thin host-side mirror of the reference's operator interface for
this one path (``forward_rays`` / ``render_rays`` / ``VanillaMLP.forward`` /
``VolumetricRenderer.forward`` / sampling / positional encoding) on top of a C-ABI
shared library of hand-written HIP kernels for gfx950 (``include/nsr.h``,
``nerf_sr_amd/csrc``).  There is no CPU fallback: importing the GPU-facing modules
without the built library raises.
"""
from . import weights, cameras  # noqa: F401  (numpy-only, importable anywhere)

__version__ = "0.1.0"
