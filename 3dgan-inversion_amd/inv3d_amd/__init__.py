"""inv3d_amd -- MI355X-native EG3D inversion hot path (TriPlaneGenerator.synthesis forward+backward on gfx950).

Layout mirrors the reference's import paths for the hot path so callers can switch by import:
    inv3d_amd.torch_utils.ops.{bias_act, upfirdn2d, conv2d_resample, conv2d_gradfix, fma}
    inv3d_amd.training.{networks_stylegan2, superresolution, triplane}
    inv3d_amd.training.volumetric_rendering.{renderer, ray_sampler, math_utils}
All compute goes through libeg3d_hip.so (include/eg3d_hip.h); there is no CPU / eager fallback.
"""
from . import _lib  # noqa: F401
from .reference_binding import install_as_reference_modules  # noqa: F401

__version__ = '0.1.0'
