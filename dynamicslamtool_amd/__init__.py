"""dynamicslamtool_amd — MI355X-native MovingObjectRemoval hot path.

Host-side Python mirror of the C-ABI in include/mor_hip.h (ctypes).  The compute lives in
csrc/ (hand-written HIP for gfx950, built into libmor_hip.so); there is no CPU fallback:
importing `engine` without the built library raises.
"""
from .params import MorParams, parse_config, ref_default_params, kitti_params  # noqa: F401
