"""tol_amd -- MI355X-native implementation of tol's SNOPT user-function path.

The product is the C-ABI library tol_amd/lib/libtolfg.so (sources in tol_amd/csrc, interface in
include/tolfg.h).  This package is the thin Python host layer over that ABI: ctypes bindings, device
buffers through torch, and the one-process-per-GPU batch driver.  There is no CPU fallback: every
evaluation needs a gfx950 device and raises TolfgError otherwise.
"""
from .capi import TolfgError, lib, lib_path, measure_lib   # noqa: F401
from . import capi                             # noqa: F401
from .host import Batch, Multi, Problem, Trajectory, device_alloc  # noqa: F401

__all__ = ["Batch", "Multi", "Problem", "Trajectory", "TolfgError", "capi", "device_alloc", "lib", "lib_path", "measure_lib"]
