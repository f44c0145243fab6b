"""
event_utils_amd -- MI355X-native (gfx950 / CDNA4) implementation of the data-parallel core of TimoStoff/event_utils:
event -> image / voxel-grid binning (lib/representations) and linear-flow contrast maximisation (lib/contrast_max).
Python keeps the reference's function signatures; all event arithmetic runs in hand-written HIP kernels behind the
C ABI of include/evk.h (libevk.so).  There is no CPU fallback.

Module map (reference module -> here):
    lib.representations.image       -> event_utils_amd.representations.image
    lib.representations.voxel_grid  -> event_utils_amd.representations.voxel_grid
    lib.contrast_max.warps          -> event_utils_amd.contrast_max.warps
    lib.contrast_max.objectives     -> event_utils_amd.contrast_max.objectives
    lib.contrast_max.events_cmax    -> event_utils_amd.contrast_max.events_cmax
    lib.util.event_util             -> event_utils_amd.util.event_util      (events_bounds_mask)
(`event_utils_amd.lib.*` aliases the same modules under the reference's own dotted paths.)
"""
from .representations import *  # noqa: F401,F403
from .contrast_max.warps import warp_function, linvel_warp, warp_events  # noqa: F401
from .contrast_max.objectives import objective_function, variance_objective, get_iwe  # noqa: F401
from .contrast_max.events_cmax import optimize, optimize_contrast  # noqa: F401
from .util.event_util import events_bounds_mask  # noqa: F401
from .events import DeviceEvents  # noqa: F401
from ._device import check_errors, error_mode  # noqa: F401
from .tiled import release_scratch  # noqa: F401

__version__ = "0.1.0"
