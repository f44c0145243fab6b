"""Aliases under the reference's dotted paths: event_utils_amd.lib.representations.image, ...contrast_max.warps, ..."""
import sys as _sys

from .. import contrast_max, representations, transforms, util, visualization  # noqa: F401

for _name, _mod in (("representations", representations), ("contrast_max", contrast_max), ("util", util),
                    ("transforms", transforms), ("visualization", visualization)):
    _sys.modules[__name__ + "." + _name] = _mod
    for _sub in ("image", "voxel_grid", "warps", "objectives", "events_cmax", "event_util", "optic_flow", "draw_flow"):
        if hasattr(_mod, _sub):
            _sys.modules[__name__ + "." + _name + "." + _sub] = getattr(_mod, _sub)
