"""Reference: lib/util/event_util.py.  Only events_bounds_mask is on the hot path (it is fused into the IWE kernel);
this standalone version keeps the reference function available."""
import numpy as np
import torch

from .. import _device as D
from .. import _lib


def events_bounds_mask(xs, ys, x_min, x_max, y_min, y_max):
    """mask = 0.0 where x<=x_min or x>x_max or y<=y_min or y>y_max, else 1.0 (reference: event_util.py:15-28; note the
    asymmetry: x == x_min is rejected, x == x_max kept).  numpy in -> float64 numpy out; device tensors in ->
    device tensor out."""
    dev = D.require_gpu()
    on_device = isinstance(xs, torch.Tensor)
    xd, yd = D.to_device(xs, torch.float64, dev), D.to_device(ys, torch.float64, dev)
    mask = torch.empty_like(xd)
    _lib.call("evk_bounds_mask_f64", D.ptr(xd), D.ptr(yd), xd.shape[0], float(x_min), float(x_max), float(y_min),
              float(y_max), D.ptr(mask), D.stream())
    return mask if on_device else mask.cpu().numpy()
