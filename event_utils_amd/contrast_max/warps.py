"""Motion-model plugins.  Reference: lib/contrast_max/warps.py (warp_function ABC :6-42, linvel_warp :44-61)."""
from abc import ABC, abstractmethod

import numpy as np
import torch

from .. import _device as D
from .. import _lib


class warp_function(ABC):
    """Base class of warps: .name, .dims and .warp(xs, ys, ts, ps, t0, params, compute_grad=False) ->
    (xs_warped, ys_warped, jacobian_x | None, jacobian_y | None) (reference: warps.py:6-42)."""

    def __init__(self, name, dims):
        self.name = name
        self.dims = dims
        super().__init__()

    @abstractmethod
    def warp(self, xs, ys, ts, ps, t0, params, compute_grad=False):
        pass


class linvel_warp(warp_function):
    """Linear velocity (global optic flow) warp (reference: warps.py:44-61):
    x' = x-(t-t0)*vx, y' = y-(t-t0)*vy, jacobian_x = [-dt; 0], jacobian_y = [0; -dt] (float64).
    When get_iwe / the objectives see this class they use the fused warp->mask->splat kernel instead of calling
    .warp() and materialising x', y' and the two (2, N) Jacobians."""

    fused_kernel = "linvel"          # informational; dispatch goes through uses_fused_linvel()

    def __init__(self):
        warp_function.__init__(self, 'linvel_warp', 2)

    def warp(self, xs, ys, ts, ps, t0, params, compute_grad=False):
        dev = D.require_gpu()
        on_device = isinstance(xs, torch.Tensor)
        xd, yd, td = (D.to_device(a, torch.float64, dev) for a in (xs, ys, ts))
        n = xd.shape[0]
        xo, yo = torch.empty_like(xd), torch.empty_like(yd)
        jx = torch.empty((2, n), dtype=torch.float64, device=dev) if compute_grad else None
        jy = torch.empty((2, n), dtype=torch.float64, device=dev) if compute_grad else None
        _lib.call("evk_warp_linvel_f64", D.ptr(xd), D.ptr(yd), D.ptr(td), n, float(t0), float(params[0]),
                  float(params[1]), D.ptr(xo), D.ptr(yo), D.ptr(jx), D.ptr(jy), D.stream())
        if on_device:
            return xo, yo, jx, jy
        return (xo.cpu().numpy(), yo.cpu().numpy(), jx.cpu().numpy() if compute_grad else None,
                jy.cpu().numpy() if compute_grad else None)


def uses_fused_linvel(warpfunc):
    """True when `warpfunc` warps exactly like linvel_warp, so that the fused warp -> mask -> splat kernels may replace
    its warp(): the class itself, or a subclass that did NOT override warp() (a plugin that did must be called)."""
    return isinstance(warpfunc, linvel_warp) and type(warpfunc).warp is linvel_warp.warp


def warp_events(xs, ys, ts, ps, t0, params, compute_grad=False):
    """Alias named by BASELINE.json's north_star: linvel_warp().warp(...) (SURVEY.md headline facts)."""
    return linvel_warp().warp(xs, ys, ts, ps, t0, params, compute_grad=compute_grad)


class xyztheta_warp(warp_function):
    """4-DoF x, y, z, rotation warp: an empty stub upstream (warps.py:63-72, body `pass`) and here."""

    def __init__(self):
        warp_function.__init__(self, 'xyztheta_warp', 4)

    def warp(self, xs, ys, ts, ps, t0, params, compute_grad=False):
        pass


class pure_rotation_warp(warp_function):
    """Pure rotation warp: an empty stub upstream (warps.py:74-83) and here."""

    def __init__(self):
        warp_function.__init__(self, 'pure_rotation_warp', 4)

    def warp(self, xs, ys, ts, ps, t0, params, compute_grad=False):
        pass
