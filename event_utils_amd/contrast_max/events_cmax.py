"""
Contrast-maximisation drivers.  Reference: lib/contrast_max/events_cmax.py (optimize_contrast :313-346, optimize
:348-368).  The BFGS loop stays on the host (scipy.optimize.fmin_bfgs, as upstream); the events are uploaded ONCE and
stay in HBM, every f / f' evaluation is a streaming pass of the fused kernel.

Not provided: grid_search_initial / grid_search_optimisation / optimize_contrast(grid_search_init=True) call an
undefined `recursive_search` upstream (events_cmax.py:233,336) and cannot run there either; plotting helpers are out of
scope.
"""
import numpy as np
import scipy.optimize as opt

from ..events import DeviceEvents
from .objectives import objective_function, variance_objective  # noqa: F401
from .warps import linvel_warp, warp_function  # noqa: F401


def optimize_contrast(xs, ys, ts, ps, warp_function, objective, optimizer=opt.fmin_bfgs, x0=None, numeric_grads=False,
                      blur_sigma=None, img_size=(180, 240), grid_search_init=False, minimum_events=200):
    """
    Optimise the contrast of a set of events with a gradient-based optimiser (reference: events_cmax.py:313-346):
    x0 = [0, 0], objective.iter_update(x0), then fmin_bfgs(f, x0, fprime | epsilon=1, args, callback=iter_update).
    xs may also be a DeviceEvents (ys, ts, ps are then ignored).
    """
    if grid_search_init and x0 is None:
        raise NotImplementedError("grid_search_init calls an undefined function upstream (events_cmax.py:336)")
    elif x0 is None:
        x0 = np.array([0, 0])
    objective.iter_update(x0)
    fused = getattr(warp_function, "fused_kernel", None) == "linvel" and isinstance(objective, objective_function)
    if fused:
        ev = xs if isinstance(xs, DeviceEvents) else DeviceEvents.from_arrays(xs, ys, ts, ps)
        args = (ev, None, None, None, warp_function, img_size, blur_sigma)      # resident events, uploaded once
    else:
        args = (xs, ys, ts, ps, warp_function, img_size, blur_sigma)
    if numeric_grads and hasattr(objective, "evaluate_numeric_gradient") and fused and optimizer is opt.fmin_bfgs:
        # same forward differences (epsilon = 1) scipy would take internally, but the three evaluations of one
        # gradient estimate share a single pass over the events
        argmax = optimizer(objective.evaluate_function, x0, fprime=objective.evaluate_numeric_gradient, args=args,
                           disp=False, callback=objective.iter_update)
    elif numeric_grads:
        argmax = optimizer(objective.evaluate_function, x0, args=args, epsilon=1, disp=False,
                           callback=objective.iter_update)
    else:
        argmax = optimizer(objective.evaluate_function, x0, fprime=objective.evaluate_gradient, args=args, disp=False,
                           callback=objective.iter_update)
    return argmax


def optimize(xs, ys, ts, ps, warp, obj, numeric_grads=True, img_size=(180, 240)):
    """optimize_contrast with blur_sigma=1.0; numeric gradients are forced when the objective has no derivative
    (reference: events_cmax.py:348-368)."""
    numeric_grads = numeric_grads if obj.has_derivative else True
    argmax_an = optimize_contrast(xs, ys, ts, ps, warp, obj, numeric_grads=numeric_grads, blur_sigma=1.0,
                                  img_size=img_size)
    return argmax_an
