"""
Contrast-maximisation drivers.  Reference: lib/contrast_max/events_cmax.py (optimize_contrast :313-346, optimize
:348-368, optimize_r2 :370-388, grid_search_initial :241-311, grid_search_optimisation :186-237, find_new_range
:162-184, draw_objective_function :103-160, grid_cmax :28-76, segmentation_mask_from_d_iwe :78-101).  The search loops
stay on the host (scipy.optimize.fmin_bfgs / the grid samplers, as upstream); the events are uploaded ONCE and stay in
HBM, every evaluation is a streaming pass of the fused kernels, and samplers that evaluate many flows on the same
events go through objective.evaluate_function_batch (three nearby flows per pass, one readback for all of them).

Upstream these samplers cannot run as shipped: grid_search_optimisation / optimize_contrast(grid_search_init=True)
call an undefined `recursive_search` (events_cmax.py:233,336), grid_search_initial needs numpy < 1.16
(np.vstack(map(...)) :294) and draw_objective_function uses `plt` without importing it.  Here `recursive_search` IS
grid_search_optimisation (the only reading consistent with its docstring); the golden vectors
(tests/golden/f14_search.npz) were produced by the reference with exactly those three names supplied.
"""
import copy

import numpy as np
import scipy.optimize as opt

from ..events import DeviceEvents
from .objectives import get_iwe, objective_function, soe_objective, variance_objective  # noqa: F401
from .warps import linvel_warp, uses_fused_linvel, warp_function  # noqa: F401


def _resident(xs, ys, ts, ps, warp_function, objective):
    """Upload the events once for the fused linear-flow path; plugin warps keep their host arrays."""
    if uses_fused_linvel(warp_function) and isinstance(objective, objective_function):
        # (float64 time stamps that are not float32 values stay on the float32 path as differences from ts[-1]: an optimiser
        # hands back an argmax, which that does not move -- DeviceEvents.from_arrays)
        ev = xs if isinstance(xs, DeviceEvents) else DeviceEvents.from_arrays(xs, ys, ts, ps, relative_time=True)
        ev.many_evaluations = True          # every caller of this is a search / an optimiser: tens of evaluations of these events
        return ev, None, None, None
    return xs, ys, ts, ps


def _evaluate_many(objective, params_list, xs, ys, ts, ps, warp_function, img_size, blur_sigma):
    batch = getattr(objective, "evaluate_function_batch", None)
    if batch is not None:
        return batch(params_list, xs, ys, ts, ps, warp_function, img_size, blur_sigma)
    return [objective.evaluate_function(params=q, xs=xs, ys=ys, ts=ts, ps=ps, warpfunc=warp_function,
                                        img_size=img_size, blur_sigma=blur_sigma) for q in params_list]


def find_new_range(search_axes, param):
    """New search range on one axis once the best sample `param` is known: from the previous to the next sample
    position around it, so that all the unsearched domain around it is covered (reference: events_cmax.py:162-184)."""
    i = np.searchsorted(search_axes, param)
    if i >= len(search_axes) - 1:
        below = above = np.abs(search_axes[-1] - search_axes[-2])
    elif i == 0:
        below = np.abs(search_axes[0] - search_axes[-1])
        above = np.abs(search_axes[0] - search_axes[1])
    else:
        below = np.abs(search_axes[i] - search_axes[i - 1])
        above = np.abs(search_axes[i] - search_axes[i + 1])
    return [param - below, param + above]


def grid_search_initial(xs, ys, ts, ps, warp_function, objective_function, img_size, param_ranges=None,
                        log_scale=True, num_samples_per_param=5):
    """
    One level of the SOFAS grid search (reference: events_cmax.py:241-311): sample every parameter axis at
    num_samples_per_param positions (evenly, or log-spaced = denser near the middle of the range), evaluate the
    objective (blur_sigma=1.0) at all num_samples_per_param^dims combinations and keep the best one.  Returns the
    upstream dict: 'params' (sample coordinates, parameter 0 varying fastest), 'eval', 'search_axes', 'min_params',
    'min_func_eval' (min_params stays None when no evaluation is below 0, as upstream).
    xs may also be a DeviceEvents (ys, ts, ps are then ignored).
    """
    assert num_samples_per_param % 2 == 1
    half = int(num_samples_per_param / 2.0) + 1
    if log_scale:
        steps = np.logspace(0, 2.0, half)[1:]
        steps = steps / steps[-1]
    else:
        steps = np.linspace(0, 1.0, half)[1:]
    if param_ranges is None:
        param_ranges = [[-150, 150] for _ in range(warp_function.dims)]
    axes = []
    for lo, hi in param_ranges:
        half_width = (hi - lo) / 2.0
        mid = lo + half_width
        axes.append(np.concatenate(((mid - steps * half_width)[::-1], np.array([mid]), mid + steps * half_width)))
    samples = list(zip(*(np.ravel(g) for g in np.meshgrid(*axes))))
    xs, ys, ts, ps = _resident(xs, ys, ts, ps, warp_function, objective_function)
    evals = _evaluate_many(objective_function, samples, xs, ys, ts, ps, warp_function, img_size, 1.0)
    best_eval, best_params = 0, None
    for q, f in zip(samples, evals):
        if f < best_eval:
            best_eval, best_params = f, q
    return {"params": samples, "eval": list(evals), "search_axes": axes, "min_params": best_params,
            "min_func_eval": best_eval}


def grid_search_optimisation(xs, ys, ts, ps, warp_function, objective_function, img_size, param_ranges=None,
                             log_scale=True, num_samples_per_param=5, depth=0, th0=1, max_iters=20):
    """
    Recursive grid search as per SOFAS (reference: events_cmax.py:186-237): grid_search_initial on the current ranges,
    then re-sample the neighbourhood of the best point (find_new_range) until the largest range is below th0 or depth
    reaches max_iters.  Returns the dict of the last level.  (Like upstream's recursive call, deeper levels use the
    default th0 / max_iters.)
    """
    assert num_samples_per_param % 2 == 1 and num_samples_per_param >= 5
    xs, ys, ts, ps = _resident(xs, ys, ts, ps, warp_function, objective_function)
    optimal = grid_search_initial(xs, ys, ts, ps, warp_function, copy.deepcopy(objective_function), img_size,
                                  param_ranges=param_ranges, log_scale=log_scale,
                                  num_samples_per_param=num_samples_per_param)
    new_ranges, widest = [], 0
    for axis, param in zip(optimal["search_axes"], optimal["min_params"]):
        r = find_new_range(axis, param)
        new_ranges.append(r)
        widest = max(widest, np.abs(r[1] - r[0]))
    if widest >= th0 and depth < max_iters:
        return recursive_search(xs, ys, ts, ps, warp_function, objective_function, img_size, param_ranges=new_ranges,
                                log_scale=log_scale, num_samples_per_param=num_samples_per_param, depth=depth + 1)
    return optimal


recursive_search = grid_search_optimisation      # the name upstream calls (events_cmax.py:233,336)


def objective_landscape(xs, ys, ts, ps, objective=None, warpfunc=None, x_range=(-200, 200), y_range=(-200, 200),
                        resolution=20, img_size=(180, 240), norm_min=None, norm_max=None):
    """The sampled, normalised image draw_objective_function shows (reference: events_cmax.py:122-133):
    img[y, x] = -f(x*resolution + x_range[0], y*resolution + y_range[0]) at blur_sigma=0, scaled to [0, 1]."""
    objective = variance_objective(minimum_events=1) if objective is None else objective
    warpfunc = linvel_warp() if warpfunc is None else warpfunc
    width, height = x_range[1] - x_range[0], y_range[1] - y_range[0]
    shape = (int(height / resolution + 0.5), int(width / resolution + 0.5))
    samples = [np.array([x * resolution + x_range[0], y * resolution + y_range[0]])
               for x in range(shape[1]) for y in range(shape[0])]
    xs, ys, ts, ps = _resident(xs, ys, ts, ps, warpfunc, objective)
    evals = _evaluate_many(objective, samples, xs, ys, ts, ps, warpfunc, img_size, 0)
    img = -np.array(evals, dtype=np.float64).reshape(shape[1], shape[0]).T
    norm_min = np.min(img) if norm_min is None else norm_min
    norm_max = np.max(img) if norm_max is None else norm_max
    return (img - norm_min) / ((norm_max - norm_min) + 1e-6)


def draw_objective_function(xs, ys, ts, ps, objective=None, warpfunc=None, x_range=(-200, 200), y_range=(-200, 200),
                            gt=(0, 0), show_gt=True, resolution=20, img_size=(180, 240), show_axes=True, norm_min=None,
                            norm_max=None, show=True):
    """Sample the objective over a range of flows and draw it with matplotlib (reference: events_cmax.py:103-160;
    objective defaults to variance_objective(minimum_events=1), warpfunc to linvel_warp()).  Returns the image
    (upstream returns None)."""
    import matplotlib.pyplot as plt
    img = objective_landscape(xs, ys, ts, ps, objective, warpfunc, x_range, y_range, resolution, img_size, norm_min,
                              norm_max)
    width, height = x_range[1] - x_range[0], y_range[1] - y_range[0]
    plt.imshow(img, interpolation='bilinear', cmap='viridis')
    if not show_axes:
        plt.xticks([])
        plt.yticks([])
    else:
        for ticks, rng in ((plt.xticks, x_range), (plt.yticks, y_range)):
            pos = ticks()[0][1:-1]
            ticks(ticks=pos, labels=["{}".format(int(v)) for v in np.linspace(rng[0], rng[1], len(pos))])
        plt.xlabel("$v_x$")
        plt.ylabel("$v_y$")
    if show_gt:
        plt.axhline(y=((gt[1] - y_range[0]) / height) * img.shape[0], color='r', linestyle='--')
        plt.axvline(x=((gt[0] - x_range[0]) / width) * img.shape[1], color='r', linestyle='--')
    if show:
        plt.show()
    return img


def segmentation_mask_from_d_iwe(d_iwe, th=None):
    """Binary mask of the pixels whose IWE derivative is large in either parameter (reference: events_cmax.py:78-101;
    thresholds default to the 95th percentile of the values above the 90th percentile of |d_iwe|).  Host-side
    post-processing of the (2, H+1, W+1) image get_iwe returned, as upstream."""
    d_iwe = np.asarray(d_iwe)
    floor = np.percentile(np.abs(d_iwe), 90)
    th = [np.percentile(c[np.abs(c) > floor], 95) if th is None else th for c in (d_iwe[0].ravel(), d_iwe[1].ravel())]
    hit = [(d_iwe[k] > th[k]).astype(int) + (d_iwe[k] < -th[k]).astype(int) for k in range(2)]
    return np.clip(hit[0] + hit[1], 0, 1)


def evk_bfgs(objective, x0, args, numeric_grads=False, callback=None, xtol=1e-3, gtol=1e-5, ftol=1e-6, maxiter=100, trace=None,
             unit_first=True, fast=True, native=None):
    """
    BFGS with a line search made for this objective: every quantity it asks for is ONE pass over the resident events and it
    asks for as few as it can.  scipy's fmin_bfgs (the reference's optimiser, events_cmax.py:343-345) runs a strong-Wolfe
    search whose every trial point is a function of the previous RESULT -- one event pass and one host round trip each,
    ~12 per iteration on these objectives, most of them spent near the optimum where the float32 images' rounding noise
    defeats its curvature test.  Here an iteration is two passes: (1) THREE candidate step lengths along the quasi-Newton
    direction, {1, 1/3, 3} x the current scale, evaluated in one three-flow pass (objective.evaluate_function_batch: they lie
    on one ray; every flow has its own LDS window origin), of which the best one that satisfies the Armijo condition is
    taken; (2) value and gradient at the accepted point (evaluate_function_and_gradient: one pass, or the three-flow
    forward-difference pass with numeric_grads) for the inverse-Hessian update.  Stops when the step is shorter than
    `xtol` (px/s), the gradient's largest component is below `gtol`, an accepted step improves the objective by less than
    `ftol` of its value (the images are float32: relative differences below ~3e-7 are summation-order noise, and a search
    that keeps following them never ends), or no candidate improves it at all.
    Returns the minimiser (numpy float64).  trace: optional list that receives (x, f, g) of every accepted point.
    fast / native: plumbing only (identical points either way) -- fast evaluates through closures bound to these events
    (objective.bind_fast), native runs the whole loop inside the library when the objective offers that (bind_native).
    """
    # (round 6) The iteration's own arithmetic is a handful of operations on dims-vectors (dims = 2 for the linear flow): as
    # numpy calls -- norm, dot, outer, eye, a dozen temporaries per iteration -- they cost ~25 us per event pass, a sixth of a
    # pass at 10 M events and a third at 1 M (tools/bfgs_profile.py).  Plain Python floats, same operations in the same order.
    n = len(x0)
    x = [float(v) for v in x0]
    # (round 6) an objective that can run this very loop inside the library (variance_objective.bind_native:
    # evk_cmax_bfgs_variance_tiled_f32, same passes, same arithmetic in C) does so: ~15 us less between two passes.  A callback
    # other than the objective's own iter_update may want to see every step as it happens: Python loop.
    if native is None:
        from .. import tiled
        native = tiled.FORCE["native_bfgs"]                # (A/B switch of the tools)
    if native and fast and n == 2 and hasattr(objective, "bind_native") and \
            (callback is None or getattr(callback, "__self__", None) is objective):
        run = objective.bind_native(*args)
        done = run(x, xtol, gtol, ftol, maxiter, numeric_grads, unit_first) if run is not None else None
        if done is not None:
            xf, points = done
            for k, (q, fv, gv) in enumerate(points):
                if trace is not None:
                    trace.append((q, fv, gv))
                if callback is not None and k > 0:
                    callback(q.copy())
            return xf
    fg_fn = objective.evaluate_function_and_numeric_gradient if numeric_grads else objective.evaluate_function_and_gradient
    rng = range(n)

    def dot(a, b):
        s = 0.0
        for i in rng:
            s += a[i] * b[i]
        return s

    def norm(a):
        return dot(a, a) ** 0.5

    def axpy(al, d, base):
        return [base[i] + al * d[i] for i in rng]

    def fg(q):
        fv, gv = fg_fn(np.array(q, dtype=np.float64), *args)
        return float(fv), [float(v) for v in gv]

    def f3(points):
        return [float(v) for v in objective.evaluate_function_batch([np.array(q, dtype=np.float64) for q in points], *args)]
    # an objective that can bind itself to these events (variance_objective.bind_fast) evaluates through closures that resolve
    # everything but the flow once: same library calls, same numbers, ~20 us less host work between two passes
    bound = objective.bind_fast(*args) if (fast and n == 2 and hasattr(objective, "bind_fast")) else None
    if bound is not None and not numeric_grads:
        fg, f3 = bound
    elif bound is not None:
        # numeric gradients (the reference's default: forward differences with epsilon = 1, events_cmax.py:343): f(x), f(x + e1),
        # f(x + e2) are ONE three-flow pass -- evaluate_function_and_numeric_gradient's arithmetic on the bound closure
        f3 = bound[1]

        def fg(q):
            pts = [list(q), [q[0] + 1.0, q[1]], [q[0], q[1] + 1.0]]
            fs = f3(pts)
            return fs[0], [(fs[1] - fs[0]) / (pts[1][0] - q[0]), (fs[2] - fs[0]) / (pts[2][1] - q[1])]

    def identity():
        return [[1.0 if i == j else 0.0 for j in rng] for i in rng]
    f, g = fg(x)
    if trace is not None:
        trace.append((np.array(x), f, np.array(g)))
    Hm = identity()
    have_curvature = False                              # the inverse Hessian carries at least one update
    scale = 1.0 / max(norm(g), 1e-12)                   # first step: a unit-length move along -g (as scipy's first trial)
    for _ in range(maxiter):
        if max(abs(v) for v in g) <= gtol:
            break
        d = [-dot(Hm[i], g) for i in rng]
        slope = dot(g, d)
        if not slope < 0.0:                              # not a descent direction: restart from steepest descent
            Hm = identity()
            have_curvature = False
            d, slope = [-v for v in g], -dot(g, g)
        # Once the inverse Hessian has been updated the quasi-Newton step itself (length 1) is the natural candidate: value
        # AND gradient there are one pass (what an accepted point needs anyway), so an iteration whose unit step satisfies the
        # Armijo condition costs ONE event pass instead of two (round 5: 18 -> ~12 passes at configs[2]).  Only when it does
        # not is the three-lengths search run, below the unit step.
        best, a, grown = None, scale, 0
        fg_new = None
        dn = norm(d)
        if unit_first and have_curvature:
            f1, g1 = fg(axpy(1.0, d, x))
            if f1 <= f + 1e-4 * slope:
                best, fg_new = (f1, 1.0), (f1, g1)
            else:
                a, grown = 1.0 / 9.0, 4          # the three lengths below the unit step: 1/27, 1/9, 1/3 -- and no growing back
        # line search: three step lengths per pass; while the longest one is the best, the next pass looks further out
        # (the first direction is -g with an unknown scale), while none satisfies the Armijo condition, closer in
        while fg_new is None and a * dn >= 0.5 * xtol:
            alphas = (a / 3.0, a, 3.0 * a)
            fs = f3([axpy(al, d, x) for al in alphas])
            ok = [(fv, al) for fv, al in zip(fs, alphas) if fv <= f + 1e-4 * al * slope]
            if ok:
                if best is None or min(ok)[0] < best[0]:
                    best = min(ok)
                if best[1] == alphas[2] and grown < 4:
                    a, grown = 9.0 * a, grown + 1
                    continue
                break
            if best is not None:
                break
            a /= 27.0
        if best is None:
            break
        step = best[1]
        x_new = axpy(step, d, x)
        f_new, g_new = fg_new if fg_new is not None else fg(x_new)
        s_vec = [x_new[i] - x[i] for i in rng]
        y_vec = [g_new[i] - g[i] for i in rng]
        sy = dot(y_vec, s_vec)
        if sy > 1e-12:
            # H <- (I - rho s y^T) H (I - rho y s^T) + rho s s^T
            have_curvature = True
            rho = 1.0 / sy
            A = [[(1.0 if i == j else 0.0) - rho * s_vec[i] * y_vec[j] for j in rng] for i in rng]
            AH = [[sum(A[i][k] * Hm[k][j] for k in rng) for j in rng] for i in rng]
            Hm = [[sum(AH[i][k] * A[j][k] for k in rng) + rho * s_vec[i] * s_vec[j] for j in rng] for i in rng]
        gain = f - f_new
        x, f, g, scale = x_new, f_new, g_new, 1.0
        if trace is not None:
            trace.append((np.array(x), f, np.array(g)))
        if callback is not None:
            callback(np.array(x))
        if norm(s_vec) < xtol or gain <= ftol * abs(f):
            break
    return np.array(x, dtype=np.float64)


def optimize_contrast(xs, ys, ts, ps, warp_function, objective, optimizer=opt.fmin_bfgs, x0=None, numeric_grads=False,
                      blur_sigma=None, img_size=(180, 240), grid_search_init=False, minimum_events=200):
    """
    Optimise the contrast of a set of events with a gradient-based optimiser (reference: events_cmax.py:313-346):
    x0 = [0, 0], objective.iter_update(x0), then fmin_bfgs(f, x0, fprime | epsilon=1, args, callback=iter_update).
    xs may also be a DeviceEvents (ys, ts, ps are then ignored).
    optimizer='evk_bfgs' (not upstream): evk_bfgs above -- the same quasi-Newton iteration with a line search that costs
    two event passes per iteration instead of scipy's ~12; fmin_bfgs stays the default, as upstream.
    """
    fused = uses_fused_linvel(warp_function) and isinstance(objective, objective_function)
    xs, ys, ts, ps = _resident(xs, ys, ts, ps, warp_function, objective)         # resident events, uploaded once
    if grid_search_init and x0 is None:
        # events_cmax.py:333-337: coarse-to-fine grid search on a copy of the objective without adaptive lifespan
        init_obj = copy.deepcopy(objective)
        init_obj.adaptive_lifespan = False
        minv = recursive_search(xs, ys, ts, ps, warp_function, init_obj, img_size, log_scale=False)
        x0 = minv["min_params"]
    elif x0 is None:
        x0 = np.array([0, 0])
    objective.iter_update(x0)
    args = (xs, ys, ts, ps, warp_function, img_size, blur_sigma)
    if isinstance(optimizer, str):
        if optimizer != "evk_bfgs":
            raise ValueError("optimizer must be a scipy-style callable or 'evk_bfgs'")
        need = ("evaluate_function_batch", "evaluate_function_and_numeric_gradient" if numeric_grads else "evaluate_function_and_gradient")
        if not all(hasattr(objective, a) for a in need):
            raise ValueError("optimizer='evk_bfgs' needs an objective with %s and %s" % need)
        return evk_bfgs(objective, x0, args, numeric_grads=numeric_grads, callback=objective.iter_update)
    last = {}

    def keep(x, fv, gv):
        last["x"], last["g"] = np.array(x, dtype=np.float64, copy=True), gv
        return fv

    def kept(x):
        return last["g"] if "x" in last and np.array_equal(last["x"], np.asarray(x, dtype=np.float64)) else None
    # (round 6) scipy's callbacks through the closures bound to these events (variance_objective.bind_fast: same library calls,
    # same numbers and return types as the public methods, ~8 us less Python per evaluation -- a tenth of an evaluation at the
    # reference's own window sizes, tools/eval_paths.py); None for any other objective, a subclass or an instance with its own
    # evaluating methods, plugin warps, sharded runs
    bound = objective.bind_fast(*args) if (fused and optimizer is opt.fmin_bfgs and hasattr(objective, "bind_fast")) else None
    if bound is not None and numeric_grads:
        f3 = bound[1]

        def f_num3(x, *a):          # evaluate_function_and_numeric_gradient's arithmetic (forward differences, epsilon = 1)
            q0, q1 = float(x[0]), float(x[1])
            e0, e1 = q0 + 1.0, q1 + 1.0
            fs = f3([[q0, q1], [e0, q1], [q0, e1]])
            return keep(x, np.float32(fs[0]), np.array([(fs[1] - fs[0]) / (e0 - q0), (fs[2] - fs[0]) / (e1 - q1)]))

        def g_num3(x, *a):
            g = kept(x)
            return (f_num3(x), last["g"])[1] if g is None else g
        argmax = optimizer(f_num3, x0, fprime=g_num3, args=args, disp=False, callback=objective.iter_update)
    elif bound is not None:
        fg = bound[0]

        def f_fg(x, *a):
            fv, gv = fg([float(x[0]), float(x[1])])
            return keep(x, np.float32(fv), np.array(gv, dtype=np.float32))

        def g_fg(x, *a):
            g = kept(x)
            return (f_fg(x), last["g"])[1] if g is None else g
        argmax = optimizer(f_fg, x0, fprime=g_fg, args=args, disp=False, callback=objective.iter_update)
    elif numeric_grads and hasattr(objective, "evaluate_function_and_numeric_gradient") and fused and optimizer is opt.fmin_bfgs:
        # same forward differences (epsilon = 1) scipy would take internally, but f(x), f(x + e1), f(x + e2) share a
        # single pass over the events, and that pass serves both the f and the f' request of a trial point
        def f_num(x, *a):
            return keep(x, *objective.evaluate_function_and_numeric_gradient(x, *a))

        def g_num(x, *a):
            g = kept(x)
            return objective.evaluate_numeric_gradient(x, *a) if g is None else g
        argmax = optimizer(f_num, x0, fprime=g_num, args=args, disp=False, callback=objective.iter_update)
    elif numeric_grads:
        argmax = optimizer(objective.evaluate_function, x0, args=args, epsilon=1, disp=False,
                           callback=objective.iter_update)
    elif hasattr(objective, "evaluate_function_and_gradient") and fused and optimizer is opt.fmin_bfgs:
        # the line search asks for f and f' at the same trial point (phi, then derphi): one pass over the events
        # yields both, the gradient is kept for the call that follows
        def f_and_keep(x, *a):
            return keep(x, *objective.evaluate_function_and_gradient(x, *a))

        def g_from_last(x, *a):
            g = kept(x)
            return objective.evaluate_gradient(x, *a) if g is None else g
        argmax = optimizer(f_and_keep, x0, fprime=g_from_last, args=args, disp=False, callback=objective.iter_update)
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


def optimize_r2(xs, ys, ts, ps, warp, obj, numeric_grads=True, img_size=(180, 240)):
    """Optimise `obj` with its default blur, then refine from that optimum with the sum-of-exponentials objective at
    blur_sigma=1.0 (reference: events_cmax.py:370-388; like upstream, img_size is NOT forwarded to optimize_contrast,
    whose default (180, 240) applies)."""
    numeric_grads = numeric_grads if obj.has_derivative else True
    xs, ys, ts, ps = _resident(xs, ys, ts, ps, warp, obj)
    argmax_an = optimize_contrast(xs, ys, ts, ps, warp, obj, numeric_grads=numeric_grads, blur_sigma=None)
    argmax_an = optimize_contrast(xs, ys, ts, ps, warp, soe_objective(), x0=argmax_an, numeric_grads=numeric_grads,
                                  blur_sigma=1.0)
    return argmax_an


def grid_cmax(xs, ys, ts, ps, roi_size=(20, 20), step=None, warp=None, obj=None, min_events=10):
    """
    Contrast maximisation per cell of a grid over the sensor (reference: events_cmax.py:28-76): for every
    step-sized region of interest holding more than min_events events, grid-search-initialised BFGS at blur 2.0, a
    refinement at blur 1.0, then the objective of the IWE of ALL events at the cell's flow.  Returns (params, rois as
    [y, x, step_y, step_x], function values).  Like upstream, `obj` is replaced per cell by
    variance_objective(adaptive_lifespan=True, minimum_events=105) and the resolution is inferred from the events
    (max + 1, lib/util/event_util.py:5-13), so xs / ys must be integer-valued.  The events are uploaded once; the cells
    are selected from the resident columns on the device.
    """
    warp = linvel_warp() if warp is None else warp
    step = roi_size if step is None else step
    if not uses_fused_linvel(warp):
        raise NotImplementedError("grid_cmax is provided for the fused linear-flow warp (upstream hard-wires it too, :47)")
    # the events go to the device once; the cells are cut out of the resident columns there
    everything = xs if isinstance(xs, DeviceEvents) else DeviceEvents.from_arrays(xs, ys, ts, ps, relative_time=True)
    ex, ey = everything.x, everything.y
    resolution = [int(ey.max().item()) + 1, int(ex.max().item()) + 1]
    results_params, results_rois, results_f_evals = [], [], []
    for xc in range(0, resolution[1], step[1]):
        in_cols = (ex >= xc) & (ex < xc + step[1])
        for yc in range(0, resolution[0], step[0]):
            sel = in_cols & (ey >= yc) & (ey < yc + step[0])
            if int(sel.sum().item()) <= min_events:
                continue
            roi = DeviceEvents(ex[sel], ey[sel], everything.t[sel], everything.p[sel])
            roi.t_offset = everything.t_offset
            obj = variance_objective(adaptive_lifespan=True, minimum_events=105)
            params = optimize_contrast(roi, None, None, None, warp, obj, numeric_grads=False, blur_sigma=2.0,
                                       img_size=resolution, grid_search_init=True)
            params = optimize_contrast(roi, None, None, None, warp, obj, numeric_grads=False, blur_sigma=1.0,
                                       img_size=resolution, x0=params)
            iwe, _ = get_iwe(params, everything, None, None, None, warp, resolution, use_polarity=True,
                             compute_gradient=False, return_events=False)
            results_params.append(params)
            results_rois.append([yc, xc, step[0], step[1]])
            results_f_evals.append(obj.evaluate_function(iwe=iwe))
    return results_params, results_rois, results_f_evals
