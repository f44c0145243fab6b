"""
Contrast-maximisation objectives on MI355X.  Mirrors the reference's lib/contrast_max/objectives.py plugin API
(objective_function ABC :10-140, get_iwe :165-199, variance_objective :202-264).  Events live on the device
(events.DeviceEvents); one evaluation = one streaming pass of the fused warp -> mask -> bilinear-splat kernel over the
events + a few image-sized kernels; only the motion parameters and 4 doubles cross PCIe.

Extensions (all default to the reference's behaviour):
  objective.sensor_size   None = quirk Q1 (the IWE canvas is always (181, 241), objectives.py:191-192);
                          (H, W) = use that sensor size for the canvas (configs with img_size != (180, 240)).
  objective.reference_exact  True = quirks Q4/Q5 (channel-mixing 3-D blur of dIWE, un-blurred IWE in the gradient);
                          False = per-channel blur and blurred IWE: the true gradient of evaluate_function.
  objective.process_group / objective.distributed  event-sharded data parallelism: each rank accumulates its shard,
                          IWE (+dIWE) are all-reduced over RCCL before blur / reductions.
"""
import weakref
from abc import ABC, abstractmethod

import numpy as np
import torch

from .. import _device as D
from .. import _lib
from .. import tiled
from ..events import DeviceEvents
from ..representations.image import _events_to_image_drv_device, image_to_event_weights
from ..util.event_util import events_bounds_mask
from .warps import uses_fused_linvel


def gaussian_kernel1d(sigma, truncate=4.0):
    """The kernel scipy.ndimage.gaussian_filter builds (order 0): radius int(truncate*sigma+0.5), normalised
    exp(-x^2/(2 sigma^2)) in float64 (the reference calls scipy at objectives.py:233,253)."""
    sigma = float(sigma)
    radius = int(truncate * sigma + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum(), radius


_kernel_cache = {}


def _blur_kernel(sigma):
    """(host float64 weights, radius) for evk_*; radius -1 = no blur (blur_sigma <= 0, objectives.py:232)."""
    if not sigma > 0:
        return None, -1
    key = float(sigma)
    if key not in _kernel_cache:
        w, radius = gaussian_kernel1d(key)
        _kernel_cache[key] = (np.ascontiguousarray(w, dtype=np.float64), radius)
    return _kernel_cache[key]


def gaussian_filter_device(src, sigma, truncate=4.0):
    """scipy.ndimage.gaussian_filter(src, sigma) (mode='reflect') for a 2-D or 3-D float32 device tensor."""
    w, radius = gaussian_kernel1d(sigma, truncate)
    src = src.contiguous()
    dst, tmp = torch.empty_like(src), torch.empty_like(src)
    dims = np.array(src.shape, dtype=np.int32)
    w = np.ascontiguousarray(w, dtype=np.float64)
    _lib.call("evk_gaussian_filter_f32", D.ptr(src), D.ptr(dst), D.ptr(tmp), src.dim(), D.host_ptr(dims),
              D.host_ptr(w), radius, D.stream())
    return dst


def _as_device_events(xs, ys, ts, ps):
    if isinstance(xs, DeviceEvents):
        return xs
    return DeviceEvents.from_arrays(xs, ys, ts, ps)


def iwe_device(params, ev, img_size, compute_gradient=False, use_polarity=True, sensor_size=None, impl=None,
               process_group=None, distributed=False, t_ref=None):
    """Fused linear-flow get_iwe on device-resident events -> (iwe, d_iwe | None) float32 device tensors of shape
    (H+1, W+1) / (2, H+1, W+1).  t_ref defaults to ts[-1] of `ev` (objectives.py:186); an event-sharded caller passes
    the GLOBAL ts[-1]."""
    dev = ev.device
    ss = (180, 240) if sensor_size is None else sensor_size       # Q1
    ch, cw = int(ss[0]) + 1, int(ss[1]) + 1
    buf = torch.zeros((3 if compute_gradient else 1, ch, cw), dtype=torch.float32, device=dev)
    iwe, diwe = buf[0], (buf[1:3] if compute_gradient else None)
    flags = (0 if use_polarity else _lib.EVK_IWE_ABS_POLARITY) | (_lib.EVK_IWE_GRADIENT if compute_gradient else 0)
    t_ref = ev.t_at(-1) if t_ref is None else t_ref - ev.t_offset       # (an ABSOLUTE time; the column may be relative)
    if len(ev):
        tiled.iwe_linvel(ev, float(t_ref), float(params[0]), float(params[1]), float(img_size[1]),
                         float(img_size[0]), ch, cw, flags, iwe, diwe, impl=impl)
    if distributed or process_group is not None:
        from .. import distributed as DD
        DD.all_reduce_sum_(buf, process_group, force=True)
    return iwe, diwe


def cut_events_to_lifespan(xs, ys, ts, ps, params, pixel_crossings, minimum_events=10000):
    """Cut the events down to the lifespan pixel_crossings / |params| before the last timestamp, keeping at least
    minimum_events; the last event is dropped (reference: objectives.py:143-163; host-side slicing, no print)."""
    dt = pixel_crossings / np.linalg.norm(params)
    s_idx = np.searchsorted(ts, ts[-1] - dt)
    if len(xs) - s_idx < minimum_events:
        s_idx = len(xs) - minimum_events
    return xs[s_idx:-1], ys[s_idx:-1], ts[s_idx:-1], ps[s_idx:-1]


def _abs_device(t):
    """|t| of a float32 / float64 tensor through evk_abs (device tensors stay on the device, host tensors come back)."""
    dev = D.require_gpu()
    td = D.to_device(t, t.dtype if t.dtype in (torch.float32, torch.float64) else torch.float64, dev)
    out = torch.empty_like(td)
    _lib.call("evk_abs", D.ptr(td), td.element_size(), td.numel(), D.ptr(out), D.stream())
    return out if t.is_cuda else out.to(t.device)


def get_iwe(params, xs, ys, ts, ps, warpfunc, img_size, compute_gradient=False, use_polarity=True,
            return_events=False, return_per_event_contrast=False, sensor_size=None):
    """
    Image of warped events and its derivative w.r.t. the motion parameters (reference: objectives.py:165-199):
    warp at t0 = ts[-1] (:186) -> events_bounds_mask(0, img_size[1], 0, img_size[0]) (:187) -> multiply everything by
    the mask (:188-190) -> events_to_image_drv with its DEFAULT sensor_size (:191-192, quirk Q1; pass sensor_size to
    override).  Returns numpy float32 (iwe, d_iwe | None [, (xs, ys)] [, per-event contrast]).
    linvel_warp uses the fused kernel; any other warp_function plugin is called as upstream and its output goes
    through the generic mask + splat kernels.
    """
    fused = uses_fused_linvel(warpfunc)
    if fused and not return_events and not return_per_event_contrast:
        ev = _as_device_events(xs, ys, ts, ps)
        iwe, diwe = iwe_device(params, ev, img_size, compute_gradient, use_polarity, sensor_size)
        return iwe.cpu().numpy(), (diwe.cpu().numpy() if diwe is not None else None)
    # generic plugin path (and return_events): materialise the warp as the reference does
    if isinstance(xs, DeviceEvents):
        ev = xs
        xs, ys, ts, ps = (c.double() for c in (ev.x, ev.y, ev.t, ev.p * ev.p_scale))
    if not use_polarity:
        ps = np.abs(ps) if not isinstance(ps, torch.Tensor) else _abs_device(ps)
    t0 = ts[-1] if not isinstance(ts, torch.Tensor) else float(ts[-1].item())
    xw, yw, jx, jy = warpfunc.warp(xs, ys, ts, ps, t0, params, compute_grad=compute_gradient)
    mask = events_bounds_mask(xw, yw, 0, img_size[1], 0, img_size[0])
    xw, yw, pm = xw * mask, yw * mask, ps * mask
    if compute_gradient:
        jx, jy = jx * mask, jy * mask
    kw = {} if sensor_size is None else {"sensor_size": tuple(sensor_size)}
    iwe, diwe = _events_to_image_drv_device(xw, yw, pm, jx, jy, kw.get("sensor_size", (180, 240)), True, 'bilinear',
                                            True, compute_gradient)
    returnval = [iwe.cpu().numpy(), diwe.cpu().numpy() if diwe is not None else None]
    to_np = (lambda a: a.cpu().numpy()) if isinstance(xw, torch.Tensor) else (lambda a: a)
    if return_events:
        returnval.append((to_np(xw), to_np(yw)))
    if return_per_event_contrast:       # local contrast of every warped event in the IWE (objectives.py:196-198)
        returnval.append(to_np(image_to_event_weights(xw, yw, iwe)))
    return tuple(returnval)


class objective_function(ABC):
    """Parent class of contrast-maximisation objectives (reference: objectives.py:10-140): constructor attributes,
    abstract evaluate_function / evaluate_gradient, iter_update (:113-127), update_lifespan (:129-140)."""

    def __init__(self, name="template", use_polarity=True, has_derivative=True, default_blur=1.0,
                 adaptive_lifespan=False, pixel_crossings=5, minimum_events=10000):
        self.name = name
        self.use_polarity = use_polarity
        self.has_derivative = has_derivative
        self.default_blur = default_blur
        self.adaptive_lifespan = adaptive_lifespan
        self.pixel_crossings = pixel_crossings
        self.minimum_events = minimum_events

        self.recompute_lifespan = True
        self.lifespan = 0.5
        self.s_idx = 0
        self.num_events = None
        # extensions, see module docstring
        self.sensor_size = None
        self.reference_exact = True
        self.process_group = None
        self.distributed = False
        self.impl = None
        self.t_ref = None
        super().__init__()

    @abstractmethod
    def evaluate_function(self, params=None, xs=None, ys=None, ts=None, ps=None, warpfunc=None, img_size=None,
                          blur_sigma=None, showimg=False, iwe=None):
        pass

    @abstractmethod
    def evaluate_gradient(self, params=None, xs=None, ys=None, ts=None, ps=None, warpfunc=None, img_size=None,
                          blur_sigma=None, showimg=False, iwe=None, d_iwe=None):
        pass

    def evaluate_function_batch(self, params_list, xs=None, ys=None, ts=None, ps=None, warpfunc=None, img_size=None,
                                blur_sigma=None):
        """evaluate_function at K parameter vectors on the same events (grid search / objective landscapes,
        events_cmax.py:127-130,297-305) -> list of K values.  Generic version: K separate evaluations on the resident
        events; variance_objective evaluates three nearby flows per pass over the events."""
        return [self.evaluate_function(q, xs, ys, ts, ps, warpfunc, img_size, blur_sigma) for q in params_list]

    def iter_update(self, params, pixel_crossings=None):
        """Callback at each optimisation step: lifespan = pixel_crossings / |params| (5 if 0) (objectives.py:113-127)."""
        pixel_crossings = self.pixel_crossings if pixel_crossings is None else pixel_crossings
        magnitude = np.linalg.norm(params)
        if magnitude == 0:
            dt = 5
        else:
            dt = pixel_crossings / magnitude
        self.lifespan = dt
        self.recompute_lifespan = True

    def update_lifespan(self, ts):
        """New start index of the events used in optimisation (objectives.py:129-140; no prints on the hot path)."""
        if self.adaptive_lifespan:
            self.s_idx = np.searchsorted(ts, ts[-1] - self.lifespan)
            self.s_idx = len(ts) - self.minimum_events if len(ts) - self.s_idx < self.minimum_events else self.s_idx
        if self.num_events is None:
            self.num_events = len(ts) - self.s_idx

    # -- shared device plumbing ------------------------------------------------------------------------------
    def _lifespan_cut(self, ev):
        """xs[s_idx:-1] ... and ps*100 (objectives.py:217-225, quirk Q10: the last event is dropped)."""
        if not self.adaptive_lifespan:
            return ev
        if self.recompute_lifespan:
            self.update_lifespan(ev.t_host())
            self.recompute_lifespan = False
        # (round 6) the cut changes once per optimiser iteration (iter_update), the line search in between evaluates the SAME
        # cut several times: one view per (event set, start index), so that its buckets are made once and not per evaluation
        # (kept ON the event set, so that it goes when the events go)
        c = ev.__dict__.get("_lifespan_cut")
        if c is not None and c[0] == int(self.s_idx):
            return c[1]
        cut = ev.slice(int(self.s_idx), -1).scaled(100.0)
        cut.many_evaluations = ev.many_evaluations
        ev.__dict__["_lifespan_cut"] = (int(self.s_idx), cut)
        return cut

    # caches that tie the object to device buffers and marshalled library calls: not part of its state (copy.deepcopy of an
    # objective -- grid_search_optimisation, optimize_contrast(grid_search_init=True) do it, as upstream -- and pickling)
    _TRANSIENT = ("_fast_memo",)

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k not in self._TRANSIENT}

    def _one_call(self, params, xs, ys, ts, ps, warpfunc, img_size, blur_sigma, grad, post_flags):
        """Whole evaluation in ONE library call (tiled.cmax_variance) -> 4 doubles on the host, or None when that
        path does not apply (plugin warp, direct-kernel fallback).
        Event-sharded run (distributed.shard_objective): the same call stops after the gather (EVK_POST_NONE), the
        persistent (1 | 3, H+1, W+1) buffer is all-reduced in place over the ranks, then one fused blur + reduction call
        finishes -- no allocation per evaluation, and every rank holds the same scalars."""
        if not uses_fused_linvel(warpfunc):
            return None
        # (round 6) the SAME evaluation as the last one but for the flow -- a scipy line search, a sampler, bench.py's loops: the
        # resolved call is repeated directly (tiled.cmax_variance_again checks that its buffers are still the current ones)
        memo, memo_key = self.__dict__.get("_fast_memo"), None
        if isinstance(xs, DeviceEvents) and not self.adaptive_lifespan and not (self.distributed or self.process_group is not None) \
                and not getattr(self, "enqueue_only", False):
            memo_key = (id(xs), grad, post_flags, blur_sigma, float(img_size[0]), float(img_size[1]), self.impl, self.use_polarity,
                        self.sensor_size, self.t_ref, tiled.FORCE["iwe_fixed"], tiled.FORCE["iwe_records"], tiled.default_impl(),
                        xs.p_scale)
            if memo is not None and memo[0] == memo_key and memo[1]() is xs:
                r = tiled.cmax_variance_again(memo[2], float(params[0]), float(params[1]))
                if r is not None:
                    return r.copy()
        ev = self._lifespan_cut(_as_device_events(xs, ys, ts, ps))
        sharded = self.distributed or self.process_group is not None
        if len(ev) == 0 and not sharded:
            return None
        dev = ev.device
        ss = (180, 240) if self.sensor_size is None else self.sensor_size
        ch, cw = int(ss[0]) + 1, int(ss[1]) + 1
        flags = (0 if self.use_polarity else _lib.EVK_IWE_ABS_POLARITY) | (_lib.EVK_IWE_GRADIENT if grad else 0)
        t_ref = ev.t_at(-1) if self.t_ref is None else self.t_ref - ev.t_offset    # (t_ref is an ABSOLUTE time)
        w, radius = _blur_kernel(blur_sigma)
        planes = 3 if grad else 1
        buf = tiled._buf("iwe_buf", planes * ch * cw * 4, dev)
        out, (scratch, nbytes) = D.out4(dev), D.reduce_scratch(dev)
        res = self.__dict__.setdefault("_res4", np.empty(4, dtype=np.float64))   # filled by the call itself
        if not sharded:
            # (enqueue_only: measurement hook of bench.py -- the evaluation is enqueued, nothing is read back, the returned
            # scalars are stale: K such calls between two HIP events give the device time of one evaluation)
            ok = tiled.cmax_variance(ev, float(t_ref), float(params[0]), float(params[1]), float(img_size[1]),
                                     float(img_size[0]), ch, cw, flags, w, radius, post_flags, buf, out, scratch, nbytes,
                                     impl=self.impl, host_out=None if getattr(self, "enqueue_only", False) else res)
            if ok and memo_key is not None:
                c = tiled.cmax_variance_entry(ev, post_flags, True, self)
                self.__dict__["_fast_memo"] = (memo_key, weakref.ref(ev), c) if c is not None else None
            return res.copy() if ok else None
        from .. import distributed as DD
        img = buf[:planes * ch * cw * 4].view(torch.float32).view(planes, ch, cw)

        def local_iwe():
            ok = len(ev) > 0 and tiled.cmax_variance(ev, float(t_ref), float(params[0]), float(params[1]),
                                                     float(img_size[1]), float(img_size[0]), ch, cw, flags, w, radius,
                                                     _lib.EVK_POST_NONE, buf, out, scratch, nbytes, impl=self.impl)
            if not ok:  # empty shard, or the direct kernels had to take over on this rank: same buffer, same collective
                img.zero_()
                if len(ev):
                    tiled.iwe_linvel(ev, float(t_ref), float(params[0]), float(params[1]), float(img_size[1]),
                                     float(img_size[0]), ch, cw, flags, img[0], img[1:3] if grad else None, impl=self.impl)
            return img

        def finish(img):
            wp = D.host_ptr(w) if w is not None else None
            if grad and (post_flags & _lib.EVK_POST_VALUE):
                _lib.call("evk_objective_variance_fg_f32", D.ptr(img), D.ptr(img[1:]), ch, cw, wp, radius,
                          post_flags & ~_lib.EVK_POST_VALUE, D.ptr(out), D.ptr(scratch), nbytes, D.stream())
            elif grad:
                _lib.call("evk_objective_variance_grad_f32", D.ptr(img), D.ptr(img[1:]), ch, cw, wp, radius, post_flags,
                          D.ptr(out), D.ptr(scratch), nbytes, D.stream())
            else:
                _lib.call("evk_objective_variance_f32", D.ptr(img), ch, cw, wp, radius, D.ptr(out), D.ptr(scratch),
                          nbytes, D.stream())
            return out.cpu().numpy()
        if DD.post_mode() == "rows":
            mode = (3 if post_flags & _lib.EVK_POST_VALUE else 1) if grad else 0
            sums = self.__dict__.setdefault("_sums8", torch.zeros(8, dtype=torch.float64, device=dev))

            def rows_post(block, y_lo, y_hi):
                _lib.call("evk_objective_variance_rows_f32", D.ptr(block), mode, int(block.shape[1]), cw, int(y_lo), int(y_hi),
                          D.host_ptr(w) if w is not None else None, radius, post_flags & ~_lib.EVK_POST_VALUE, D.ptr(sums),
                          D.ptr(scratch), nbytes, D.stream())
                return sums
            return DD.sharded_evaluate_rows(local_iwe, rows_post, max(radius, 0), mode, self.process_group)
        return DD.sharded_evaluate(local_iwe, finish, self.process_group)

    def _iwe(self, params, xs, ys, ts, ps, warpfunc, img_size, compute_gradient):
        fused = uses_fused_linvel(warpfunc)
        if fused:
            ev = self._lifespan_cut(_as_device_events(xs, ys, ts, ps))
            return iwe_device(params, ev, img_size, compute_gradient, self.use_polarity, self.sensor_size, self.impl,
                              self.process_group, self.distributed, self.t_ref)
        if self.adaptive_lifespan:
            if self.recompute_lifespan:
                self.update_lifespan(ts)
                self.recompute_lifespan = False
            xs, ys, ts, ps = xs[self.s_idx:-1], ys[self.s_idx:-1], ts[self.s_idx:-1], ps[self.s_idx:-1]
            ps = ps * 100
        dev = D.require_gpu()
        iwe, diwe = get_iwe(params, xs, ys, ts, ps, warpfunc, img_size, compute_gradient=compute_gradient,
                            use_polarity=self.use_polarity, sensor_size=self.sensor_size)
        return D.to_device(iwe, torch.float32, dev), (D.to_device(diwe, torch.float32, dev) if diwe is not None else None)


class variance_objective(objective_function):
    """Variance objective (Gallego et al.; reference: objectives.py:202-264)."""

    def __init__(self, adaptive_lifespan=False, minimum_events=10000):
        super().__init__(name="variance", use_polarity=True, has_derivative=True, default_blur=1.0,
                         adaptive_lifespan=adaptive_lifespan, pixel_crossings=5, minimum_events=minimum_events)

    def evaluate_function(self, params=None, xs=None, ys=None, ts=None, ps=None, warpfunc=None, img_size=None,
                          blur_sigma=None, showimg=False, iwe=None):
        """-var(blur(iwe) - mean) over the whole padded image (objectives.py:211-236, Q6)."""
        dev = D.require_gpu()
        blur_sigma = self.default_blur if blur_sigma is None else blur_sigma
        if iwe is None:
            res = self._one_call(params, xs, ys, ts, ps, warpfunc, img_size, blur_sigma, False, 0)
            if res is not None:
                return np.float32(-res[1])
            iwe, _ = self._iwe(params, xs, ys, ts, ps, warpfunc, img_size, False)
        else:
            iwe = D.to_device(iwe, torch.float32, dev)
        w, radius = _blur_kernel(blur_sigma)
        iwe = iwe.contiguous()
        out, (scratch, nbytes) = D.out4(dev), D.reduce_scratch(dev)
        # fused blur (both axes) + mean / variance reduction: one launch + a 1-block finalise
        _lib.call("evk_objective_variance_f32", D.ptr(iwe), iwe.shape[0], iwe.shape[1],
                  D.host_ptr(w) if w is not None else None, radius, D.ptr(out), D.ptr(scratch), nbytes, D.stream())
        loss = out[1].item()
        return np.float32(-loss)

    def _batch3_setup(self, xs, ys, ts, ps, warpfunc, blur_sigma):
        """Device state shared by the three-flows-per-pass launches, or None when that kernel does not apply (plugin
        warp, event-sharded run, no events)."""
        if (not uses_fused_linvel(warpfunc) or self.distributed or self.process_group is not None):
            return None
        ev = self._lifespan_cut(_as_device_events(xs, ys, ts, ps))
        if len(ev) == 0:
            return None
        dev = ev.device
        ss = (180, 240) if self.sensor_size is None else self.sensor_size
        ch, cw = int(ss[0]) + 1, int(ss[1]) + 1
        flags = 0 if self.use_polarity else _lib.EVK_IWE_ABS_POLARITY
        t_ref = ev.t_at(-1) if self.t_ref is None else self.t_ref - ev.t_offset    # (t_ref is an ABSOLUTE time)
        w, radius = _blur_kernel(blur_sigma)
        buf = tiled._buf("iwe_buf", 3 * ch * cw * 4, dev)
        scratch, nbytes = D.reduce_scratch(dev)

        def launch(trio, out12, img_size, host_out=None):
            return tiled.cmax_variance_batch3(ev, float(t_ref), [float(q[0]) for q in trio], [float(q[1]) for q in trio],
                                              float(img_size[1]), float(img_size[0]), ch, cw, flags, w, radius, buf,
                                              out12, scratch, nbytes, impl=self.impl, host_out=host_out)
        return ev, float(t_ref), launch

    def _evaluators_overridden(self):
        """A subclass that redefines an evaluating method (a plugin objective derived from this one) must be evaluated through
        its own methods: the bound short cuts below call the library for THIS class's arithmetic."""
        cls = type(self)
        names = ("evaluate_function", "evaluate_gradient", "evaluate_function_and_gradient", "evaluate_function_batch",
                 "evaluate_numeric_gradient", "evaluate_function_and_numeric_gradient", "_one_call", "_batch3_setup")
        # (a method replaced on the INSTANCE -- a tracing wrapper, as tests/test_gpu_parity.py installs -- counts as well)
        return any(getattr(cls, m, None) is not getattr(variance_objective, m) or m in self.__dict__ for m in names)

    def bind_fast(self, xs, ys, ts, ps, warpfunc, img_size, blur_sigma):
        """(fg, f3) closures for a loop that evaluates THIS objective on THESE events many times (events_cmax.evk_bfgs):
        fg(q) -> (f, [g0, g1]) = evaluate_function_and_gradient, f3([q0, q1, q2]) -> [f0, f1, f2] = evaluate_function_batch,
        same library calls, same numbers -- with everything that does not depend on q resolved ONCE (device, buffers, blur
        weights, the marshalled arguments of the calls: tiled.cmax_variance's cache entry is patched directly).  The gap
        between two passes of an optimisation -- result read, Python, next enqueue -- was 41-43 us on the kernel timeline
        (tools/bfgs_timeline.sh), a third of a pass at 10 M events.  None when the one-call path does not apply (plugin warp,
        sharded run, adaptive lifespan, direct-kernel regime): the caller then uses the public methods."""
        if (not uses_fused_linvel(warpfunc) or self.distributed or self.process_group is not None or self.adaptive_lifespan
                or getattr(self, "enqueue_only", False) or self._evaluators_overridden()):
            return None
        ev = _as_device_events(xs, ys, ts, ps)
        if len(ev) == 0:
            return None
        blur = self.default_blur if blur_sigma is None else blur_sigma
        post = (_lib.EVK_POST_MIX if self.reference_exact else _lib.EVK_POST_BLUR_IWE) | _lib.EVK_POST_VALUE
        f32 = np.float32

        def fg(q):
            # (the first call, and any call whose flow the tiled kernels cannot take, goes through the public method)
            c = cfg[0]
            if c is not None:
                r = tiled.cmax_variance_again(c, float(q[0]), float(q[1]))
                if r is not None:
                    return float(f32(-r[3])), [float(f32(-r[0])), float(f32(-r[1]))]
            fv, gv = self.evaluate_function_and_gradient(np.asarray(q, dtype=np.float64), ev, None, None, None, warpfunc, img_size, blur)
            if c is None:
                cfg[0] = tiled.cmax_variance_entry(ev, post | 0, True, self)
            return float(fv), [float(v) for v in gv]

        def f3(points):
            c = cf3[0]
            if c is not None:
                r = tiled.cmax_variance_batch3_again(c, [float(p_[0]) for p_ in points], [float(p_[1]) for p_ in points])
                if r is not None:
                    return [float(f32(-r[4 * k + 1])) for k in range(3)]
            vals = self.evaluate_function_batch([np.asarray(p_, dtype=np.float64) for p_ in points], ev, None, None, None,
                                                warpfunc, img_size, blur)
            if c is None:
                cf3[0] = tiled.cmax_variance_entry(ev, None, False, self)
            return [float(v) for v in vals]
        cfg, cf3 = [None], [None]
        return fg, f3

    def bind_native(self, xs, ys, ts, ps, warpfunc, img_size, blur_sigma):
        """run(x0, xtol, gtol, ftol, maxiter, numeric_grads, unit_first) -> (x, [(x, f, g), ...]) | None: events_cmax.evk_bfgs on
        THIS objective and THESE events as ONE library call (tiled.cmax_bfgs; include/evk.h:
        evk_cmax_bfgs_variance_tiled_f32) -- the same passes at the same flows, the iteration's arithmetic in C.  None (from
        here, or from run) when the one-call path does not apply: the same conditions as bind_fast, or a trial flow the tiled
        kernels cannot take; the caller then runs its Python loop."""
        if (not uses_fused_linvel(warpfunc) or self.distributed or self.process_group is not None or self.adaptive_lifespan
                or getattr(self, "enqueue_only", False) or self._evaluators_overridden()):
            return None
        ev = _as_device_events(xs, ys, ts, ps)
        if len(ev) == 0:
            return None
        blur = self.default_blur if blur_sigma is None else blur_sigma
        post = _lib.EVK_POST_MIX if self.reference_exact else _lib.EVK_POST_BLUR_IWE
        dev = ev.device
        ss = (180, 240) if self.sensor_size is None else self.sensor_size
        ch, cw = int(ss[0]) + 1, int(ss[1]) + 1
        flags = 0 if self.use_polarity else _lib.EVK_IWE_ABS_POLARITY
        t_ref = ev.t_at(-1) if self.t_ref is None else self.t_ref - ev.t_offset    # (t_ref is an ABSOLUTE time)
        w, radius = _blur_kernel(blur)

        def run(x0, xtol, gtol, ftol, maxiter, numeric_grads, unit_first):
            buf = tiled._buf("iwe_buf", 3 * ch * cw * 4, dev)
            scratch, nbytes = D.reduce_scratch(dev)
            cap = int(maxiter) + 1
            res = tiled.cmax_bfgs(ev, float(t_ref), [float(x0[0]), float(x0[1])], float(img_size[1]), float(img_size[0]), ch, cw,
                                  flags, w, radius, post, buf, D.out4(dev, 12), scratch, nbytes,
                                  [xtol, gtol, ftol, maxiter, 1.0 if numeric_grads else 0.0, 1.0 if unit_first else 0.0], cap,
                                  impl=self.impl)
            if res is None or res[5] != 0.0:
                return None
            self.native_passes = int(res[4])
            rows = res[6:6 + 5 * min(int(res[3]), cap)].reshape(-1, 5)
            return res[:2].copy(), [(r[:2].copy(), float(r[2]), r[3:5].copy()) for r in rows]
        return run

    def evaluate_numeric_gradient(self, params=None, xs=None, ys=None, ts=None, ps=None, warpfunc=None, img_size=None,
                                  blur_sigma=None, epsilon=1.0, with_value=False):
        """Forward-difference gradient of evaluate_function with absolute step `epsilon` -- exactly what
        scipy.optimize.fmin_bfgs(..., epsilon=1) estimates internally on the reference's default path
        (events_cmax.py:343: x1 = x + eps*e_i, grad_i = (f(x1) - f(x)) / (x1_i - x_i)) -- but f(x), f(x + eps e1),
        f(x + eps e2) are evaluated in ONE pass over the events.  Falls back to three separate
        evaluations when the batched kernel does not apply."""
        x0 = np.asarray(params, dtype=np.float64)
        pts = [x0.copy()]
        for i in range(len(x0)):
            x1 = x0.copy()
            x1[i] = x0[i] + epsilon
            pts.append(x1)
        blur_sigma = self.default_blur if blur_sigma is None else blur_sigma
        fs = None
        setup = self._batch3_setup(xs, ys, ts, ps, warpfunc, blur_sigma) if len(x0) == 2 else None
        if setup is not None:
            out = D.out4(setup[0].device, 12)
            res = np.empty(12, dtype=np.float64)
            if setup[2](pts, out, img_size, res):
                res = res.reshape(3, 4)
                fs = [np.float32(-res[k, 1]) for k in range(3)]
        if fs is None:
            fs = [self.evaluate_function(q, xs, ys, ts, ps, warpfunc, img_size, blur_sigma) for q in pts]
        grad = np.empty(len(x0), dtype=np.float64)
        for i in range(len(x0)):
            grad[i] = (np.float64(fs[i + 1]) - np.float64(fs[0])) / (pts[i + 1][i] - x0[i])
        return (fs[0], grad) if with_value else grad

    def evaluate_function_and_numeric_gradient(self, params=None, xs=None, ys=None, ts=None, ps=None, warpfunc=None,
                                               img_size=None, blur_sigma=None, epsilon=1.0):
        """(evaluate_function(params), evaluate_numeric_gradient(params)): f(x) is one of the three values the
        forward differences need anyway, so a line-search trial point costs ONE pass over the events."""
        return self.evaluate_numeric_gradient(params, xs, ys, ts, ps, warpfunc, img_size, blur_sigma, epsilon,
                                              with_value=True)

    def evaluate_function_batch(self, params_list, xs=None, ys=None, ts=None, ps=None, warpfunc=None, img_size=None,
                                blur_sigma=None):
        """f at K flows on the same resident events: consecutive flows are grouped in threes and each group costs ONE
        pass over the events (evk_cmax_variance_batch3_tiled_f32: every flow's plane has its own LDS window origin, so the
        flows of a group may lie anywhere -- round 4; until then only flows within 6 px of displacement shared a pass); all
        passes are enqueued back to back and the K results come back in a single readback.  Groups the batched kernel
        cannot take (direct-kernel regime, flows too large for the LDS windows) are evaluated one flow at a time.
        `self.batch_passes` counts the event passes of the last call."""
        pts = [np.asarray(q, dtype=np.float64) for q in params_list]
        K = len(pts)
        vals = [None] * K
        blur = self.default_blur if blur_sigma is None else blur_sigma
        setup = self._batch3_setup(xs, ys, ts, ps, warpfunc, blur) if K and all(len(q) == 2 for q in pts) else None
        if setup is not None and K <= 3:
            # one trio (a line search's three step lengths): the call brings its 12 doubles to the host itself -- no
            # allocation, no copy command, no stream synchronisation (as _one_call)
            ev, t_ref, launch = setup
            idx = [min(k, K - 1) for k in range(3)]
            res = self.__dict__.setdefault("_res12", np.empty(12, dtype=np.float64))
            done = []
            if launch([pts[i] for i in idx], D.out4(ev.device, 12), img_size, res):
                done.append((0, idx))
                r4 = res.reshape(3, 4)
                for k, i in enumerate(idx):
                    vals[i] = np.float32(-r4[k, 1])
        elif setup is not None:
            ev, t_ref, launch = setup
            out = torch.empty(4 * 3 * ((K + 2) // 3), dtype=torch.float64, device=ev.device)
            done = []
            for c in range(0, K, 3):
                idx = [min(c + k, K - 1) for k in range(3)]
                trio = [pts[i] for i in idx]
                if launch(trio, out[4 * c:4 * c + 12], img_size):
                    done.append((c, idx))
            if done:
                res = out.cpu().numpy().reshape(-1, 4)
                for c, idx in done:
                    for k, i in enumerate(idx):
                        vals[i] = np.float32(-res[c + k, 1])
        self.batch_passes = len(done) if setup is not None else 0
        for i in range(K):
            if vals[i] is None:
                self.batch_passes += 1
                vals[i] = self.evaluate_function(pts[i], xs, ys, ts, ps, warpfunc, img_size, blur_sigma)
        return vals

    def evaluate_function_and_gradient(self, params=None, xs=None, ys=None, ts=None, ps=None, warpfunc=None,
                                       img_size=None, blur_sigma=None):
        """(evaluate_function(params), evaluate_gradient(params)) from ONE pass over the events: both come from the
        same IWE / dIWE, and a BFGS line search asks for both at every trial point (events_cmax.py:345; scipy's
        phi / derphi).  Values identical to the two separate calls."""
        dev = D.require_gpu()
        blur_sigma = self.default_blur if blur_sigma is None else blur_sigma
        flags = _lib.EVK_POST_MIX if self.reference_exact else _lib.EVK_POST_BLUR_IWE
        res = self._one_call(params, xs, ys, ts, ps, warpfunc, img_size, blur_sigma, True, flags | _lib.EVK_POST_VALUE)
        if res is None:
            iwe, d_iwe = self._iwe(params, xs, ys, ts, ps, warpfunc, img_size, True)
            w, radius = _blur_kernel(blur_sigma)
            iwe, d_iwe = iwe.contiguous(), d_iwe.contiguous()
            out, (scratch, nbytes) = D.out4(dev), D.reduce_scratch(dev)
            _lib.call("evk_objective_variance_fg_f32", D.ptr(iwe), D.ptr(d_iwe), iwe.shape[0], iwe.shape[1],
                      D.host_ptr(w) if w is not None else None, radius, flags, D.ptr(out), D.ptr(scratch), nbytes,
                      D.stream())
            res = out.cpu().numpy()
        return np.float32(-res[3]), -(res[:2].astype(np.float32))

    def evaluate_gradient(self, params=None, xs=None, ys=None, ts=None, ps=None, warpfunc=None, img_size=None,
                          blur_sigma=None, showimg=False, iwe=None, d_iwe=None):
        """-mean(2 (iwe-mean(iwe)) * blur(d_iwe)[i]) (objectives.py:238-264).  reference_exact keeps Q4 (3-D blur mixes
        the two channels) and Q5 (IWE is NOT blurred here)."""
        dev = D.require_gpu()
        blur_sigma = self.default_blur if blur_sigma is None else blur_sigma
        flags = 1 if self.reference_exact else 2       # EVK_POST_MIX (Q4) | EVK_POST_BLUR_IWE (consistent gradient)
        if iwe is None or d_iwe is None:
            res = self._one_call(params, xs, ys, ts, ps, warpfunc, img_size, blur_sigma, True, flags)
            if res is not None:
                return -(res[:2].astype(np.float32))
            iwe, d_iwe = self._iwe(params, xs, ys, ts, ps, warpfunc, img_size, True)
        else:
            iwe, d_iwe = D.to_device(iwe, torch.float32, dev), D.to_device(d_iwe, torch.float32, dev)
        w, radius = _blur_kernel(blur_sigma)
        if d_iwe.shape[0] != 2:
            raise ValueError("d_iwe must have 2 channels (the reference hard-codes 2, image.py:210)")
        iwe, d_iwe = iwe.contiguous(), d_iwe.contiguous()
        out, (scratch, nbytes) = D.out4(dev), D.reduce_scratch(dev)
        _lib.call("evk_objective_variance_grad_f32", D.ptr(iwe), D.ptr(d_iwe), iwe.shape[0], iwe.shape[1],
                  D.host_ptr(w) if w is not None else None, radius, flags, D.ptr(out), D.ptr(scratch), nbytes,
                  D.stream())
        g = out[:2].cpu().numpy()
        return -(g.astype(np.float32))


# ---------------------------------------------------------------------------------------------------------------
# The other objectives of the reference (objectives.py:266-596): same IWE, a different scalar reduction.
# Upstream these classes skip objective_function.__init__ (so e.g. soe has no pixel_crossings and cannot go through
# optimize()); here they all inherit the full base state, which is a superset of the upstream behaviour.
# Not provided: zhu_timestamp_objective (calls the undefined events_to_zhu_timestamp_image, :545).
# ---------------------------------------------------------------------------------------------------------------
class _reduction_objective(objective_function):
    def _stats(self, params, xs, ys, ts, ps, warpfunc, img_size, blur_sigma, iwe, p=0.0, thresh=0.0):
        """[mean, var, sum v, sum v^2, sum exp v, sum exp(-p v), count(v > thresh), max v] of the blurred IWE."""
        dev = D.require_gpu()
        if iwe is None:
            iwe, _ = self._iwe(params, xs, ys, ts, ps, warpfunc, img_size, False)
        else:
            iwe = D.to_device(iwe, torch.float32, dev)
        blur_sigma = self.default_blur if blur_sigma is None else blur_sigma
        w, radius = _blur_kernel(blur_sigma)
        iwe = iwe.contiguous()
        out, (scratch, nbytes) = D.out4(dev, 8), D.reduce_scratch(dev)
        _lib.call("evk_objective_stats_f32", D.ptr(iwe), iwe.shape[0], iwe.shape[1],
                  D.host_ptr(w) if w is not None else None, radius, float(p), float(thresh), D.ptr(out), D.ptr(scratch),
                  nbytes, D.stream())
        return out.cpu().numpy(), iwe.numel()

    def _gradsums(self, params, xs, ys, ts, ps, warpfunc, img_size, blur_sigma, iwe, d_iwe, gfun, gparam, blur_iwe):
        """(sum g(a) d0, sum g(a) d1, number of pixels) with d = 3-D blurred dIWE (Q4) and a = raw or blurred IWE."""
        dev = D.require_gpu()
        if iwe is None or d_iwe is None:
            iwe, d_iwe = self._iwe(params, xs, ys, ts, ps, warpfunc, img_size, True)
        else:
            iwe, d_iwe = D.to_device(iwe, torch.float32, dev), D.to_device(d_iwe, torch.float32, dev)
        blur_sigma = self.default_blur if blur_sigma is None else blur_sigma
        w, radius = _blur_kernel(blur_sigma)
        iwe, d_iwe = iwe.contiguous(), d_iwe.contiguous()
        out, (scratch, nbytes) = D.out4(dev, 8), D.reduce_scratch(dev)
        flags = 1 | (2 if blur_iwe else 0)
        _lib.call("evk_objective_gradsums_f32", D.ptr(iwe), D.ptr(d_iwe), iwe.shape[0], iwe.shape[1],
                  D.host_ptr(w) if w is not None else None, radius, flags, gfun, float(gparam), D.ptr(out),
                  D.ptr(scratch), nbytes, D.stream())
        r = out.cpu().numpy()
        return np.array([r[6], r[7]]), iwe.numel()


class sos_objective(_reduction_objective):
    """Sum of squares (reference: objectives.py:308-353): -mean(blur(iwe)^2); gradient -mean(blur3d(d_iwe)[i] * 2 iwe)
    with the un-blurred IWE."""

    def __init__(self, adaptive_lifespan=False, minimum_events=10000):
        super().__init__(name="sos", use_polarity=True, has_derivative=True, default_blur=1.0,
                         adaptive_lifespan=adaptive_lifespan, pixel_crossings=5, minimum_events=minimum_events)
        self.current_num_events = minimum_events
        self.div = 1

    def evaluate_function(self, params=None, xs=None, ys=None, ts=None, ps=None, warpfunc=None, img_size=None,
                          blur_sigma=None, showimg=False, iwe=None):
        st, n = self._stats(params, xs, ys, ts, ps, warpfunc, img_size, blur_sigma, iwe)
        return np.float32(-(st[3] / n) / (self.div * self.div))

    def evaluate_gradient(self, params=None, xs=None, ys=None, ts=None, ps=None, warpfunc=None, img_size=None,
                          blur_sigma=None, showimg=False, iwe=None, d_iwe=None):
        g, n = self._gradsums(params, xs, ys, ts, ps, warpfunc, img_size, blur_sigma, iwe, d_iwe, 0, 0.0, False)
        return -(2.0 * g / n / (self.div * self.div)).astype(np.float32)


class rms_objective(_reduction_objective):
    """"Root mean squared" objective (reference: objectives.py:266-306).  As written upstream the loss is
    -norm(blur(iwe), 2)^2 / pixels with np.linalg.norm(., 2) of a MATRIX, i.e. its largest singular value (:282), not
    the Frobenius norm; reproduced as is.  The blurred IWE stays on the device and its spectral norm comes from a Lanczos
    iteration on the Gram operator (evk_spectral_norm_sq_f32).  Gradient: -2 mean(iwe * blur3d(d_iwe)[i]), un-blurred IWE."""

    def __init__(self):
        super().__init__(name="rms", use_polarity=True, has_derivative=True, default_blur=1.0)

    def evaluate_function(self, params=None, xs=None, ys=None, ts=None, ps=None, warpfunc=None, img_size=None,
                          blur_sigma=None, showimg=False, iwe=None):
        dev = D.require_gpu()
        if iwe is None:
            iwe, _ = self._iwe(params, xs, ys, ts, ps, warpfunc, img_size, False)
        else:
            iwe = D.to_device(iwe, torch.float32, dev)
        blur_sigma = self.default_blur if blur_sigma is None else blur_sigma
        if blur_sigma > 0:
            iwe = gaussian_filter_device(iwe.contiguous(), blur_sigma)
        # largest singular value squared = largest eigenvalue of the Gram operator: a Lanczos iteration in float64 in one
        # workgroup (evk_spectral.hip; round 5: until then torch formed the Gram matrix and rocSOLVER its eigenvalues -- and the
        # device SVD before that did not converge on the nearly rank-one images of a handful of events)
        iwe = iwe.contiguous()
        h, w = int(iwe.shape[0]), int(iwe.shape[1])
        nbytes = int(_lib.lib().evk_spectral_scratch_bytes(h, w))
        if nbytes <= 0:
            raise ValueError("rms_objective: images beyond 4096 pixels a side are not supported")
        scratch = tiled._buf("spectral", nbytes, dev)
        out = D.out4(dev)
        _lib.call("evk_spectral_norm_sq_f32", D.ptr(iwe), h, w, D.ptr(out), D.ptr(scratch), nbytes, D.stream())
        res = out[:2].cpu().numpy()
        norm2 = max(float(res[0]), 0.0)
        if res[1] > 1e-6:   # (float32 result: 1e-6 relative is below its rounding; the kernel aims for 1e-10)
            import warnings
            warnings.warn("rms_objective: the spectral-norm iteration stopped with a relative residual of %.2e (a lower bound of "
                          "sigma_max^2 is returned)" % float(res[1]), RuntimeWarning)
        return np.float32(-norm2 / (iwe.shape[0] * iwe.shape[1]))

    def evaluate_gradient(self, params=None, xs=None, ys=None, ts=None, ps=None, warpfunc=None, img_size=None,
                          blur_sigma=None, showimg=False, iwe=None, d_iwe=None):
        g, n = self._gradsums(params, xs, ys, ts, ps, warpfunc, img_size, blur_sigma, iwe, d_iwe, 0, 0.0, False)
        return -(2.0 * g / n)


class soe_objective(_reduction_objective):
    """Sum of exponentials (reference: objectives.py:355-399): -mean(exp(blur(iwe))), |polarity|, default blur 2.5."""

    def __init__(self):
        super().__init__(name="soe", use_polarity=False, has_derivative=True, default_blur=2.5)

    def evaluate_function(self, params=None, xs=None, ys=None, ts=None, ps=None, warpfunc=None, img_size=None,
                          blur_sigma=None, showimg=False, iwe=None):
        st, n = self._stats(params, xs, ys, ts, ps, warpfunc, img_size, blur_sigma, iwe)
        return -(st[4] / n)

    def evaluate_gradient(self, params=None, xs=None, ys=None, ts=None, ps=None, warpfunc=None, img_size=None,
                          blur_sigma=None, showimg=False, iwe=None, d_iwe=None):
        g, n = self._gradsums(params, xs, ys, ts, ps, warpfunc, img_size, blur_sigma, iwe, d_iwe, 1, 0.0, True)
        return -(g / n)


class moa_objective(_reduction_objective):
    """Max of accumulations (reference: objectives.py:401-426): -max(blur(iwe)); no analytic derivative."""

    def __init__(self):
        super().__init__(name="moa", use_polarity=False, has_derivative=False, default_blur=3.0)

    def evaluate_function(self, params=None, xs=None, ys=None, ts=None, ps=None, warpfunc=None, img_size=None,
                          blur_sigma=None, showimg=False, iwe=None):
        st, _ = self._stats(params, xs, ys, ts, ps, warpfunc, img_size, blur_sigma, iwe)
        return np.float32(-st[7])

    def evaluate_gradient(self, iwe=None, d_iwe=None, blur_sigma=None, showimg=False):
        return None


class isoa_objective(_reduction_objective):
    """Inverse sum of accumulations (reference: objectives.py:428-474): +count(blur(iwe) > thresh) (positive, as
    upstream); gradient -sum(blur3d(d_iwe)[i] * [blur(iwe) > thresh])."""

    def __init__(self, thresh=0.5):
        super().__init__(name="isoa", use_polarity=False, has_derivative=True, default_blur=1.0)
        self.thresh = thresh

    def evaluate_function(self, params=None, xs=None, ys=None, ts=None, ps=None, warpfunc=None, img_size=None,
                          blur_sigma=None, showimg=False, iwe=None):
        st, _ = self._stats(params, xs, ys, ts, ps, warpfunc, img_size, blur_sigma, iwe, thresh=self.thresh)
        return np.int64(round(st[6]))

    def evaluate_gradient(self, params=None, xs=None, ys=None, ts=None, ps=None, warpfunc=None, img_size=None,
                          blur_sigma=None, showimg=False, iwe=None, d_iwe=None):
        g, _ = self._gradsums(params, xs, ys, ts, ps, warpfunc, img_size, blur_sigma, iwe, d_iwe, 2, self.thresh, True)
        return -(g.astype(np.float32))


class sosa_objective(_reduction_objective):
    """Sum of suppressed accumulations (reference: objectives.py:476-520): -sum(exp(-p blur(iwe)));
    gradient -sum(blur3d(d_iwe)[i] * (-p exp(-p blur(iwe))))."""

    def __init__(self, p=3):
        super().__init__(name="sosa", use_polarity=False, has_derivative=True, default_blur=2.0)
        self.p = p

    def evaluate_function(self, params=None, xs=None, ys=None, ts=None, ps=None, warpfunc=None, img_size=None,
                          blur_sigma=None, showimg=False, iwe=None):
        st, _ = self._stats(params, xs, ys, ts, ps, warpfunc, img_size, blur_sigma, iwe, p=self.p)
        return -st[5]

    def evaluate_gradient(self, params=None, xs=None, ys=None, ts=None, ps=None, warpfunc=None, img_size=None,
                          blur_sigma=None, showimg=False, iwe=None, d_iwe=None):
        g, _ = self._gradsums(params, xs, ys, ts, ps, warpfunc, img_size, blur_sigma, iwe, d_iwe, 3, self.p, True)
        return -(-self.p * g)


class r1_objective(_reduction_objective):
    """R1 (reference: objectives.py:560-596): -sos*sosa, or -sos while sosa keeps growing (stateful last_sosa)."""

    def __init__(self, p=3):
        super().__init__(name="r1", use_polarity=False, has_derivative=False, default_blur=1.0)
        self.p = p
        self.last_sosa = 0

    def evaluate_function(self, params=None, xs=None, ys=None, ts=None, ps=None, warpfunc=None, img_size=None,
                          blur_sigma=None, showimg=False, iwe=None):
        st, n = self._stats(params, xs, ys, ts, ps, warpfunc, img_size, blur_sigma, iwe, p=self.p)
        sos, sosa = st[3] / n, st[5]
        if sosa > self.last_sosa:
            return -sos
        self.last_sosa = sosa
        return -sos * sosa

    def evaluate_gradient(self, params=None, xs=None, ys=None, ts=None, ps=None, warpfunc=None, img_size=None,
                          blur_sigma=None, showimg=False, iwe=None, d_iwe=None):
        return None
