"""
Event -> image accumulation on MI355X.  Same names, positional order, defaults, return dtypes and exception types as
the reference's lib/representations/image.py; the per-event work runs in libevk.so (HIP), there is no CPU path.

Reference citations are file:line in the reference checkout (lib/representations/image.py unless stated).
"""
import numpy as np
import torch

from .. import _device as D
from .. import _lib

_INF = float("inf")


def _is_int_tensor(t):
    return t.dtype in (torch.int64, torch.int32, torch.int16, torch.int8, torch.uint8)


def _clip_thresholds(img_size, clip_out_of_range, interpolation, padding):
    """image.py:73-74 / :193-194."""
    if not clip_out_of_range:
        return _INF, _INF
    clipx = img_size[1] if interpolation is None and padding == False else img_size[1] - 1  # noqa: E712
    clipy = img_size[0] if interpolation is None and padding == False else img_size[0] - 1  # noqa: E712
    return float(clipx), float(clipy)


def events_to_image(xs, ys, ps, sensor_size=(180, 240), interpolation=None, padding=False, meanval=False, default=0):
    """
    Place events into an image (reference: image.py:5-44).  numpy in, float64 numpy out.
    Nearest branch (:28-41): accumulate on the (H+1, W+1) canvas, crop to (H, W) (:44); integer weights are
    accumulated in int32 on the device, so the result is bit-exact with np.bincount.  Integer coordinates are
    required (TypeError otherwise), coordinates outside the canvas raise ValueError (:30-36).
    Bilinear branch (:18-27): defers to events_to_image_torch WITHOUT sensor_size (always (180, 240)), as upstream.
    """
    img_size = (sensor_size[0] + 1, sensor_size[1] + 1)
    if interpolation == 'bilinear':
        xt, yt, pt = (torch.from_numpy(np.ascontiguousarray(a)).float() for a in (xs, ys, ps))
        img = events_to_image_torch(xt, yt, pt, clip_out_of_range=True, interpolation='bilinear', padding=padding)
        img[img == 0] = default
        img = img.numpy()
        if meanval:
            event_count_image = events_to_image_torch(xt, yt, torch.ones_like(xt), clip_out_of_range=True,
                                                      padding=padding).numpy()
    else:
        xs, ys, ps = np.asarray(xs), np.asarray(ys), np.asarray(ps)
        if not (np.issubdtype(xs.dtype, np.integer) and np.issubdtype(ys.dtype, np.integer)):
            raise TypeError("only int indices permitted")      # np.ravel_multi_index, image.py:31
        dev = D.require_gpu()
        from .. import tiled
        n = xs.shape[0]
        xd, yd = D.to_device(xs, torch.int32), D.to_device(ys, torch.int32)
        oob = D.OobCounter(dev)
        impl = tiled.default_impl()
        int_w = np.issubdtype(ps.dtype, np.integer) or ps.dtype == np.bool_

        def count_image(wcol):
            """int32 event image of the int32 weight column (None: the count image): the one-pass partition + LDS tiles
            (evk_image2.hip) above the crossover, else one global int32 atomic per event -- bit-identical either way."""
            if tiled.can_tile_image((xd, yd, wcol), impl):
                c = torch.empty(img_size, dtype=torch.int32, device=dev)
                if tiled.image2("i32", xd, yd, wcol, n, img_size[0], img_size[1], 0.0, 0.0, c, oob, fresh=True):
                    return c
            c = torch.zeros(img_size, dtype=torch.int32, device=dev)
            _lib.call("evk_image_nearest_i32", D.ptr(xd), D.ptr(yd), D.ptr(wcol), n, img_size[0], img_size[1], D.ptr(c),
                      oob.ptr, D.stream())
            return c

        # (two reductions without temporaries: abs(astype(int64)).max() took 18 ms at 10 M events, min() and max() take 3)
        if int_w and (n == 0 or float(max(abs(int(ps.min())), abs(int(ps.max())))) * n < 2 ** 31):
            wd_ = D.to_device(ps, torch.int32)      # keep every device temporary alive until the launch
            canvas = count_image(wd_)
        else:
            canvas = torch.zeros(img_size, dtype=torch.float64, device=dev)
            wd_ = D.to_device(ps, torch.float64)
            _lib.call("evk_image_nearest_f64", D.ptr(xd), D.ptr(yd), D.ptr(wd_), n,
                      img_size[0], img_size[1], D.ptr(canvas), oob.ptr, D.stream())
        if meanval:
            event_count_image = count_image(None).cpu().numpy().astype(np.float64)
        oob.raise_if_set(ValueError, "events outside the (H+1, W+1) canvas %s" % (img_size,))
        img = canvas.cpu().numpy().astype(np.float64)
    if meanval:
        img = np.divide(img, event_count_image, out=np.ones_like(img) * default, where=event_count_image != 0)
    return img[0:sensor_size[0], 0:sensor_size[1]]


def events_to_image_torch(xs, ys, ps, device=None, sensor_size=(180, 240), clip_out_of_range=True,
                          interpolation=None, padding=True, default=0):
    """
    Event tensors -> float32 image tensor on `device` (reference: image.py:46-100), nearest or bilinear.
    Quirks kept: nearest-branch clipping moves rejected events to pixel (0,0) with their weight intact (Q8, :93-95);
    with the default padding=True, interpolation=None the thresholds are W-1 / H-1 (:73-74); float coordinates are
    truncated toward zero (.long(), :88-91); out-of-range indices raise IndexError (:96-99).
    """
    if device is None:
        device = xs.device
    if interpolation == 'bilinear' and padding:
        img_size = (sensor_size[0] + 1, sensor_size[1] + 1)
    else:
        img_size = list(sensor_size)
    dev = D.require_gpu()
    clipx, clipy = _clip_thresholds(img_size, clip_out_of_range, interpolation, padding)
    if ps.dtype == torch.float64:
        raise RuntimeError("Index put requires the source and destination dtypes match, got Float for the "
                           "destination and Double for the source.")
    n = xs.shape[0]
    from .. import tiled
    oob = D.OobCounter(dev)
    xd, yd = D.to_device(xs, torch.float32), D.to_device(ys, torch.float32)   # ints < 2^24 are exact in f32
    pd = D.to_device(ps.squeeze() if ps.dim() > 1 else ps, torch.float32)
    bilinear = interpolation == 'bilinear' and not _is_int_tensor(xs)
    if not bilinear and ps.dtype != torch.float32:
        raise RuntimeError("Index put requires the source and destination dtypes match, got Float for the "
                           "destination and %s for the source." % str(ps.dtype))
    # Above the crossover: one-pass partition + LDS tiles (evk_image2.hip); below it, or for columns it cannot take
    # (unaligned views), one global atomic per contribution (evk_scatter.hip).  Same semantics either way.
    img = None
    if xd.shape == pd.shape:
        xd, yd, pd = tiled.realign((xd, yd, pd), tiled.default_impl(), 4 if bilinear else 1)    # (device slices: tiled.realign)
    if tiled.can_tile_image((xd, yd, pd), tiled.default_impl(), bilinear) and xd.shape == pd.shape:
        fresh = (not bilinear) and float(default) == 0.0
        if fresh:           # every pixel is written: no memset
            img = torch.empty(tuple(img_size), dtype=torch.float32, device=dev)
        else:
            img = torch.full(tuple(img_size), float(default), dtype=torch.float32, device=dev)
        if not tiled.image2("bilinear" if bilinear else "f32", xd, yd, pd, n, img_size[0], img_size[1], clipx, clipy, img,
                            oob, fresh=fresh):
            img = None
    if img is None:
        img = torch.full(tuple(img_size), float(default), dtype=torch.float32, device=dev)
        _lib.call("evk_image_bilinear_f32" if bilinear else "evk_image_nearest_f32", D.ptr(xd), D.ptr(yd), D.ptr(pd), n,
                  img_size[0], img_size[1], clipx, clipy, D.ptr(img), oob.ptr, D.stream())
    # events and image that stay on the device: the default raises before returning, as upstream (one-pass path: the call
    # waits for its partition kernel's report only); EVK_ERRORS=deferred never waits for the host (the reference's own CUDA
    # path reports an out-of-range index_put_ asynchronously too)
    resident = xs.is_cuda and torch.device(device).type == "cuda"
    oob.raise_if_set(IndexError, "index out of range for image of size %s" % (tuple(img_size),), deferrable=resident)
    return img.to(device)


def interpolate_to_image(pxs, pys, dxs, dys, weights, img):
    """Accumulate with bilinear weights into `img` IN PLACE (reference: image.py:102-115).  Tensors in, `img` is
    returned; a CPU `img` is round-tripped through the GPU."""
    dev = D.require_gpu()
    work = img if img.is_cuda else img.to(dev)
    work = work if work.is_contiguous() else work.contiguous()
    oob = D.OobCounter(dev)
    a = [D.to_device(pxs, torch.int64), D.to_device(pys, torch.int64), D.to_device(dxs, torch.float32),
         D.to_device(dys, torch.float32), D.to_device(weights, torch.float32)]
    n = pxs.shape[0]
    from .. import tiled
    impl = tiled.default_impl()
    if all(c.dim() == 1 and c.shape[0] == n for c in a):
        a = list(tiled.realign(tuple(a), impl, 4))                                                 # (device slices: tiled.realign)
    # (round 6) the one-pass partition + LDS windows for events whose pixel + fraction is a float32 coordinate -- what every
    # upstream caller passes --, the direct kernel's four global atomics for the others, decided per event by the partition kernel
    fast = (work.dtype == torch.float32 and work.dim() == 2 and impl in ("tiled", "auto")
            and 0 < n <= 4_000_000_000 and (impl == "tiled" or n >= tiled.TILED_MIN_EVENTS_SPLAT_INDEXED)
            and all(c.dim() == 1 and c.shape[0] == n and tiled.column_ok(c) for c in a)
            and tiled.splat_indexed2(*a, n, work.shape[0], work.shape[1], work, oob))
    if not fast:
        _lib.call("evk_splat_indexed_f32", D.ptr(a[0]), D.ptr(a[1]), D.ptr(a[2]), D.ptr(a[3]), D.ptr(a[4]), n, work.shape[0],
                  work.shape[1], D.ptr(work), oob.ptr, D.stream())
    oob.raise_if_set(IndexError, "index out of range for image of size %s" % (tuple(img.shape),))
    if work is not img:
        img.copy_(work)
    return img


def interpolate_to_derivative_img(pxs, pys, dxs, dys, d_img, w1, w2):
    """Derivative-of-bilinear accumulate into `d_img` (C, H, W) IN PLACE (reference: image.py:117-136)."""
    dev = D.require_gpu()
    work = d_img if d_img.is_cuda else d_img.to(dev)
    work = work if work.is_contiguous() else work.contiguous()
    oob = D.OobCounter(dev)
    a = [D.to_device(pxs, torch.int64), D.to_device(pys, torch.int64), D.to_device(dxs, torch.float32),
         D.to_device(dys, torch.float32), D.to_device(w1, torch.float32), D.to_device(w2, torch.float32)]
    n = pxs.shape[0]
    from .. import tiled
    impl = tiled.default_impl()
    # (round 6) two channels, float32 image: the one-pass partition + LDS windows, the tile kernel fetching every event's four
    # weights by its index; events whose pixel + fraction is not a float32 coordinate take the direct kernel's code there
    fast = False
    if (work.dtype == torch.float32 and work.dim() == 3 and work.shape[0] == 2 and impl in ("tiled", "auto")
            and 0 < n <= 4_000_000_000 and (impl == "tiled" or n >= tiled.TILED_MIN_EVENTS_SPLAT_DRV)
            and all(c.dim() == 1 and c.shape[0] == n for c in a[:4]) and all(tuple(c.shape) == (2, n) and c.is_contiguous() for c in a[4:])):
        a[:4] = tiled.realign(tuple(a[:4]), impl, 8)
        fast = all(tiled.column_ok(c) for c in a[:4]) and tiled.splat_drv_indexed2(*a, n, work.shape[1], work.shape[2], work, oob)
    if not fast:
        _lib.call("evk_splat_drv_indexed_f32", D.ptr(a[0]), D.ptr(a[1]), D.ptr(a[2]), D.ptr(a[3]), D.ptr(a[4]), D.ptr(a[5]),
                  work.shape[0], n, work.shape[1], work.shape[2], D.ptr(work), oob.ptr, D.stream())
    oob.raise_if_set(IndexError, "index out of range for image of size %s" % (tuple(d_img.shape),))
    if work is not d_img:
        d_img.copy_(work)
    return d_img


def _events_to_image_drv_device(xn, yn, pn, jacobian_xn, jacobian_yn, sensor_size, clip_out_of_range, interpolation,
                                padding, compute_gradient):
    dev = D.require_gpu()
    img_size = (sensor_size[0] + 1, sensor_size[1] + 1) if padding else tuple(sensor_size)
    clipx, clipy = _clip_thresholds(img_size, clip_out_of_range, interpolation, padding)
    n = len(xn)
    img = torch.zeros(img_size, dtype=torch.float32, device=dev)
    d_img = torch.zeros((2,) + tuple(img_size), dtype=torch.float32, device=dev) if compute_gradient else None
    jx = D.to_device(jacobian_xn, torch.float64) if compute_gradient else None
    jy = D.to_device(jacobian_yn, torch.float64) if compute_gradient else None
    oob = D.OobCounter(dev)
    xd, yd, pd = D.to_device(xn, torch.float64), D.to_device(yn, torch.float64), D.to_device(pn, torch.float64)
    from .. import tiled
    impl = tiled.default_impl()
    # (round 6) the one-pass partition + LDS windows (the image and its two derivative planes), the tile kernel fetching p and
    # the Jacobians by the event's index; masked / wrapping / out-of-range events take the direct kernel's code there
    fast = (impl in ("tiled", "auto") and 0 < n <= 4_000_000_000 and (impl == "tiled" or n >= tiled.TILED_MIN_EVENTS_SPLAT_DRV)
            and all(c.dim() == 1 and c.shape[0] == n and c.is_contiguous() and c.data_ptr() % 16 == 0 for c in (xd, yd, pd))
            and (jx is None or all(tuple(c.shape) == (2, n) and c.is_contiguous() for c in (jx, jy)))
            and tiled.image_drv2(xd, yd, pd, jx, jy, n, img_size[0], img_size[1], clipx, clipy, img, d_img, oob))
    if not fast:
        _lib.call("evk_image_drv_f64", D.ptr(xd), D.ptr(yd), D.ptr(pd), D.ptr(jx), D.ptr(jy), n, img_size[0], img_size[1], clipx,
                  clipy, D.ptr(img), D.ptr(d_img), oob.ptr, D.stream())
    oob.raise_if_set(IndexError, "index out of range for image of size %s" % (img_size,))
    return img, d_img


def events_to_image_drv(xn, yn, pn, jacobian_xn, jacobian_yn, device=None, sensor_size=(180, 240),
                        clip_out_of_range=True, interpolation='bilinear', padding=True, compute_gradient=False):
    """
    Events (+ per-event Jacobians) -> IWE and dIWE (reference: image.py:162-217).  float64 numpy in; float32 numpy
    out, padded and un-cropped: (H+1, W+1) and (2, H+1, W+1) (or None).  Coordinates are cast to float32 before
    floor/frac (Q7, :179-183).
    """
    img, d_img = _events_to_image_drv_device(xn, yn, pn, jacobian_xn, jacobian_yn, sensor_size, clip_out_of_range,
                                             interpolation, padding, compute_gradient)
    return img.cpu().numpy(), (d_img.cpu().numpy() if d_img is not None else None)


def image_to_event_weights(xs, ys, img):
    """
    Value of `img` at every event by reverse bilinear interpolation (reference: image.py:138-160); events with
    x >= W-1 or y >= H-1 get 0.  numpy in -> float64 numpy out; device tensors in -> device tensor out.
    """
    dev = D.require_gpu()
    on_device = isinstance(xs, torch.Tensor)
    xd, yd = D.to_device(xs, torch.float64, dev), D.to_device(ys, torch.float64, dev)
    # float32 images (the IWE) are read as they are; anything wider (float64, integers) as float64 -- numpy promotes
    # img[...] * weights to float64 either way, so the values that enter the products are the image's own
    if isinstance(img, torch.Tensor):
        wide = img.dtype not in (torch.float32, torch.float16, torch.bfloat16)
    else:
        wide = np.asarray(img).dtype not in (np.float32, np.float16)
    imgd = D.to_device(img, torch.float64 if wide else torch.float32, dev)
    out = torch.empty_like(xd)
    oob = D.OobCounter(dev)
    _lib.call("evk_image_gather_bilinear_f64img" if wide else "evk_image_gather_bilinear_f64", D.ptr(xd), D.ptr(yd),
              xd.shape[0], D.ptr(imgd), imgd.shape[0],
              imgd.shape[1], D.ptr(out), oob.ptr, D.stream())
    oob.raise_if_set(IndexError, "index out of range for image of size %s" % (tuple(imgd.shape),))
    return out if on_device else out.cpu().numpy()


def _timestamp_images_device(xd, yd, td, pd, img_size, clip_out_of_range, interpolation, padding, mode, ta, tdiv, resident=False):
    """(4, H, W) float32 device tensor [ts_pos, cnt_pos, ts_neg, cnt_neg]; the count planes start at ONE (upstream
    quirk, image.py:269,271: img_*_cnt = torch.ones).  ta is None (modes 0 / 1): the time constants come from td[0] / td[-1] --
    read by the one-pass kernels themselves, read back for the direct kernel.  resident: events and images stay on the device, the
    IndexError check waits for the partition kernel's report only (or not at all under EVK_ERRORS=deferred)."""
    dev = xd.device
    clipx, clipy = _clip_thresholds(img_size, clip_out_of_range, interpolation, padding)
    out = torch.empty((4,) + tuple(img_size), dtype=torch.float32, device=dev)
    _lib.call("evk_timestamp_planes_init_f32", D.ptr(out), int(img_size[0]) * int(img_size[1]), D.stream())
    oob = D.OobCounter(dev)
    from .. import tiled
    n = xd.shape[0]
    # Above the crossover: one partition + four LDS windows per tile (evk_image2.hip, round 6); below it, or for columns the
    # one-pass path cannot take (unaligned views), eight global atomics per event (evk_scatter.hip).  Same semantics either way.
    impl = tiled.default_impl()
    from_events = ta is None
    if xd.shape == yd.shape == td.shape == pd.shape:
        xd, yd, td, pd = tiled.realign((xd, yd, td, pd), impl, 8)                                 # (device slices: tiled.realign)
    fast = (tiled.can_tile_image((xd, yd, td, pd), impl, True) and xd.shape == yd.shape == td.shape == pd.shape
            and (impl == "tiled" or n >= tiled.TILED_MIN_EVENTS_TIMESTAMP)
            and tiled.timestamp_images2(xd, yd, td, pd, n, img_size[0], img_size[1], clipx, clipy, mode, 0.0 if from_events else ta,
                                        1.0 if from_events else tdiv, out, oob, from_events=from_events))
    if not fast:
        if from_events:
            ta, tdiv = _timestamp_constants(D.ends(td), mode)
        _lib.call("evk_timestamp_images_f32", D.ptr(xd), D.ptr(yd), D.ptr(td), D.ptr(pd), n, img_size[0], img_size[1],
                  clipx, clipy, mode, float(ta), float(tdiv), D.ptr(out), oob.ptr, D.stream())
    oob.raise_if_set(IndexError, "index out of range for image of size %s" % (tuple(img_size),), deferrable=resident and fast)
    return out


def _timestamp_constants(ends, mode):
    """(ta, tdiv) of events_to_timestamp_image_torch (image.py:326-329), float32 arithmetic: ta = ts[0] (reversed: ts[-1]),
    tdiv = (ts[-1] - ts[0]) + 1e-6."""
    t_first, t_last = (np.float32(e) for e in ends)
    return (t_last if mode == 1 else t_first), np.float32(np.float32(t_last - t_first) + np.float32(1e-6))


def _timestamp_finalise(planes):
    """cnt[cnt == 0] = 1; time / cnt for both classes (image.py:278-282) -> (2, H, W) float32 device tensor [pos, neg]."""
    res = torch.empty((2,) + tuple(planes.shape[1:]), dtype=torch.float32, device=planes.device)
    _lib.call("evk_timestamp_finalise_f32", D.ptr(planes), planes[0].numel(), D.ptr(res[0]), D.ptr(res[1]), D.stream())
    return res


def events_to_timestamp_image(xn, yn, ts, pn, device=None, sensor_size=(180, 240), clip_out_of_range=True,
                              interpolation='bilinear', padding=True, normalize_timestamps=True):
    """
    Average-timestamp images of the positive and non-positive events (Zhu et al.; reference: image.py:219-283).
    numpy in -> two float32 numpy images.  Timestamps are normalised as (ts - ts[0]) / (ts[-1] - ts[0]... upstream
    divides by (ts[-1] + 1e-6) of the ts[0]-shifted float32 column (:251) -- kept.
    """
    dev = D.require_gpu()
    img_size = (sensor_size[0] + 1, sensor_size[1] + 1) if padding else tuple(sensor_size)
    tsf = (np.asarray(ts) - ts[0]).astype(np.float32)                       # torch.from_numpy(ts - t0).float()
    xd, yd = D.to_device(np.asarray(xn), torch.float32, dev), D.to_device(np.asarray(yn), torch.float32, dev)
    td, pd = D.to_device(tsf, torch.float32, dev), D.to_device(np.asarray(pn), torch.float32, dev)
    if normalize_timestamps:
        mode, ta, tdiv = 0, tsf[0], np.float32(tsf[-1] + np.float32(1e-6))
    else:
        mode, ta, tdiv = 2, 0.0, 1.0
    img = _timestamp_images_device(xd, yd, td, pd, img_size, clip_out_of_range, interpolation, padding, mode, ta, tdiv)
    res = _timestamp_finalise(img).cpu().numpy()
    return res[0], res[1]


def events_to_timestamp_image_torch(xs, ys, ts, ps, device=None, sensor_size=(180, 240), clip_out_of_range=True,
                                    interpolation='bilinear', padding=True, timestamp_reverse=False):
    """
    Torch twin (reference: image.py:285-353): nts = (ts - ts[0]) / (ts[-1] - ts[0] + 1e-6), or
    (-ts + ts[-1]) / (...) with timestamp_reverse.  Tensors in -> two float32 tensors on `device`.
    """
    if device is None:
        device = xs.device
    dev = D.require_gpu()
    xs, ys, ps, ts = xs.squeeze(), ys.squeeze(), ps.squeeze(), ts.squeeze()
    img_size = (sensor_size[0] + 1, sensor_size[1] + 1) if padding else tuple(sensor_size)
    xd, yd = D.to_device(xs, torch.float32, dev), D.to_device(ys, torch.float32, dev)
    td, pd = D.to_device(ts, torch.float32, dev), D.to_device(ps, torch.float32, dev)
    mode = 1 if timestamp_reverse else 0
    if ts.is_cuda and ts.dim() == 1:
        ta = tdiv = None            # ts[0], ts[-1] are read on the device (no round trip before the launches)
    else:                           # (a host tensor: read directly)
        ta, tdiv = _timestamp_constants(D.ends(td) if ts.dim() != 1 else D.ends(ts[[0, -1]].to(torch.float32)), mode)
    resident = xs.is_cuda and torch.device(device).type == "cuda"
    img = _timestamp_images_device(xd, yd, td, pd, img_size, clip_out_of_range, interpolation, padding, mode, ta, tdiv,
                                   resident=resident)
    res = _timestamp_finalise(img)
    return res[0].to(device), res[1].to(device)


# ---- the stateful image classes (image.py:355-396) ------------------------------------------------------------------
def _f64_columns(dev, *cols):
    """Event columns of any numeric kind (lists, numpy, torch on any device) -> float64 device columns of the length zip()
    would iterate (the shortest one)."""
    n = min(len(c) for c in cols)
    out = []
    for c in cols:
        if not isinstance(c, torch.Tensor):
            c = np.asarray(c)
        out.append(D.to_device(c[:n], torch.float64, dev).reshape(-1))
    return n, out


class _DeviceImage:
    """float64 (H, W) image resident on the GPU with the reference's attributes.  Upstream's `.image` is a plain mutable
    ndarray (image.py:358,380) that callers edit in place (`obj.image[y, x] = v`, `obj.image *= k`) and keep references to; here
    `.image` hands out a host MIRROR that stays coupled to the device image for as long as the caller holds it: its contents are
    uploaded before every device-side operation of the object (so in-place edits are seen) and refreshed after every one that
    changes the image (so a kept reference shows the new events) -- one small copy each way per call, only once `.image` has
    been touched; a mirror the caller no longer references is dropped at the next call."""

    def __init__(self, sensor_size):
        self.sensor_size = sensor_size
        self.num_pixels = sensor_size[0] * sensor_size[1]
        self._dev = D.require_gpu()
        self._img = torch.ones(tuple(int(v) for v in sensor_size), dtype=torch.float64, device=self._dev)
        self._mirror = None

    def _host_mirror(self):
        return self._mirror

    def _sync_in(self):
        """Before a device-side operation: in-place edits of the handed-out mirror become the device image.  The mirror is held
        strongly until here (`obj.image[y, x] = v` edits a temporary that nobody else references); once uploaded it is kept
        only while the caller still holds a reference of their own."""
        h = self._mirror
        if h is not None:
            import sys
            self._img.copy_(torch.from_numpy(np.ascontiguousarray(h, dtype=np.float64)).reshape(self._img.shape))
            if sys.getrefcount(h) <= 3:          # self._mirror, h, getrefcount's argument: the caller dropped theirs
                self._mirror = h = None
        return h

    def _sync_out(self, h):
        """After an operation that changed the device image: the mirror a caller still holds shows it."""
        if h is not None:
            h[...] = self._img.cpu().numpy()

    @property
    def image(self):
        if self._mirror is None:
            self._mirror = self._img.cpu().numpy()
        return self._mirror

    @image.setter
    def image(self, value):
        a = np.asarray(value, dtype=np.float64)
        self._img = D.to_device(np.ascontiguousarray(a), torch.float64, self._dev)
        self._mirror = None

    @property
    def device_image(self):
        """The resident image (a torch float64 tensor; no copy).  Edits of a host mirror handed out by `.image` are uploaded
        first; a caller that writes to this tensor directly should re-read `.image` afterwards."""
        self._sync_in()
        return self._img

    def _raise(self, oob):
        oob.raise_if_set(IndexError, "index out of bounds for image of size %s" % (tuple(self._img.shape),))


class TimestampImage(_DeviceImage):
    """Time-stamp image (reference: image.py:355-375): every event writes its time stamp to pixel (int(y), int(x)) in
    stream order -- the last event of a pixel wins --, get_image() ranks the pixel values densely and scales the ranks to
    [0, 1].  The image lives on the GPU: add_events is one atomicMax per event on the event's position plus a gather,
    get_image a radix sort of the pixels.  An out-of-range event raises IndexError as upstream's assignment does (upstream
    has applied the events before it by then; here the in-range events of the whole batch are applied)."""

    def set_init(self, value):
        self._img = torch.full_like(self._img, float(value))
        self._sync_out(self._host_mirror())

    def add_event(self, x, y, t, p):
        self.add_events([x], [y], [t], None)

    def add_events(self, xs, ys, ts, ps):
        n, (xd, yd, td) = _f64_columns(self._dev, xs, ys, ts)
        if n == 0:
            return
        H, W = self._img.shape
        oob = D.OobCounter(self._dev)
        last = torch.empty(H * W, dtype=torch.int32, device=self._dev)
        mirror = self._sync_in()
        _lib.call("evk_timestamp_image_add_f64", D.ptr(xd), D.ptr(yd), D.ptr(td), n, H, W, D.ptr(self._img), D.ptr(last),
                  oob.ptr, D.stream())
        self._sync_out(mirror)
        self._raise(oob)

    def get_image(self):
        H, W = self._img.shape
        npix = H * W
        nbytes = int(_lib.lib().evk_dense_rank_scratch_bytes(npix))
        scratch = torch.empty(nbytes + 256, dtype=torch.uint8, device=self._dev)
        off = (-scratch.data_ptr()) % 256
        out = torch.empty((H, W), dtype=torch.float64, device=self._dev)
        import ctypes
        self._sync_in()
        _lib.call("evk_dense_rank_f64", D.ptr(self._img), npix, D.ptr(out), ctypes.c_void_p(scratch.data_ptr() + off), nbytes,
                  D.stream())
        return out.cpu().numpy()


class EventImage(_DeviceImage):
    """Event-count image (reference: image.py:377-396): add_event accumulates p at (int(y), int(x)); get_image() scales the
    image to [0, 1].  Upstream's add_events passes a literal 0 for the polarity (image.py:387-389), so it changes nothing
    and only fails on out-of-range indices -- reproduced by default; use_polarity=True (not upstream) accumulates `ps`."""

    def add_event(self, x, y, t, p):
        self._add([x], [y], [p])

    def add_events(self, xs, ys, ts, ps, use_polarity=False):
        if use_polarity:
            self._add(xs, ys, ps)
        else:
            n = min(len(xs), len(ys), len(ts))
            self._add(xs[:n], ys[:n], None)

    def _add(self, xs, ys, ps):
        if ps is None:
            n, (xd, yd) = _f64_columns(self._dev, xs, ys)
            pd = None
        else:
            n, (xd, yd, pd) = _f64_columns(self._dev, xs, ys, ps)
        if n == 0:
            return
        H, W = self._img.shape
        oob = D.OobCounter(self._dev)
        mirror = self._sync_in()
        _lib.call("evk_event_image_add_f64", D.ptr(xd), D.ptr(yd), D.ptr(pd), n, H, W, D.ptr(self._img), oob.ptr, D.stream())
        self._sync_out(mirror)
        self._raise(oob)

    def get_image(self):
        H, W = self._img.shape
        out = torch.empty((H, W), dtype=torch.float64, device=self._dev)
        scratch = torch.empty(int(_lib.lib().evk_minmax_scratch_bytes()) // 8, dtype=torch.float64, device=self._dev)
        self._sync_in()
        _lib.call("evk_minmax_normalise_f64", D.ptr(self._img), H * W, D.ptr(out), D.ptr(scratch), D.stream())
        return out.cpu().numpy()
