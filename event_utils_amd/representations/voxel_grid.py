"""
Event -> voxel grid on MI355X.  Mirrors the reference's lib/representations/voxel_grid.py (same names, argument
order, defaults, dtypes); the B per-bin passes of the reference are ONE pass over the events in libevk.so.
Citations: voxel_grid.py in the reference checkout unless stated.
"""
import numpy as np
import torch

from .. import _device as D
from .. import _lib


def _voxel_f32_device(xd, yd, td, pd, B, sensor_size, t_first, t_last, out=None, check=True, impl=None, fresh=None,
                      native=None, deferrable=False):
    """Device-resident core of events_to_voxel_torch.  out=None: a new (B, H, W) float32 grid is returned; else the
    events are accumulated into `out` (fresh=True: `out` is overwritten instead, no memset needed).
    native = events.NativeColumns replaces the four float32 columns (on-disk dtypes, widened in the kernels).
    t_first = None: ts[0] / ts[-1] are taken from the column on the device (no transfer before the launch).
    deferrable: the out-of-range check may be reported asynchronously (_device.error_mode)."""
    dev = xd.device if native is None else native.t.device
    H, W = int(sensor_size[0]), int(sensor_size[1])
    if out is None:
        out = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        fresh = True
    oob = D.OobCounter(dev) if check else None
    from .. import tiled
    tiled.voxel_f32(xd, yd, td, pd, None if t_first is None else float(t_first), None if t_first is None else float(t_last),
                    B, H, W, out, oob, impl=impl, fresh=bool(fresh), native=native)
    if check:
        oob.raise_if_set(IndexError, "index out of range for voxel grid of size %s" % ((B, H, W),), deferrable)
    return out


def events_to_voxel_torch(xs, ys, ts, ps, B, device=None, sensor_size=(180, 240), temporal_bilinear=True):
    """
    Events -> (B, H, W) float32 voxel grid with temporal bilinear weights (reference: voxel_grid.py:114-153).
    t_norm = (ts-ts[0])/(ts[-1]-ts[0])*(B-1) in float32 (:133-134, unguarded: dt == 0 gives NaN, Q9), weights
    ps*max(0, 1-|t_norm-b|) (:138-139), nearest-pixel accumulate with float coordinates truncated toward zero and
    out-of-range coordinates raising IndexError (:140-142 -> image.py:87-99, clip_out_of_range=False).
    Like the reference the float path needs float32 ts / ps (float64 raises RuntimeError).
    """
    if not temporal_bilinear:
        raise NotImplementedError("temporal_bilinear=False is dead code upstream (undefined names, "
                                  "voxel_grid.py:144-147)")
    from ..events import DeviceEvents
    if isinstance(xs, DeviceEvents):
        # resident events (ys, ts, ps ignored), e.g. DeviceEvents.from_native(...) holding the on-disk dtypes
        ev = xs
        if len(ev) == 0:
            raise IndexError("index -1 is out of bounds for dimension 0 with size 0")
        if ev.dtype != torch.float32:
            raise RuntimeError("events_to_voxel_torch needs float32 event columns (voxel_grid.py:138-142)")
        native = ev.native if ev._cols is None else None
        cols = (None,) * 4 if native is not None else (ev.x, ev.y, ev.t, ev.p)
        out = _voxel_f32_device(*cols, B, sensor_size, ev.t_at(0), ev.t_at(-1), native=native,
                                deferrable=device is None or torch.device(device).type == "cuda")
        return out if device is None else out.to(device)
    if device is None:
        device = xs.device
    size = lambda a: a.shape[0] if hasattr(a, "shape") and len(a.shape) else len(a)   # (len() of a tensor costs ~1 us)
    n_ev = size(xs)
    assert (n_ev == size(ys) and n_ev == size(ts) and n_ev == size(ps))
    if (xs.dtype == torch.int16 and ys.dtype == torch.int16 and ts.dtype == torch.float32 and n_ev
            and ps.dtype in (torch.uint8, torch.int8, torch.bool)):
        # int16 coordinates / 8-bit polarities as stored on disk: valid upstream too (xs.long(), ps * weights promotes
        # to float32); here they are read as they are (9 B/event) instead of being widened to four float32 columns
        ev = DeviceEvents.from_native(xs, ys, ts, ps, polarity="literal", t_offset=0.0)
        out = _voxel_f32_device(None, None, None, None, B, sensor_size, ev.t_at(0), ev.t_at(-1), native=ev.native)
        return out.to(device)
    if ts.dtype == torch.float64 or ps.dtype == torch.float64:
        raise RuntimeError("Index put requires the source and destination dtypes match, got Float for the "
                           "destination and Double for the source.")
    dev = xs.device if (xs.is_cuda and _lib._lib is not None) else D.require_gpu()   # (resident columns: the library is loaded)
    xd, yd = D.to_device(xs, torch.float32, dev), D.to_device(ys, torch.float32, dev)
    td, pd = D.to_device(ts, torch.float32, dev), D.to_device(ps, torch.float32, dev)
    if n_ev == 0:
        raise IndexError("index -1 is out of bounds for dimension 0 with size 0")   # ts[-1], voxel_grid.py:133
    # ts[0] / ts[-1] are read by the kernels themselves; events and grid that stay on the device wait for the partition
    # kernel's report only (strict, the default) or not at all (EVK_ERRORS=deferred)
    resident = xs.is_cuda and (device is xs.device or torch.device(device).type == "cuda")
    out = _voxel_f32_device(xd, yd, td, pd, B, sensor_size, None, None, deferrable=resident)
    return out if out.device == device else out.to(device)


def events_to_voxel(xs, ys, ts, ps, B, sensor_size=(180, 240), temporal_bilinear=True):
    """
    numpy events -> (B, H, W) float64 voxel grid (reference: voxel_grid.py:184-217): float64 t_norm and weights,
    integer coordinates on the (H+1, W+1) canvas of events_to_image (image.py:17,28-44): non-integer coordinates
    raise TypeError, coordinates outside the canvas ValueError, x == W / y == H fall in the cropped pad.
    """
    return _events_to_voxel_numpy(xs, ys, ts, (ps,), B, sensor_size, temporal_bilinear)[0]


def _events_to_voxel_numpy(xs, ys, ts, weight_columns, B, sensor_size, temporal_bilinear):
    """events_to_voxel for every weight column of `weight_columns` on the SAME events: x, y, t go to the device once
    (events_to_neg_pos_voxel voxelises its events twice, voxel_grid.py:240-241: host arrays, so the link is what it costs)."""
    for ps in weight_columns:
        assert (len(xs) == len(ys) and len(ys) == len(ts) and len(ts) == len(ps))
    if not temporal_bilinear:
        raise NotImplementedError("temporal_bilinear=False is dead code upstream (voxel_grid.py:213-214)")
    xs, ys = np.asarray(xs).squeeze(), np.asarray(ys).squeeze()
    ts = np.asarray(ts, dtype=np.float64)
    if not (np.issubdtype(xs.dtype, np.integer) and np.issubdtype(ys.dtype, np.integer)):
        raise TypeError("only int indices permitted")
    dev = D.require_gpu()
    H, W = int(sensor_size[0]), int(sensor_size[1])
    xd, yd = D.to_device(xs, torch.int32), D.to_device(ys, torch.int32)
    td = D.to_device(ts, torch.float64)
    grids = []
    for ps in weight_columns:
        out = torch.zeros((B, H, W), dtype=torch.float64, device=dev)
        oob = D.OobCounter(dev)
        pd = D.to_device(np.asarray(ps).squeeze(), torch.float64)
        _lib.call("evk_voxel_f64", D.ptr(xd), D.ptr(yd), D.ptr(td), D.ptr(pd), len(xs),
                  float(ts[0]), float(ts[-1]), B, H, W, D.ptr(out), oob.ptr, D.stream())
        oob.raise_if_set(ValueError, "events outside the (H+1, W+1) canvas")
        grids.append(out.cpu().numpy())
    return grids


def events_to_neg_pos_voxel_torch(xs, ys, ts, ps, B, device=None, sensor_size=(180, 240), temporal_bilinear=True):
    """Separate voxel grids of positive / non-positive events (reference: voxel_grid.py:155-182).  Above the tiled
    crossover both grids come from ONE pass over the events (EVK_VOXEL_SPLIT_POLARITY) instead of two voxelisations."""
    from ..events import DeviceEvents
    ev = xs if isinstance(xs, DeviceEvents) else None
    raw_dev = None      # the device of raw tensors wrapped below: grids go back THERE by default, as upstream (xs.device)
    if (ev is None and temporal_bilinear and all(isinstance(a, torch.Tensor) for a in (xs, ys, ts, ps)) and len(xs)
            and xs.dtype == torch.int16 and ys.dtype == torch.int16 and ts.dtype == torch.float32
            and ps.dtype in (torch.uint8, torch.int8, torch.bool)):
        # int16 coordinates / 8-bit polarities as stored on disk (valid upstream too: ps > 0 / ps <= 0 on the stored values)
        raw_dev = xs.device
        ev = DeviceEvents.from_native(xs, ys, ts, ps, polarity="literal", t_offset=0.0)
    if ev is not None:
        # resident events (ys, ts, ps ignored): the on-disk dtypes are partitioned as they are (9 / 13 B per event)
        if not temporal_bilinear:
            raise NotImplementedError("temporal_bilinear=False is dead code upstream (voxel_grid.py:144-147)")
        if len(ev) == 0:
            raise IndexError("index -1 is out of bounds for dimension 0 with size 0")
        from .. import tiled
        H, W = int(sensor_size[0]), int(sensor_size[1])
        native = ev.native if ev._cols is None else None
        oob = D.OobCounter(ev.device)
        cols = (None,) * 4 if native is not None else (ev.x, ev.y, ev.t, ev.p)
        both = tiled.voxel_neg_pos_f32(*cols, ev.t_at(0), ev.t_at(-1), B, H, W, oob, native=native)
        # where the grids go: resident DeviceEvents stay on the GPU; raw tensors get them on THEIR device (host tensors
        # therefore synchronously, errors included -- the float path below and events_to_voxel_torch do the same)
        target = device if device is not None else (raw_dev if raw_dev is not None else ev.device)
        if both is not None:
            resident = (raw_dev is None or raw_dev.type == "cuda") and torch.device(target).type == "cuda"
            oob.raise_if_set(IndexError, "index out of range for voxel grid of size %s" % ((B, H, W),), deferrable=resident)
            both = both.to(target)
            return both[0], both[1]
        x, y, t, p = ev.x, ev.y, ev.t, ev.p          # (widened once) -> the two-voxelisation route below
        out_dev = target
        pw, nw = _polarity_weights(p)
        pos = events_to_voxel_torch(x, y, t, pw, B, device=out_dev, sensor_size=sensor_size)
        neg = events_to_voxel_torch(x, y, t, nw, B, device=out_dev, sensor_size=sensor_size)
        return pos, neg
    if (temporal_bilinear and all(isinstance(a, torch.Tensor) for a in (xs, ys, ts, ps)) and len(xs)
            and ts.dtype != torch.float64 and ps.dtype != torch.float64):
        from .. import tiled
        dev = D.require_gpu()
        cols = [D.to_device(a, torch.float32, dev) for a in (xs, ys, ts, ps)]
        H, W = int(sensor_size[0]), int(sensor_size[1])
        oob = D.OobCounter(dev)
        both = tiled.voxel_neg_pos_f32(*cols, None, None, B, H, W, oob)     # (ts[0], ts[-1] are read on the device)
        if both is not None:
            # events and grids that stay on the device never wait for the host (as events_to_voxel_torch: _device.error_mode)
            resident = xs.is_cuda and torch.device(xs.device if device is None else device).type == "cuda"
            oob.raise_if_set(IndexError, "index out of range for voxel grid of size %s" % ((B, H, W),), deferrable=resident)
            both = both.to(xs.device if device is None else device)
            return both[0], both[1]
    pos_weights, neg_weights = _polarity_weights(ps)
    voxel_pos = events_to_voxel_torch(xs, ys, ts, pos_weights, B, device=device, sensor_size=sensor_size,
                                      temporal_bilinear=temporal_bilinear)
    voxel_neg = events_to_voxel_torch(xs, ys, ts, neg_weights, B, device=device, sensor_size=sensor_size,
                                      temporal_bilinear=temporal_bilinear)
    return voxel_pos, voxel_neg


def _polarity_weights(ps):
    """The two weight columns of voxel_grid.py:173-174 -- torch.where(ps > 0, 1.0, 0.0), torch.where(ps <= 0, 1.0, 0.0) as
    float32 -- from one kernel (evk_polarity_weights_f32), returned where `ps` lives."""
    dev = D.require_gpu()
    if isinstance(ps, torch.Tensor) and ps.dtype == torch.float64:      # (upstream's dtype error is raised by the voxelisation)
        pd = D.to_device(ps, torch.float64, dev).to(torch.float32)
    else:
        pd = D.to_device(ps, torch.float32, dev)
    pos, neg = torch.empty_like(pd), torch.empty_like(pd)
    _lib.call("evk_polarity_weights_f32", D.ptr(pd), pd.numel(), D.ptr(pos), D.ptr(neg), D.stream())
    if isinstance(ps, torch.Tensor) and not ps.is_cuda:
        return pos.to(ps.device), neg.to(ps.device)
    return pos, neg


def events_to_neg_pos_voxel(xs, ys, ts, ps, B, sensor_size=(180, 240), temporal_bilinear=True):
    """numpy twin (reference: voxel_grid.py:219-243): np.where(ps, 1, 0) / np.where(ps, 0, 1) weights."""
    pos_weights = np.where(ps, 1, 0)
    neg_weights = np.where(ps, 0, 1)
    voxel_pos, voxel_neg = _events_to_voxel_numpy(xs, ys, ts, (pos_weights, neg_weights), B, sensor_size, temporal_bilinear)
    return voxel_pos, voxel_neg


def _voxel_windows(xs, ys, ts, ps, B, bounds, sensor_size):
    """All windows [bounds[k], bounds[k+1]) of the stream in ONE launch (evk_voxel_segments_f32) -> list of (B, H, W)
    float32 tensors on xs.device (views of one (S, B, H, W) allocation)."""
    device = xs.device
    if ts.dtype == torch.float64 or ps.dtype == torch.float64:
        raise RuntimeError("Index put requires the source and destination dtypes match, got Float for the "
                           "destination and Double for the source.")
    nseg = len(bounds) - 1
    if nseg <= 0:
        return []
    dev = D.require_gpu()
    H, W = int(sensor_size[0]), int(sensor_size[1])
    xd, yd = D.to_device(xs, torch.float32, dev), D.to_device(ys, torch.float32, dev)
    td, pd = D.to_device(ts, torch.float32, dev), D.to_device(ps, torch.float32, dev)
    bounds = np.asarray(bounds, dtype=np.int64)
    from .. import tiled
    impl = tiled.default_impl()
    if (impl != "direct" and (impl == "tiled" or int(np.min(np.diff(bounds))) >= _WINDOW_MIN_EVENTS)
            and tiled.voxel2_shape(H, W, B) is not None and tiled.can_tile((xd, yd, td, pd), "tiled")):
        # LARGE windows: each one through the one-pass path (partition + LDS tiles, two launches per window, ts[0] / ts[-1] of
        # the window read on the device) instead of two global atomics per event: 10 windows of 1 M events 0.3 ms against 1 ms
        grids = []
        oob = D.OobCounter(dev)
        for a, b in zip(bounds[:-1].tolist(), bounds[1:].tolist()):
            cols = [c[a:b] for c in (xd, yd, td, pd)]
            # (a window may start at any event: read where it lies, tiled.column_ok; the last window of a stream that ends with
            # its storage has no readable slack behind it and is copied)
            cols = [c if tiled.column_ok(c) else c.clone() for c in cols]
            out = torch.empty((B, H, W), dtype=torch.float32, device=dev)
            tiled.voxel_f32(*cols, None, None, B, H, W, out, oob, fresh=True)
            grids.append(out.to(device))
        oob.raise_if_set(IndexError, "index out of range for voxel grid of size %s" % ((B, H, W),))
        return grids
    # one launch per chunk of windows: blockIdx.y holds at most 65535 of them, and a chunk's grids are bounded in memory
    # (they move to xs.device before the next chunk is built, as the reference's per-window list would)
    per_chunk = int(max(1, min(65535, _WINDOW_CHUNK_BYTES // (B * H * W * 4))))
    grids = []
    for c0 in range(0, nseg, per_chunk):
        c1 = min(c0 + per_chunk, nseg)
        seg = torch.as_tensor(bounds[c0:c1 + 1], device=dev)
        out = torch.zeros((c1 - c0, B, H, W), dtype=torch.float32, device=dev)
        oob = D.OobCounter(dev)
        max_len = int(np.max(np.diff(bounds[c0:c1 + 1])))
        _lib.call("evk_voxel_segments_f32", D.ptr(xd), D.ptr(yd), D.ptr(td), D.ptr(pd), D.ptr(seg), c1 - c0, max_len, B, H, W,
                  D.ptr(out), oob.ptr, D.stream())
        oob.raise_if_set(IndexError, "index out of range for voxel grid of size %s" % ((B, H, W),))
        out = out.to(device)
        grids.extend(out[k] for k in range(c1 - c0))
    return grids


_WINDOW_CHUNK_BYTES = 8 << 30
# windows at least this long go through the one-pass path one by one (~25 us of launches and host work per window against two
# global atomics per event in the all-windows launch: tools/windows_time.py)
_WINDOW_MIN_EVENTS = 350_000


def voxel_grids_fixed_n_torch(xs, ys, ts, ps, B, n, sensor_size=(180, 240), temporal_bilinear=True):
    """One voxel grid per n consecutive events (reference: voxel_grid.py:37-57; note its range(0, len-n, n) drops the
    last full window when len is a multiple of n -- kept).  All windows are built in one kernel launch."""
    if not temporal_bilinear:
        raise NotImplementedError("temporal_bilinear=False is dead code upstream (voxel_grid.py:144-147)")
    starts = list(range(0, len(xs) - n, n))
    if not starts:
        return []
    return _voxel_windows(xs, ys, ts, ps, B, starts + [starts[-1] + n], sensor_size)


def _device_searchable(ts):
    return isinstance(ts, torch.Tensor) and ts.is_cuda and ts.dim() == 1 and ts.dtype in (torch.float32, torch.float64)


def _searchsorted_device(ts, keys):
    """np.searchsorted(ts.cpu().numpy(), keys) for a device time column without copying it to the host (voxel_grid.py:104-105:
    upstream copies the whole column, 40 MB per 10 M events, for two indices): evk_searchsorted_left, float64 comparisons as
    numpy's.  -> int64 numpy array."""
    ts = ts if ts.is_contiguous() else ts.contiguous()
    kd = torch.from_numpy(np.ascontiguousarray(keys, dtype=np.float64)).to(ts.device)
    out = torch.empty(kd.shape[0], dtype=torch.int64, device=ts.device)
    _lib.call("evk_searchsorted_left", D.ptr(ts), ts.element_size(), ts.shape[0], D.ptr(kd), kd.shape[0], D.ptr(out), D.stream())
    return out.cpu().numpy()


def events_to_voxel_timesync_torch(xs, ys, ts, ps, B, t0, t1, device=None, np_ts=None, sensor_size=(180, 240),
                                   temporal_bilinear=True):
    """Voxel grid of the events with t0 <= t < t1 (reference: voxel_grid.py:82-112)."""
    assert (t1 > t0)
    if device is None:
        device = xs.device
    if np_ts is None and _device_searchable(ts):
        start_idx, end_idx = (int(v) for v in _searchsorted_device(ts, [t0, t1]))
    else:
        if np_ts is None:
            np_ts = ts.cpu().numpy()
        start_idx = np.searchsorted(np_ts, t0)
        end_idx = np.searchsorted(np_ts, t1)
    assert (start_idx < end_idx)
    return events_to_voxel_torch(xs[start_idx:end_idx], ys[start_idx:end_idx], ts[start_idx:end_idx],
                                 ps[start_idx:end_idx], B, device, sensor_size=sensor_size,
                                 temporal_bilinear=temporal_bilinear)


def voxel_grids_fixed_t_torch(xs, ys, ts, ps, B, t, sensor_size=(180, 240), temporal_bilinear=True):
    """One voxel grid per time window of width t (reference: voxel_grid.py:59-80 via events_to_voxel_timesync_torch
    :82-112: window k = events with t_start_k <= ts < t_start_k + t, searchsorted on the host).  The windows are
    consecutive, so all of them are built in one kernel launch."""
    if not temporal_bilinear:
        raise NotImplementedError("temporal_bilinear=False is dead code upstream (voxel_grid.py:144-147)")
    t_first, t_last = D.ends(ts)
    t_starts = np.arange(t_first, t_last - t, t)
    if len(t_starts) == 0:
        return []
    if _device_searchable(ts):      # the bounds of all windows from the device column: 16 bytes per window come back, not the column
        idx = _searchsorted_device(ts, np.concatenate([t_starts, t_starts + t]))
        lo, hi = idx[:len(t_starts)], idx[len(t_starts):]
    else:
        np_ts = ts.cpu().numpy()
        lo = np.searchsorted(np_ts, t_starts)
        hi = np.searchsorted(np_ts, t_starts + t)
    assert np.all(lo < hi)                                  # voxel_grid.py:108
    if np.array_equal(hi[:-1], lo[1:]):                     # contiguous windows: one launch
        return _voxel_windows(xs, ys, ts, ps, B, list(lo) + [hi[-1]], sensor_size)
    return [events_to_voxel_torch(xs[a:b], ys[a:b], ts[a:b], ps[a:b], B, sensor_size=sensor_size,
                                  temporal_bilinear=temporal_bilinear) for a, b in zip(lo, hi)]
