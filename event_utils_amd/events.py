"""Device-resident event stream: four SoA columns x, y, t, p in HBM (float32: 16 B/event; float64: 32 B/event), or the
reference's on-disk dtypes as they are (NativeColumns: int16 x, y, float64 / float32 t, uint8 p; 13 / 9 B/event).
Created once per optimisation / window so that every objective evaluation is a pure streaming pass with only the
motion parameters and a few scalars crossing PCIe."""
import numpy as np
import torch

from . import _device as D
from . import _lib


def _abs_max(col):
    """max |col| of a float32 / float64 device column as a python float (evk_abs_max: one streaming kernel, the maximum taken on
    the bit patterns, so a NaN propagates as torch's max() does); one 8-byte read-back."""
    c = col if col.is_contiguous() else col.contiguous()
    out = torch.empty(1, dtype=torch.int64, device=c.device)
    _lib.call("evk_abs_max", D.ptr(c), c.element_size(), c.numel(), D.ptr(out), D.stream())
    bits = int(out.item())
    if c.element_size() == 4:
        return float(np.array([bits & 0xFFFFFFFF], dtype=np.uint32).view(np.float32)[0])
    return float(np.array([bits], dtype=np.int64).view(np.float64)[0])


def time_rebase():
    """Whether DeviceEvents.from_arrays may keep float64 time stamps as float32 differences from ts[-1] (default) or must keep the
    float64 column (EVK_TIME_F64=exact: bit-level agreement with a float64 host computation, direct kernels)."""
    import os
    return os.environ.get("EVK_TIME_F64", "relative") != "exact"


def _narrow_f32(a):
    """(float32 copy of the column, whether it holds exactly the same values)."""
    if a.dtype == np.float32:
        return a, True
    if a.dtype == np.bool_:
        return a.astype(np.float32), True
    b = a.astype(np.float32)
    if np.issubdtype(a.dtype, np.integer):
        return b, bool(a.size == 0 or np.abs(a).max() < 2 ** 24)
    return b, bool(np.array_equal(b, a))


def _f32_lossless(a):
    """Whether every value of the host array survives a conversion to float32 (the 'auto' precision policy for columns that are
    checked on the host; float64 columns are checked by evk_narrow_f64_f32 with the same rule)."""
    return _narrow_f32(np.asarray(a))[1]


class NativeColumns:
    """Device-resident events in the dtypes the reference's files store them in (lib/data_formats/event_packagers.py:
    90-93: xs, ys int16, ts float64, ps bool; h5_to_memmap.py:119-121: xy int16 (N, 2), t float64, p uint8) -- 13 B /
    event, or 9 B with float32 t.  The kernels widen them in registers (include/evk.h, "native on-disk dtypes"):
    x, y -> float, t -> (float)(t - t_offset) with the subtraction in float64, p per p_kind."""
    P_KINDS = {"pm1": 0, "u8": 1, "i8": 2}        # EVK_P_U8_PM1, EVK_P_U8, EVK_P_I8

    def __init__(self, x, y, t, p, xy_stride, t_offset, p_kind):
        self.x, self.y, self.t, self.p = x, y, t, p          # y is None for an interleaved (N, 2) xy array
        self.xy_stride, self.t_offset, self.p_kind = int(xy_stride), float(t_offset), int(p_kind)
        self.n = int(t.shape[0])
        self.t_kind = 1 if t.dtype == torch.float64 else 0   # EVK_T_F64 / EVK_T_F32

    def head(self):
        """The argument prefix shared by evk_bucket_events_native_f32 and evk_native_to_columns_f32."""
        return (D.ptr(self.x), D.ptr(self.y) if self.y is not None else None, self.xy_stride, D.ptr(self.t), self.t_kind,
                self.t_offset, D.ptr(self.p), self.p_kind, self.n)

    def aligned(self):
        return all(a is None or a.data_ptr() % 16 == 0 for a in (self.x, self.y, self.t, self.p))

    def widen(self):
        """-> four float32 SoA columns (evk_native_to_columns_f32)."""
        from . import _lib
        dev = self.t.device
        out = [torch.empty(self.n, dtype=torch.float32, device=dev) for _ in range(4)]
        _lib.call("evk_native_to_columns_f32", *self.head(), *(D.ptr(o) for o in out), D.stream())
        return out


import weakref

_LIVE = weakref.WeakSet()      # every DeviceEvents alive: release_scratch() drops the call caches they hold on the scratch


class DeviceEvents:
    def __init__(self, x, y, t, p, t_host=None, native=None):
        _LIVE.add(self)
        if native is None:
            assert x.shape == y.shape == t.shape == p.shape and x.dim() == 1
            assert x.dtype == y.dtype == t.dtype == p.dtype and x.dtype in (torch.float32, torch.float64)
        self._cols = None if native is not None else (x, y, t, p)
        self.native = native           # NativeColumns: the float32 columns are then widened on first use only
        self._t_host = t_host          # optional host copy of t (float64) for cheap ts[k] / searchsorted
        self.p_scale = 1.0             # adaptive lifespan multiplies ps by 100 (objectives.py:225), folded here
        self._buckets = {}             # cache of tile-bucketed layouts (see tiled.py)
        self._p_absmax = None
        # hint of the optimisers (events_cmax._resident): this set will be evaluated many times, so the objective buckets it by
        # output tile at any event count (tiled.TILED_MIN_EVENTS_IWE_REUSED) instead of only from 150 k events.  Without the hint
        # the first evaluation of a small set takes the direct kernels and the second one buckets it
        self.many_evaluations = False
        self._iwe_plans = 0            # evaluations planned on this set so far (the second one buckets it too)
        self._t_ends = None            # (ts[0], ts[-1]) when known without touching the column
        self.t_offset = 0.0            # absolute time of the column's zero (from_arrays: float64 stamps kept relative to ts[-1])

    # marshalled library calls cached on the object (tiled.cmax_variance: ctypes pointers into per-stream scratch) are not part
    # of its state: a copy / pickle of resident events carries the columns and the buckets only
    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if not k.startswith("_cmax") and k != "_lifespan_cut"}

    def __setstate__(self, state):
        self.__dict__.update(state)
        _LIVE.add(self)

    def _columns(self):
        if self._cols is None:
            self._cols = tuple(self.native.widen())
        return self._cols

    x = property(lambda self: self._columns()[0])
    y = property(lambda self: self._columns()[1])
    t = property(lambda self: self._columns()[2])
    p = property(lambda self: self._columns()[3])

    @property
    def device(self):
        return self.native.t.device if self._cols is None else self._cols[0].device

    # -- construction --------------------------------------------------------------------------------------
    @classmethod
    def from_arrays(cls, xs, ys, ts, ps, precision="auto", device=None, relative_time=False):
        """numpy arrays / torch tensors -> device columns.  precision: 'f32', 'f64' or 'auto' (float32 when every
        column is exactly representable in float32 -- lossless -- else float64).
        relative_time (with 'auto', host arrays): float64 time stamps that are not float32 values -- absolute seconds with
        microsecond resolution, what the reference's h5 / rosbag readers deliver -- no longer force float64 columns: they
        are kept as float32 DIFFERENCES from ts[-1] (see below).  What the optimisers ask for (events_cmax._resident); single
        evaluations keep the exact float64 route."""
        device = device or D.require_gpu()
        if isinstance(xs, torch.Tensor):
            if precision == "auto":
                precision = "f64" if any(a.dtype == torch.float64 for a in (xs, ys, ts, ps)) else "f32"
            dt = torch.float32 if precision == "f32" else torch.float64
            cols = [D.to_device(a, dt, device) for a in (xs, ys, ts, ps)]
            # device SLICES (xs[a:b]) start off a 16-byte boundary three times out of four, and the bucketing kernels read 16 bytes
            # at a time: an event set is evaluated again and again, so misaligned columns are copied once (tiled.realign)
            if all(c.dim() == 1 for c in cols) and cols[0].shape[0] >= 1024:
                cols = [c if c.data_ptr() % 16 == 0 else c.clone() for c in cols]
            return cls(*cols)
        cols = [np.asarray(a).reshape(-1) for a in (xs, ys, ts, ps)]
        n = len(cols[2])
        t_offset, dev_cols = 0.0, None
        if precision == "auto":
            # float64 columns (the reference's host arrays) go up as they are and are narrowed ON THE DEVICE, which also says
            # whether they survived exactly (evk_narrow_f64_f32): two numpy passes per column on the host took ten times as
            # long as the whole optimisation that follows (2 M events: 15 ms against 1.3 ms).  Other dtypes: one float32 copy
            # (it is what gets uploaded) and one comparison.
            raw = [D.to_device(c, torch.float64, device) if c.dtype == np.float64 and n else None for c in cols]
            flags = torch.zeros(4, dtype=torch.int32, device=device)
            dev_cols, exact = [None] * 4, [True] * 4
            for i, c in enumerate(cols):
                if raw[i] is not None:
                    dev_cols[i] = torch.empty(n, dtype=torch.float32, device=device)
                    _lib.call("evk_narrow_f64_f32", D.ptr(raw[i]), n, 0.0, D.ptr(dev_cols[i]), D.ptr(flags[i:]), D.stream())
                else:
                    host32, exact[i] = _narrow_f32(c)
                    dev_cols[i] = D.to_device(host32, torch.float32, device)
            if any(r is not None for r in raw):
                for i, f in enumerate(flags.tolist()):
                    exact[i] = exact[i] and not f
            if all(exact):
                precision = "f32"
            elif relative_time and n and time_rebase() and cols[2].dtype.kind in "fiu" and np.isfinite(cols[2][-1]):
                # Only the time stamps need float64 -- absolute seconds with microsecond resolution, what the reference's h5 /
                # rosbag readers deliver.  The kernels only ever use time DIFFERENCES (t - t_ref, (t - ts[0]) / (ts[-1] - ts[0])),
                # so the column is kept RELATIVE to ts[-1] (the default reference time, objectives.py:186), subtracted in float64
                # and then rounded to float32: the error of a difference is <= 2^-24 of the event's distance from ts[-1] --
                # 6e-9 s over a 0.1 s window -- and the events stay on the float32 (bucketed) path instead of the float64 direct
                # kernels (3-15 x on optimize_contrast, tools/f64_time_probe.py).  The float32 kernels also WARP in float32
                # (x - dt * v rounds to ulp(x) ~ 6e-5 px at x ~ 1000, where the reference's float64 numpy does not): objective
                # values and gradients -- sums over the image -- move by ~1e-6 relative, the argmax by < 1e-3 px/s, but single
                # IWE pixels of a sparse image by up to a few 1e-5 of the maximum, and the derivative images are discontinuous
                # where a warped event crosses a pixel boundary.  Hence opt-in: the optimisers use it (only the argmax leaves
                # them), get_iwe / evaluate_* on host arrays keep the float64 columns.  EVK_TIME_F64=exact: never.
                # Coordinates or weights that are not float32 values (undistorted sub-pixel coordinates) go the same way: their
                # rounding -- <= ulp(x) / 2 -- is of the size of the float32 warp's own.
                if not exact[2]:
                    t_offset = float(cols[2][-1])
                    if raw[2] is not None:
                        _lib.call("evk_narrow_f64_f32", D.ptr(raw[2]), n, t_offset, D.ptr(dev_cols[2]), None, D.stream())
                    else:       # (integer time stamps -- microseconds since the epoch: the subtraction is exact)
                        dev_cols[2] = D.to_device((cols[2] - cols[2][-1]).astype(np.float32), torch.float32, device)
                precision = "f32"
            else:
                precision = "f64"
                dev_cols = [raw[i] if raw[i] is not None else D.to_device(c, torch.float64, device) for i, c in enumerate(cols)]
        dt = torch.float32 if precision == "f32" else torch.float64
        if dev_cols is None:
            dev_cols = [D.to_device(c, dt, device) for c in cols]
        ev = cls(*dev_cols)
        ev.t_offset = t_offset
        if n:   # ts[0] / ts[-1] as the kernels see them, without touching the column (the whole host copy only on demand: t_host())
            ends = [np.float64(cols[2][k]) - t_offset for k in (0, -1)]
            ev._t_ends = tuple(float(np.float32(e)) if dt == torch.float32 else float(e) for e in ends)
        return ev

    @classmethod
    def from_native(cls, xs, ys, ts, ps, polarity="pm1", t_offset=None, device=None):
        """Events in their on-disk dtypes -> device, WITHOUT host-side casts (13 B/event over PCIe and in HBM instead
        of 16; replaces the widening of lib/data_loaders/memmap_dataset.py:19-24 / hdf5_dataset.py:18-23).
          xs, ys   int16 columns, or xs = an (N, 2) int16 xy array and ys = None            (numpy or torch)
          ts       float64 or float32; the kernels use (float)(ts - t_offset), t_offset defaults to ts[0]
          ps       bool / uint8 {0, 1} with polarity='pm1' (-> 2p - 1, what the loaders' get_events returns),
                   polarity='literal' to use the stored values as they are (uint8 / bool / int8)."""
        device = device or D.require_gpu()

        def up(a, dtypes, what):
            if isinstance(a, torch.Tensor):
                a = a.view(torch.uint8) if a.dtype == torch.bool else a
                if a.dtype not in dtypes:
                    raise TypeError("%s must be one of %s, got %s" % (what, dtypes, a.dtype))
                return a.contiguous().to(device)
            a = np.asarray(a)
            a = a.view(np.uint8) if a.dtype == np.bool_ else a
            t = torch.from_numpy(np.ascontiguousarray(a))
            if t.dtype not in dtypes:
                raise TypeError("%s must be one of %s, got %s" % (what, dtypes, t.dtype))
            return t.to(device)
        if ys is None:
            xy = up(xs, (torch.int16,), "xy")
            if xy.dim() != 2 or xy.shape[1] != 2:
                raise ValueError("an interleaved coordinate array must have shape (N, 2)")
            x, y, stride, n = xy, None, 2, xy.shape[0]
        else:
            x, y = up(xs, (torch.int16,), "xs").reshape(-1), up(ys, (torch.int16,), "ys").reshape(-1)
            stride, n = 1, x.shape[0]
        t = up(ts, (torch.float64, torch.float32), "ts").reshape(-1)
        p = up(ps, (torch.uint8, torch.int8), "ps").reshape(-1)
        if not (t.shape[0] == n and p.shape[0] == n and (y is None or y.shape[0] == n)):
            raise ValueError("event columns differ in length")
        if polarity not in ("pm1", "literal"):
            raise ValueError("polarity must be 'pm1' or 'literal'")
        if polarity == "pm1" and p.dtype != torch.uint8:
            raise TypeError("polarity='pm1' maps uint8 / bool {0, 1} to -1 / +1; int8 columns are used literally")
        p_kind = NativeColumns.P_KINDS["pm1" if polarity == "pm1" else ("i8" if p.dtype == torch.int8 else "u8")]
        # ts[0], ts[-1]: read from the caller's host array when there is one, else ONE transfer from the device
        ends = None
        if n:
            ends = (float(np.asarray(ts).reshape(-1)[0]), float(np.asarray(ts).reshape(-1)[-1])) \
                if not isinstance(ts, torch.Tensor) else D.ends(t)
        if t_offset is None:
            t_offset = ends[0] if n else 0.0
        ev = cls(None, None, None, None, native=NativeColumns(x, y, t, p, stride, t_offset, p_kind))
        ev.t_offset = float(t_offset)
        if n:   # as the kernels see them: (float)(t - t_offset), the subtraction in float64
            ev._t_ends = tuple(float(np.float32(np.float64(e) - t_offset)) for e in ends)
        return ev

    # -- array-ish protocol used by the objective code -------------------------------------------------------
    def __len__(self):
        return self.native.n if self._cols is None else self._cols[0].shape[0]

    @property
    def dtype(self):
        return torch.float32 if self._cols is None else self._cols[0].dtype

    def t_at(self, k):
        """ts[k] as a python float (float64 value of the stored column)."""
        if self._t_host is not None:
            return float(self._t_host[k])
        if self._t_ends is not None and k in (0, -1):
            return self._t_ends[k]
        return float(self.t[k].item())

    def t_host(self):
        if self._t_host is None:
            self._t_host = self.t.double().cpu().numpy()
        return self._t_host

    def p_absmax(self):
        """max |p| (one device reduction, cached): bounds the accumulator sums for fixed-point LDS accumulation."""
        if self._p_absmax is None:
            if self._cols is None and self.native.p_kind == 0:
                self._p_absmax = 1.0 if len(self) else 0.0          # {0, 1} -> -1 / +1
            else:
                self._p_absmax = _abs_max(self.p) if len(self) else 0.0
        return self._p_absmax

    def fresh_view(self):
        """The same resident columns as a NEW event set -- no buckets, no cached calls, what it knows about ts[0] / ts[-1] kept:
        the state right after an upload (measurements of the first-use path: bench.py, tools/)."""
        ev = DeviceEvents(*(self._cols or (None,) * 4), t_host=self._t_host, native=self.native)
        ev._cols, ev._t_ends, ev.t_offset, ev.p_scale = self._cols, self._t_ends, self.t_offset, self.p_scale
        return ev

    def slice(self, start, stop):
        """View of events [start:stop) (python slice semantics, no copy)."""
        sl = slice(start, stop)
        ev = DeviceEvents(self.x[sl], self.y[sl], self.t[sl], self.p[sl],
                          t_host=None if self._t_host is None else self._t_host[sl])
        ev.p_scale = self.p_scale
        ev.t_offset = self.t_offset
        return ev

    def scaled(self, factor):
        ev = DeviceEvents(*(self._cols or (None,) * 4), t_host=self._t_host, native=self.native)
        ev._cols, ev._t_ends = self._cols, self._t_ends
        ev.p_scale = self.p_scale * factor
        ev._buckets = self._buckets
        ev._p_absmax = self._p_absmax
        ev.t_offset = self.t_offset
        return ev
