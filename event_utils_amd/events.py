"""Device-resident event stream: four SoA columns x, y, t, p in HBM (float32: 16 B/event; float64: 32 B/event).
Created once per optimisation / window so that every objective evaluation is a pure streaming pass with only the
motion parameters and a few scalars crossing PCIe."""
import numpy as np
import torch

from . import _device as D


def _f32_lossless(a):
    a = np.asarray(a)
    if a.dtype == np.float32 or np.issubdtype(a.dtype, np.integer) and (a.size == 0 or np.abs(a).max() < 2 ** 24):
        return True
    if a.dtype == np.bool_:
        return True
    return bool(np.array_equal(a.astype(np.float32).astype(np.float64), a.astype(np.float64)))


class DeviceEvents:
    def __init__(self, x, y, t, p, t_host=None):
        assert x.shape == y.shape == t.shape == p.shape and x.dim() == 1
        assert x.dtype == y.dtype == t.dtype == p.dtype and x.dtype in (torch.float32, torch.float64)
        self.x, self.y, self.t, self.p = x, y, t, p
        self._t_host = t_host          # optional host copy of t (float64) for cheap ts[k] / searchsorted
        self.p_scale = 1.0             # adaptive lifespan multiplies ps by 100 (objectives.py:225), folded here
        self._buckets = {}             # cache of tile-bucketed layouts (see tiled.py)
        self._p_absmax = None

    # -- construction --------------------------------------------------------------------------------------
    @classmethod
    def from_arrays(cls, xs, ys, ts, ps, precision="auto", device=None):
        """numpy arrays / torch tensors -> device columns.  precision: 'f32', 'f64' or 'auto' (float32 when every
        column is exactly representable in float32 -- lossless -- else float64)."""
        device = device or D.require_gpu()
        if isinstance(xs, torch.Tensor):
            if precision == "auto":
                precision = "f64" if any(a.dtype == torch.float64 for a in (xs, ys, ts, ps)) else "f32"
            dt = torch.float32 if precision == "f32" else torch.float64
            return cls(*(D.to_device(a, dt, device) for a in (xs, ys, ts, ps)))
        cols = [np.asarray(a).reshape(-1) for a in (xs, ys, ts, ps)]
        if precision == "auto":
            precision = "f32" if all(_f32_lossless(c) for c in cols) else "f64"
        dt = torch.float32 if precision == "f32" else torch.float64
        t_host = cols[2].astype(np.float64) if dt == torch.float64 else cols[2].astype(np.float32).astype(np.float64)
        return cls(*(D.to_device(c, dt, device) for c in cols), t_host=t_host)

    # -- array-ish protocol used by the objective code -------------------------------------------------------
    def __len__(self):
        return self.x.shape[0]

    @property
    def dtype(self):
        return self.x.dtype

    def t_at(self, k):
        """ts[k] as a python float (float64 value of the stored column)."""
        if self._t_host is not None:
            return float(self._t_host[k])
        return float(self.t[k].item())

    def t_host(self):
        if self._t_host is None:
            self._t_host = self.t.double().cpu().numpy()
        return self._t_host

    def p_absmax(self):
        """max |p| (one device reduction, cached): bounds the accumulator sums for fixed-point LDS accumulation."""
        if self._p_absmax is None:
            self._p_absmax = float(self.p.abs().max().item()) if len(self) else 0.0
        return self._p_absmax

    def slice(self, start, stop):
        """View of events [start:stop) (python slice semantics, no copy)."""
        sl = slice(start, stop)
        ev = DeviceEvents(self.x[sl], self.y[sl], self.t[sl], self.p[sl],
                          t_host=None if self._t_host is None else self._t_host[sl])
        ev.p_scale = self.p_scale
        return ev

    def scaled(self, factor):
        ev = DeviceEvents(self.x, self.y, self.t, self.p, t_host=self._t_host)
        ev.p_scale = self.p_scale * factor
        ev._buckets = self._buckets
        ev._p_absmax = self._p_absmax
        return ev
