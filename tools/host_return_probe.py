"""Where the host-side outliers of tools/api_sweep.py spend their time: device-to-host return routes for large results
(pageable .cpu() against a pinned staging tensor), and cProfile of the slow entries.    python tools/host_return_probe.py"""
import cProfile
import os
import pstats
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import event_utils_amd as E  # noqa: E402
from event_utils_amd.contrast_max import events_cmax as C, objectives as O  # noqa: E402
from event_utils_amd.representations import image as I  # noqa: E402

warnings.simplefilter("ignore")


def med(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        del r
    return float(np.median(ts)) * 1e3


dev = torch.device("cuda:0")
print("# device-to-host routes, float64 tensors")
for mb in (1, 8, 16, 48):
    t = torch.randn(mb * 131072, dtype=torch.float64, device=dev)

    def r_cpu():
        return t.cpu().numpy()

    def r_pinned_view():
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t, non_blocking=False)
        return h.numpy()

    def r_pinned_copy():
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t, non_blocking=False)
        return h.numpy().copy()

    keep = []

    def r_pinned_view_kept():   # the caller keeps every result: no reuse of the cached pinned block
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t, non_blocking=False)
        keep.append(h)
        return h.numpy()
    print("%3d MB  .cpu().numpy() %7.3f ms | pinned view %7.3f | pinned + np copy %7.3f | pinned view, results kept %7.3f"
          % (mb, med(r_cpu), med(r_pinned_view), med(r_pinned_copy), med(r_pinned_view_kept)), flush=True)
    del keep

n, H, W = 1_000_000, 180, 240
x, y, t, p = bench.structured_scene(3, n, H, W)
xi, yi = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)
x64, y64, p64 = xi.astype(np.float64), yi.astype(np.float64), p.astype(np.float64)
t64 = 1_600_000_000.0 + np.round(t.astype(np.float64) * 1e6) / 1e6
w = E.linvel_warp()
q = np.array([38.0, -24.0])


def obj(cls=O.variance_objective, **kw):
    o = cls(**kw)
    o.sensor_size = (H, W)
    return o


def prof(name, fn, top=14):
    fn()
    pr = cProfile.Profile()
    pr.enable()
    fn()
    torch.cuda.synchronize()
    pr.disable()
    print("\n# ---- %s: %.3f ms" % (name, med(fn, 3)), flush=True)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(top)


trel = t64 - t64[0]
prof("linvel_warp.warp(numpy, grad) 1M", lambda: w.warp(x64, y64, t64, p64, t64[-1], q, compute_grad=True))
prof("events_to_timestamp_image 1M", lambda: I.events_to_timestamp_image(x64, y64, trel, p64, sensor_size=(H, W)))
n2 = 50_000
sl = slice(0, n2)
prof("optimize(sos) 50k", lambda: C.optimize(x64[sl], y64[sl], t64[sl], p64[sl], w, obj(O.sos_objective), numeric_grads=True, img_size=(H, W)), 22)
prof("optimize_r2 50k", lambda: C.optimize_r2(x64[sl], y64[sl], t64[sl], p64[sl], w, obj(), numeric_grads=False), 22)
prof("optimize(analytic) 50k", lambda: C.optimize(x64[sl], y64[sl], t64[sl], p64[sl], w, obj(), numeric_grads=False, img_size=(H, W)), 22)
