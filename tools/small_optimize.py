"""optimize_contrast at the reference's own sizes (a DAVIS 240x180 sensor, windows of 5 k - 300 k events; 640x480 too): scipy's
fmin_bfgs with numeric gradients (the reference's default, events_cmax.py:343), with the analytic gradient, and evk_bfgs --
with the event set bucketed at any count (DeviceEvents.many_evaluations, round 6) and with the 'auto' threshold of a single
evaluation (150 k events: below it the direct kernels).     python tools/small_optimize.py"""
import os
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import event_utils_amd as E  # noqa: E402
from event_utils_amd import tiled  # noqa: E402
from event_utils_amd.contrast_max.events_cmax import optimize_contrast  # noqa: E402

w = E.linvel_warp()
warnings.simplefilter("ignore")
for H, W in ((180, 240), (480, 640)):
    for n in (5_000, 30_000, 100_000, 300_000):
        x, y, t, p = bench.structured_scene(3, n, H, W)
        ev0 = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
        row = []
        for name, kw in (("fmin_bfgs numeric", dict(numeric_grads=True)), ("fmin_bfgs analytic", dict(numeric_grads=False)),
                         ("evk_bfgs", dict(numeric_grads=False, optimizer="evk_bfgs"))):
            for reused in (1, 150_000):
                tiled.TILED_MIN_EVENTS_IWE_REUSED = reused
                ev = ev0.fresh_view()   # (no buckets, no cached calls of the other mode)

                def run():
                    o = E.variance_objective()
                    o.sensor_size, o.reference_exact = (H, W), False
                    return optimize_contrast(ev, None, None, None, w, o, blur_sigma=1.0, img_size=(H, W), **kw)
                run()
                ts = []
                for _ in range(5):
                    torch.cuda.synchronize(); t0 = time.perf_counter(); a = run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
                row.append("%s %s %.2f ms" % (name, "bucketed" if reused == 1 else "auto", float(np.median(ts)) * 1e3))
        tiled.TILED_MIN_EVENTS_IWE_REUSED = 1
        print("%dx%d n=%-7d %s -> %s" % (W, H, n, " | ".join(row), np.round(a, 2)), flush=True)
