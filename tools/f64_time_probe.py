"""What float64 time stamps cost: optimize_contrast on host arrays whose ts are float64 seconds with an absolute offset (what the
reference's h5 / rosbag readers deliver; not representable in float32) against the same events with float32-exact ts.
    python tools/f64_time_probe.py"""
import os
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import event_utils_amd as E  # noqa: E402
from event_utils_amd.contrast_max.events_cmax import optimize_contrast  # noqa: E402

warnings.simplefilter("ignore")
w = E.linvel_warp()
for n, H, W in ((100_000, 180, 240), (2_000_000, 480, 640)):
    x, y, t, p = bench.structured_scene(3, n, H, W)
    x64, y64, p64 = x.astype(np.float64), y.astype(np.float64), p.astype(np.float64)
    t_abs = 1_600_000_000.0 + np.round(t.astype(np.float64) * 1e6) / 1e6          # epoch seconds, microsecond resolution
    t_rel = (np.round(t.astype(np.float64) * 1e6) / 1e6).astype(np.float32).astype(np.float64)
    for name, ts in (("float32-exact ts", t_rel), ("float64 epoch ts", t_abs)):
        for opt in ("evk_bfgs", "fmin_bfgs"):
            def run():
                o = E.variance_objective()
                o.sensor_size, o.reference_exact = (H, W), False
                kw = dict(optimizer="evk_bfgs") if opt == "evk_bfgs" else {}
                return optimize_contrast(x64, y64, ts, p64, w, o, numeric_grads=False, blur_sigma=1.0, img_size=(H, W), **kw)
            run()
            tt = []
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter(); a = run(); torch.cuda.synchronize(); tt.append(time.perf_counter() - t0)
            print("n=%-8d %dx%d  %-17s %-9s %.2f ms (upload included) -> %s" % (n, W, H, name, opt, float(np.median(tt)) * 1e3, np.round(a, 3)), flush=True)
