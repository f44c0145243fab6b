"""A stream of windows: optimize_contrast on fresh host arrays of varying size, hundreds of times -- device memory in use must
stay bounded (resident events and their buckets go with their DeviceEvents; the per-stream scratch is grow-only up to the
largest window).      python tools/stream_leak_check.py"""
import gc
import os
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import event_utils_amd as E  # noqa: E402
from event_utils_amd.contrast_max.events_cmax import optimize_contrast  # noqa: E402

warnings.simplefilter("ignore")
H, W = 180, 240
rng = np.random.default_rng(0)
w = E.linvel_warp()
x, y, t, p = bench.structured_scene(3, 400_000, H, W)
marks = []
for it in range(400):
    n = int(rng.integers(5_000, 400_000))
    lo = int(rng.integers(0, 400_000 - n + 1))
    sl = slice(lo, lo + n)
    o = E.variance_objective()
    o.sensor_size, o.reference_exact = (H, W), False
    ts = 1.6e9 + t[sl].astype(np.float64)
    a = optimize_contrast(x[sl].astype(np.float64), y[sl].astype(np.float64), ts, p[sl].astype(np.float64), w, o,
                          optimizer="evk_bfgs" if it % 2 else None or __import__("scipy.optimize").optimize.fmin_bfgs,
                          numeric_grads=bool(it % 3 == 0), blur_sigma=1.0, img_size=(H, W))
    if it % 50 == 49:
        gc.collect()
        torch.cuda.synchronize()
        marks.append(torch.cuda.memory_allocated() / 1e6)
        print("after %3d windows: %.1f MB allocated, %.1f MB reserved" % (it + 1, marks[-1], torch.cuda.memory_reserved() / 1e6), flush=True)
assert marks[-1] <= 1.25 * max(marks[1:4]) + 8, marks
E.release_scratch()
gc.collect()
print("after release_scratch: %.1f MB allocated" % (torch.cuda.memory_allocated() / 1e6))
print("LEAK CHECK ok")
