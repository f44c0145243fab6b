"""End-to-end time of the public calls on HOST arrays (the reference's calling convention), against what the link alone would
take, at the sizes of configs[0] / configs[1]: where the host side of the drop-in API spends its time.
    python tools/host_entry_times.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import event_utils_amd as E  # noqa: E402


def med(fn, reps=7):
    fn(); fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


rng = np.random.default_rng(0)
for n, H, W in ((1_000_000, 180, 240), (10_000_000, 480, 640)):
    xi, yi = rng.integers(0, W, n), rng.integers(0, H, n)                 # int64, as numpy makes them
    pi = rng.integers(0, 2, n) * 2 - 1
    t = np.sort(rng.uniform(0, 0.1, n)) + 1_600_000_000.0
    xf, yf, pf = xi.astype(np.float64), yi.astype(np.float64), pi.astype(np.float64)
    rows = [
        ("events_to_image (int64 x, y, p)", lambda: E.events_to_image(xi, yi, pi, sensor_size=(H, W)), 3 * 8 * n),
        ("events_to_voxel (int64 x, y; float64 t, p; 5 bins)", lambda: E.events_to_voxel(xi, yi, t, pf, 5, sensor_size=(H, W)), 4 * 8 * n),
        ("get_iwe (float64 columns, value only)", lambda: E.get_iwe(np.array([30.0, -20.0]), xf, yf, t, pf, E.linvel_warp(), (H, W),
                                                                  sensor_size=(H, W)), 4 * 8 * n),
    ]
    x32, y32, p32 = (torch.from_numpy(a.astype(np.float32)) for a in (xf, yf, pf))
    t32 = torch.from_numpy((t - t[0]).astype(np.float32))
    rows += [
        ("events_to_image_torch (CPU float32 tensors)", lambda: E.events_to_image_torch(x32, y32, p32, sensor_size=(H, W)), 3 * 4 * n),
        ("events_to_voxel_torch (CPU float32 tensors, 5 bins)", lambda: E.events_to_voxel_torch(x32, y32, t32, p32, 5, sensor_size=(H, W)), 4 * 4 * n),
    ]
    for name, fn, nbytes in rows:
        ms = med(fn)
        print("n=%-9d %dx%d  %-52s %8.2f ms  (the columns over a 50 GB/s link: %.2f ms)" % (n, W, H, name, ms, nbytes / 50e9 * 1e3), flush=True)
