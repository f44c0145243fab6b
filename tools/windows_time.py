"""voxel_grids_fixed_n_torch: windows of 1 M events (the one-pass path per window) against the all-windows-in-one-launch
global-atomic kernel (EVK_IMPL=direct), 10 M events 640x480x5.   usage: python tools/windows_time.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import event_utils_amd as E  # noqa: E402

n, H, W, B = 10_000_000, 480, 640, 5
rng = np.random.default_rng(0)
cols = [torch.from_numpy(a).cuda() for a in (rng.integers(0, W, n).astype(np.float32), rng.integers(0, H, n).astype(np.float32),
                                              np.sort(rng.uniform(0, 1, n)).astype(np.float32),
                                              (rng.integers(0, 2, n) * 2 - 1).astype(np.float32))]
for win in (1_000_000, 500_000, 400_001):
    for impl in ("auto", "direct"):
        os.environ["EVK_IMPL"] = impl
        for _ in range(2):
            g = E.voxel_grids_fixed_n_torch(*cols, B, win, sensor_size=(H, W))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            g = E.voxel_grids_fixed_n_torch(*cols, B, win, sensor_size=(H, W))
        torch.cuda.synchronize()
        print("windows of %7d events (%2d grids)  EVK_IMPL=%-6s %.3f ms per call" % (win, len(g), impl, (time.perf_counter() - t0) * 100))
