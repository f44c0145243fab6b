"""Voxel path on a structured (moving-edge) scene against uniform-random events: kernel times of one call."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from event_utils_amd import tiled  # noqa: E402

torch.cuda.set_device(0)
for (H, W, n) in ((480, 640, 10_000_000), (720, 1280, 50_000_000)):
    for scene in ("uniform", "edges", "blob"):
        if scene == "uniform":
            rng = np.random.default_rng(1)
            x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
            t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32); p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
        elif scene == "edges":
            x, y, t, p = bench.structured_scene(3, n, H, W)
            x, y = np.floor(x), np.floor(y)
        else:   # half of the events inside a 100 x 100 px blob
            rng = np.random.default_rng(2)
            x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
            hot = rng.random(n) < 0.5
            x[hot] = (W // 3 + rng.integers(0, 100, hot.sum())).astype(np.float32)
            y[hot] = (H // 3 + rng.integers(0, 100, hot.sum())).astype(np.float32)
            t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32); p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
        cols = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (x, y, t, p)]
        k = tiled.time_voxel_kernels([cols], float(t[0]), float(t[-1]), 5, H, W, impl="tiled", reps=10)
        print("%dx%d n=%d %-8s total %.4f ms  %s" % (W, H, n, scene, k["total_ms"], k["kernels_ms"]), flush=True)
        del cols
