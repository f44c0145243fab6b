import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from event_utils_amd import tiled
rng = np.random.default_rng(1)
n, H, W, B = 10_000_000, 480, 640, 5
x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32)
p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
for name, pp in (("unit", p), ("narrow", p * np.array([0.25, 0.5, 1.0, 1.5, 2.0, 3.0], np.float32)[np.arange(n) % 6]), ("wide", (p * rng.uniform(0.5, 1.5, n)).astype(np.float32))):
    cols = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (x, y, t, pp)]
    k = tiled.time_voxel_kernels([cols], float(t[0]), float(t[-1]), B, H, W, impl="tiled", reps=20)
    print(name, k["total_ms"], k["kernels_ms"])
