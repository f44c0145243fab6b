"""direct vs tiled voxel / IWE time as a function of the event count (sets the 'auto' threshold)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from event_utils_amd import tiled
from event_utils_amd.representations.voxel_grid import _voxel_f32_device
H, W, B = 480, 640, 5
rng = np.random.default_rng(1)
for n in (20_000, 50_000, 100_000, 200_000, 400_000, 1_000_000, 3_000_000):
    x = torch.from_numpy(rng.integers(0, W, n).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.integers(0, H, n).astype(np.float32)).cuda()
    t = torch.from_numpy(np.sort(rng.uniform(0, 0.1, n)).astype(np.float32)).cuda()
    p = torch.from_numpy((rng.integers(0, 2, n) * 2 - 1).astype(np.float32)).cuda()
    out = torch.empty((B, H, W), device="cuda")
    r = {}
    for impl in ("direct", "tiled"):
        r[impl] = tiled._time_ms(lambda: _voxel_f32_device(x, y, t, p, B, (H, W), 0.0, 0.1, out=out, check=False, impl=impl, fresh=True), 20)
    print(n, {k: round(v * 1e3, 1) for k, v in r.items()}, "us")
