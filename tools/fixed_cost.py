"""Fixed costs of the voxel call's two kernels: stage timings at small event counts, next to an (almost) empty launch."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_utils_amd import tiled, _lib
from event_utils_amd import _device as D
t1 = torch.zeros(64, device="cuda"); o1 = torch.zeros(64, device="cuda")
empty = lambda: _lib.call("evk_normalise_time_f32", D.ptr(t1), 64, 0.0, 1.0, 5, D.ptr(o1), D.stream())
print("empty launch %.2f us" % (tiled._time_ms(empty, 200) * 1e3))
H, W, B = 480, 640, 5
for n in (400_000, 1_000_000, 2_500_000, 5_000_000, 10_000_000):
    rng = np.random.default_rng(1)
    x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32); p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    k = tiled.time_voxel_kernels([cols], float(t[0]), float(t[-1]), B, H, W, impl="tiled", reps=50)
    print(n, "total %.1f us" % (k["total_ms"] * 1e3), {a: round(b * 1e3, 1) for a, b in k["kernels_ms_exact"].items()})
