"""Times the bucketing and tile kernels of the voxel path for several tile shapes (GPU)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from event_utils_amd import tiled
H, W, B, n = 480, 640, 5, 10_000_000
if len(sys.argv) > 1 and sys.argv[1] == "720p":
    H, W, n = 720, 1280, 50_000_000
rng = np.random.default_rng(1)
x = torch.from_numpy(rng.integers(0, W, n).astype(np.float32)).cuda()
y = torch.from_numpy(rng.integers(0, H, n).astype(np.float32)).cuda()
t = torch.from_numpy(np.sort(rng.uniform(0, 0.1, n)).astype(np.float32)).cuda()
p = torch.from_numpy((rng.integers(0, 2, n) * 2 - 1).astype(np.float32)).cuda()
for shape in ("3x3", "4x3", "4x4", "5x4", "5x5", "6x5", "6x6"):
    os.environ["EVK_VOXEL_TILE"] = shape
    try:
        r = tiled.time_voxel_kernels([(x, y, t, p)], 0.0, 0.1, B, H, W, impl="tiled", reps=10)
        print(shape, json.dumps(r))
    except Exception as e:
        print(shape, "ERR", e)
