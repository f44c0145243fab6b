"""Where the public events_to_voxel_torch call spends more than the internal entry point (device time per call, HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import event_utils_amd as E
from event_utils_amd import tiled
from event_utils_amd.representations.voxel_grid import _voxel_f32_device
torch.cuda.set_device(0)
n, H, W, B = 10_000_000, 480, 640, 5
rng = np.random.default_rng(1)
x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32); p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
c = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
out = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
t0, t1 = float(t[0]), float(t[-1])
variants = {
    "internal (host ts ends, resident grid, no check)": lambda: _voxel_f32_device(*c, B, (H, W), t0, t1, out=out, check=False, impl="tiled", fresh=True),
    "internal + device ts ends": lambda: _voxel_f32_device(*c, B, (H, W), None, None, out=out, check=False, impl="tiled", fresh=True),
    "internal + check": lambda: _voxel_f32_device(*c, B, (H, W), t0, t1, out=out, check=True, impl="tiled", fresh=True),
    "internal + check + device ts ends": lambda: _voxel_f32_device(*c, B, (H, W), None, None, out=out, check=True, impl="tiled", fresh=True),
    "internal + new grid": lambda: _voxel_f32_device(*c, B, (H, W), t0, t1, check=False, impl="tiled", fresh=True),
    "public": lambda: E.events_to_voxel_torch(*c, B, sensor_size=(H, W)),
}
for rep in range(2):
    for name, fn in variants.items():
        print("%-52s %.4f ms" % (name, tiled._time_ms(fn, 40)), flush=True)
