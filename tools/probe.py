"""GPU probe: streaming ceiling and atomic rates of the direct kernels (evidence for DESIGN.md's kernel choices)."""
import json
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from event_utils_amd import tiled, _lib, _device as D
from event_utils_amd.events import DeviceEvents

dev = torch.device("cuda", 0)
res = {}
for (H, W, n) in ((480, 640, 10_000_000), (720, 1280, 50_000_000)):
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.integers(0, W, n).astype(np.float32)).to(dev)
    y = torch.from_numpy(rng.integers(0, H, n).astype(np.float32)).to(dev)
    t = torch.from_numpy(np.sort(rng.uniform(0, 0.1, n)).astype(np.float32)).to(dev)
    p = torch.from_numpy((rng.integers(0, 2, n) * 2 - 1).astype(np.float32)).to(dev)
    z = torch.zeros_like(p)
    B = 5
    out = torch.zeros((B, H, W), device=dev)
    key = "%dx%d_%dM" % (W, H, n // 1000000)
    r = {}
    ms = tiled._time_ms(lambda: tiled.voxel_f32(x, y, t, z, 0.0, 0.1, B, H, W, out, impl="direct"), 10)
    r["read_only_ms"] = ms; r["read_only_GBs"] = 16 * n / ms / 1e6
    ms = tiled._time_ms(lambda: tiled.voxel_f32(x, y, t, p, 0.0, 0.1, B, H, W, out, impl="direct"), 10)
    r["voxel_direct_ms"] = ms; r["voxel_direct_Mev_s"] = n / ms / 1e3; r["voxel_atomics_G_s"] = 2 * n / ms / 1e6
    img = torch.zeros((H, W), device=dev)
    inf = float("inf")
    ms = tiled._time_ms(lambda: _lib.call("evk_image_nearest_f32", D.ptr(x), D.ptr(y), D.ptr(p), n, H, W, inf, inf, D.ptr(img), None, D.stream()), 10)
    r["image_nearest_f32_ms"] = ms; r["image_atomics_G_s"] = n / ms / 1e6
    xr = x + 0.37; yr = y + 0.61
    xr.clamp_(1, W - 1.5); yr.clamp_(1, H - 1.5)
    ev = DeviceEvents(xr, yr, t, p)
    buf = torch.zeros((3, H + 1, W + 1), device=dev)
    for flags, name, na in ((0, "iwe", 4), (2, "iwe_grad", 12)):
        ms = tiled._time_ms(lambda: tiled.iwe_linvel(ev, 0.1, 30.0, -20.0, float(W), float(H), H + 1, W + 1, flags, buf[0], buf[1:3], impl="direct"), 5)
        r[name + "_direct_ms"] = ms; r[name + "_Mev_s"] = n / ms / 1e3; r[name + "_atomics_G_s"] = na * n / ms / 1e6
    # sorted-by-pixel events: contention-free but same-address streaks (upper bound of what bucketing buys for atomics)
    res[key] = r
    del x, y, t, p, z, xr, yr, ev
print(json.dumps(res, indent=1))
