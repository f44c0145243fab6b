"""Voxel path on spatially clustered event streams (real data is edges and blobs, not uniform noise): per-call time for
10 M events, 640x480x5, tiled vs direct.   python tools/clustered_voxel.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_utils_amd import tiled  # noqa: E402
from event_utils_amd.representations.voxel_grid import _voxel_f32_device  # noqa: E402

H, W, B, n = 480, 640, 5, 10_000_000
rng = np.random.default_rng(0)


def scene(kind):
    if kind == "uniform":
        x, y = rng.integers(0, W, n), rng.integers(0, H, n)
    elif kind == "edges":            # 24 vertical + 18 horizontal lines, 1.5 px wide
        vert = rng.random(n) < 0.5
        x = np.where(vert, rng.choice(np.arange(20, W - 20, 25), n) + rng.normal(0, 0.7, n), rng.uniform(0, W, n))
        y = np.where(vert, rng.uniform(0, H, n), rng.choice(np.arange(20, H - 20, 25), n) + rng.normal(0, 0.7, n))
    elif kind == "blob":             # 90 % of the events inside a 64 x 48 patch
        inb = rng.random(n) < 0.9
        x = np.where(inb, rng.uniform(300, 364, n), rng.uniform(0, W, n))
        y = np.where(inb, rng.uniform(200, 248, n), rng.uniform(0, H, n))
    else:                            # "pixel": half of the events on ONE pixel (a hot pixel)
        hot = rng.random(n) < 0.5
        x = np.where(hot, 123, rng.integers(0, W, n))
        y = np.where(hot, 77, rng.integers(0, H, n))
    x = np.clip(np.floor(x), 0, W - 1).astype(np.float32)
    y = np.clip(np.floor(y), 0, H - 1).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    return [torch.from_numpy(a).cuda() for a in (x, y, t, p)]


out = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
for kind in ("uniform", "edges", "blob", "pixel"):
    cols = scene(kind)
    res = []
    for impl in ("tiled", "direct"):
        fn = lambda: _voxel_f32_device(*cols, B, (H, W), 0.0, 0.1, out=out, check=False, impl=impl, fresh=True)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / 10)
        ref = out.clone() if impl == "tiled" else ref
    err = (out - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
    k = tiled.time_voxel_kernels([cols], 0.0, 0.1, B, H, W, impl="tiled", reps=5)["kernels_ms"]
    print("%-8s tiled %.3f ms (%5.1f Gev/s)   direct %.3f ms   rel.diff %.1e   %s" %
          (kind, res[0], n / res[0] / 1e6, res[1], err, {a: round(b, 3) for a, b in k.items()}))
