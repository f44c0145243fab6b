"""Stage timings of the one-pass event-image path (evk_image2.hip) against the direct global-atomic kernels:
10 M events 640x480 (and 1 M events 240x180, configs[0]), integer / float32 nearest and bilinear, unit and float weights.
usage: python tools/image2_time.py [--scenes]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_utils_amd import _lib, tiled  # noqa: E402
from event_utils_amd import _device as D  # noqa: E402

INF = float("inf")


def run(n, H, W, scene="uniform", reps=20):
    rng = np.random.default_rng(1)
    x = rng.uniform(0, W - 1, n).astype(np.float32); y = rng.uniform(0, H - 1, n).astype(np.float32)
    if scene == "blob":
        hot = rng.random(n) < 0.5
        x[hot] = rng.uniform(300, 340, hot.sum()).astype(np.float32); y[hot] = rng.uniform(200, 230, hot.sum()).astype(np.float32)
    elif scene == "edges":
        t = np.linspace(0, 1, n); hot = rng.random(n) < 0.8
        x[hot] = np.clip(100 + 400 * t[hot] + rng.normal(0, 0.7, hot.sum()), 0, W - 1.001).astype(np.float32)
    pu = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    pf = rng.normal(size=n).astype(np.float32)
    xd, yd, pud, pfd = (torch.from_numpy(a).cuda() for a in (x, y, pu, pf))
    xi, yi, pi = xd.int(), yd.int(), pud.int()
    out = {}
    rot = 4 if "--rotate" in sys.argv else 1      # 4 copies of the columns: every call reads its events from HBM
    for name, kind, cols, dt in (("int32 nearest", "i32", (xi, yi, pi), torch.int32),
                                 ("f32 nearest unit", "f32", (xd, yd, pud), torch.float32),
                                 ("f32 nearest float", "f32", (xd, yd, pfd), torch.float32),
                                 ("bilinear unit", "bilinear", (xd, yd, pud), torch.float32),
                                 ("bilinear float", "bilinear", (xd, yd, pfd), torch.float32)):
        img = torch.zeros((H, W), dtype=dt, device="cuda")
        copies = [cols] + [tuple(c.clone() for c in cols) for _ in range(rot - 1)]
        it = [0]

        def call(stage=0, copies=copies, it=it, kind=kind, img=img):
            it[0] += 1
            return tiled.image2(kind, *copies[it[0] % len(copies)], n, H, W, INF, INF, img, None, fresh=False, stage=stage)
        assert call()
        total = tiled._time_ms(call, reps)
        part = tiled._time_ms(lambda: call(_lib.EVK_VOXEL2_PARTITION_ONLY), reps)
        tiles = tiled._time_ms(lambda: call(_lib.EVK_VOXEL2_TILES_ONLY), reps)
        if kind == "i32":
            direct = lambda: _lib.call("evk_image_nearest_i32", D.ptr(cols[0]), D.ptr(cols[1]), D.ptr(cols[2]), n, H, W, D.ptr(img), None, D.stream())
        else:
            fn = "evk_image_bilinear_f32" if kind == "bilinear" else "evk_image_nearest_f32"
            direct = lambda: _lib.call(fn, D.ptr(cols[0]), D.ptr(cols[1]), D.ptr(cols[2]), n, H, W, INF, INF, D.ptr(img), None, D.stream())
        dms = tiled._time_ms(direct, 3)
        out[name] = (total, part, tiles, dms)
        print("%-8s n=%9d %dx%d %-18s total %.4f ms = %6.1f Gev/s (%.3f of 8 TB/s at 12 B/ev) | partition %.4f tiles %.4f | direct %.3f ms"
              % (scene, n, W, H, name, total, n / total / 1e6, 12 * n / total / 1e6 / 8000, part, tiles, dms), flush=True)
    return out


if __name__ == "__main__":
    print("tiles", tiled.voxel2_shape(480, 640, 1), tiled.voxel2_shape(180, 240, 1), tiled.voxel2_shape(481, 641, 1))
    run(10_000_000, 480, 640)
    run(1_000_000, 180, 240)
    if "--scenes" in sys.argv:
        run(10_000_000, 480, 640, "blob")
        run(10_000_000, 480, 640, "edges")
    if "--big" in sys.argv:
        run(50_000_000, 720, 1280)
