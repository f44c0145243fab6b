"""Direct (global atomics) against one-pass (partition + LDS tiles) event-image kernels over the event count: where does
EVK_IMPL=auto switch?  240x180 and 640x480, integer nearest and bilinear."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_utils_amd import tiled, _lib
from event_utils_amd import _device as D
INF = float("inf")
for (H, W) in ((180, 240), (480, 640)):
    for n in (50_000, 100_000, 200_000, 400_000, 800_000, 1_600_000):
        rng = np.random.default_rng(1)
        x = rng.uniform(0, W - 1, n).astype(np.float32); y = rng.uniform(0, H - 1, n).astype(np.float32)
        p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
        xd, yd, pd = (torch.from_numpy(a).cuda() for a in (x, y, p))
        xi, yi, pi = xd.int(), yd.int(), pd.int()
        row = []
        for kind, cols, dt, fn in (("i32", (xi, yi, pi), torch.int32, "evk_image_nearest_i32"), ("bilinear", (xd, yd, pd), torch.float32, "evk_image_bilinear_f32")):
            img = torch.zeros((H, W), dtype=dt, device="cuda")
            t_tiled = tiled._time_ms(lambda: tiled.image2(kind, *cols, n, H, W, INF, INF, img, None), 100)
            if kind == "i32":
                direct = lambda: _lib.call(fn, D.ptr(cols[0]), D.ptr(cols[1]), D.ptr(cols[2]), n, H, W, D.ptr(img), None, D.stream())
            else:
                direct = lambda: _lib.call(fn, D.ptr(cols[0]), D.ptr(cols[1]), D.ptr(cols[2]), n, H, W, INF, INF, D.ptr(img), None, D.stream())
            t_dir = tiled._time_ms(direct, 100)
            row.append("%s tiled %.1f us direct %.1f us" % (kind, t_tiled * 1e3, t_dir * 1e3))
        print("%dx%d n=%8d  %s" % (W, H, n, "   ".join(row)), flush=True)
