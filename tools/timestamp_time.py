"""events_to_timestamp_image_torch on device tensors: the one-pass path (evk_timestamp_images2_f32) against the direct
global-atomic kernel (evk_timestamp_images_f32), kernel time by HIP events.    python tools/timestamp_time.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import event_utils_amd as E  # noqa: E402
from event_utils_amd import tiled  # noqa: E402

for n, H, W, scene in ((50_000, 180, 240, "uniform"), (200_000, 180, 240, "uniform"), (1_000_000, 180, 240, "uniform"),
                       (1_000_000, 180, 240, "edges"), (10_000_000, 480, 640, "uniform"), (10_000_000, 480, 640, "edges"),
                       (50_000_000, 720, 1280, "uniform"))[slice(*(map(int, sys.argv[1:3]) if len(sys.argv) > 2 else (None,)))]:
    if scene == "edges":
        x, y, t, p = bench.structured_scene(3, n, H, W)
    else:
        x, y, t, p = bench.synth(5, n, 0.0, 0.1, real_xy=True)
        x, y = x * (W / 640.0), y * (H / 480.0)
    cols = [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda() for a in (x, y, t, p)]
    row = []
    for impl in ("direct", "tiled"):
        os.environ["EVK_IMPL"] = impl
        fn = lambda: E.events_to_timestamp_image_torch(*cols, sensor_size=(H, W))   # noqa: E731
        reps = 3 if (impl == "direct" and n > 5_000_000) else 20
        row.append(tiled._time_ms(fn, reps))
    alg = 16.0 * n + 4 * 4.0 * (H + 1) * (W + 1)
    print("n=%-9d %4dx%-4d %-8s direct %9.4f ms | one-pass %8.4f ms (%.0f Mev/s, %.3f of 8 TB/s on 16 B/event + 4 planes) | x%.1f"
          % (n, W, H, scene, row[0], row[1], n / row[1] / 1e3, alg / (row[1] * 1e-3) / 8e12, row[0] / row[1]), flush=True)
