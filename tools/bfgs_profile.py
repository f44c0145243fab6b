"""Where the host's share of optimize_contrast(optimizer='evk_bfgs') goes: cProfile of one warm call on the moving-edge scene
(10 M events 640x480 by default; N H W as arguments), and the wall time of 5 calls."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import event_utils_amd as E  # noqa: E402
from event_utils_amd.contrast_max.events_cmax import optimize_contrast  # noqa: E402

n, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (10_000_000, 480, 640)
if os.environ.get("EVK_BFGS_PYTHON_LOOP") == "1":          # A/B: the Python loop over the bound closures instead of the library's loop
    from event_utils_amd import tiled
    tiled.FORCE["native_bfgs"] = False
x, y, t, p = bench.structured_scene(3, n, H, W)
ev = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
w = E.linvel_warp()


def run():
    o = E.variance_objective()
    o.sensor_size, o.reference_exact = (H, W), False
    return optimize_contrast(ev, None, None, None, w, o, optimizer="evk_bfgs", numeric_grads=False, blur_sigma=1.0, img_size=(H, W))


run(); run()
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); a = run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("n=%d %dx%d: %s ms -> %s" % (n, W, H, [round(v * 1e3, 3) for v in ts], np.round(a, 3)))
pr = cProfile.Profile()
pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
