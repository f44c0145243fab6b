"""C4 (50 M events, 1280x720, moving-edge scene): evaluation times of f, analytic gradient and the three-flow numeric
gradient under the LDS accumulator modes (EVK_IWE_FIXED=64 | 32).  --px: sensor-pixel (integer) coordinates."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import event_utils_amd as E  # noqa: E402
from event_utils_amd.events import DeviceEvents  # noqa: E402

torch.cuda.set_device(0)
H, W, N = 720, 1280, 50_000_000
if "--vga" in sys.argv:      # the same scene at C3's size
    H, W, N = 480, 640, 10_000_000
if "--small" in sys.argv:
    H, W, N = 260, 346, 2_000_000
x, y, t, p = bench.structured_scene(3, N, H, W)
if "--px" in sys.argv:
    x, y = np.floor(x), np.floor(y)
ev = DeviceEvents.from_arrays(x, y, t, p, precision="f32")
w = E.linvel_warp()
prm = np.array([30.0, -20.0])
for fixed in ("64", "32", "64", "32"):
    os.environ["EVK_IWE_FIXED"] = fixed
    obj = E.variance_objective()
    obj.sensor_size = (H, W)
    row = []
    for name, fn in (("f", obj.evaluate_function), ("grad", obj.evaluate_gradient),
                     ("f+numgrad", obj.evaluate_function_and_numeric_gradient), ("f+grad", obj.evaluate_function_and_gradient)):
        for _ in range(2):
            fn(prm, ev, None, None, None, w, (H, W), 1.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn(prm, ev, None, None, None, w, (H, W), 1.0)
        torch.cuda.synchronize()
        row.append("%s %.4f" % (name, (time.perf_counter() - t0) / 20 * 1e3))
    print("FIXED=%s records=%s: %s ms" % (fixed, [b.iwe_flag for b in ev._buckets.values()], "  ".join(row)), flush=True)
