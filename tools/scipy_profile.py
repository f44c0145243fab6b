"""cProfile of optimize_contrast with the reference's default optimiser (scipy fmin_bfgs; numeric gradients or analytic) at a
small size: how the wall time splits between scipy's own code, this package's Python and the library calls.
    python tools/scipy_profile.py [N H W] [--analytic]"""
import cProfile
import os
import pstats
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import event_utils_amd as E  # noqa: E402
from event_utils_amd.contrast_max.events_cmax import optimize_contrast  # noqa: E402

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
n, H, W = (int(v) for v in argv[:3]) if len(argv) >= 3 else (100_000, 180, 240)
numeric = "--analytic" not in sys.argv
x, y, t, p = bench.structured_scene(3, n, H, W)
ev = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
w = E.linvel_warp()
warnings.simplefilter("ignore")
calls = [0]


def run():
    o = E.variance_objective()
    o.sensor_size, o.reference_exact = (H, W), False
    a = optimize_contrast(ev, None, None, None, w, o, numeric_grads=numeric, blur_sigma=1.0, img_size=(H, W))
    return a


run(); run()
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); a = run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("n=%d %dx%d numeric=%s: %s ms -> %s" % (n, W, H, numeric, [round(v * 1e3, 2) for v in ts], np.round(a, 3)))
pr = cProfile.Profile()
pr.enable(); run(); pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
