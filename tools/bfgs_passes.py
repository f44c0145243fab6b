"""Per-pass wall time of one warm optimize_contrast(optimizer='evk_bfgs') run: every fg / f3 call of the bound closures is timed,
and what is left of the run's wall time is the optimiser's own share (set-up, arithmetic, callbacks).
    python tools/bfgs_passes.py [N H W]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import event_utils_amd as E  # noqa: E402
from event_utils_amd.contrast_max import objectives  # noqa: E402
from event_utils_amd.contrast_max.events_cmax import optimize_contrast  # noqa: E402

n, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (10_000_000, 480, 640)
x, y, t, p = bench.structured_scene(3, n, H, W)
ev = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
w = E.linvel_warp()
log = []
orig = objectives.variance_objective.bind_fast


def bind(self, *a):
    r = orig(self, *a)
    if r is None:
        return None
    fg, f3 = r

    def tfg(q):
        t0 = time.perf_counter(); v = fg(q); log.append(("fg", t0, time.perf_counter(), tuple(q))); return v

    def tf3(pts):
        t0 = time.perf_counter(); v = f3(pts); log.append(("f3", t0, time.perf_counter(), tuple(pts[1]))); return v
    return tfg, tf3


objectives.variance_objective.bind_fast = bind


def run():
    o = E.variance_objective()
    o.sensor_size, o.reference_exact = (H, W), False
    return optimize_contrast(ev, None, None, None, w, o, optimizer="evk_bfgs", numeric_grads=False, blur_sigma=1.0, img_size=(H, W))


run(); run()
for _ in range(3):
    log.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter(); a = run(); torch.cuda.synchronize(); t1 = time.perf_counter()
    inside = sum(e[2] - e[1] for e in log)
    print("n=%d: run %.1f us, %d passes %.1f us inside, before first %.1f us, after last %.1f us, between passes %s" % (
        n, (t1 - t0) * 1e6, len(log), inside * 1e6, (log[0][1] - t0) * 1e6, (t1 - log[-1][2]) * 1e6,
        [round((log[i + 1][1] - log[i][2]) * 1e6, 1) for i in range(len(log) - 1)]))
print(" ".join("%s %.0f" % (e[0], (e[2] - e[1]) * 1e6) for e in log))
print(np.round(a, 3))
