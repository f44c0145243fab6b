"""Objective landscape (draw_objective_function's 20 x 20 samples, events_cmax.py:103-131) on configs[2]-sized input:
K separate evaluations vs evaluate_function_batch (three nearby flows per pass, one readback).
  python tools/landscape_bench.py [n_events] [H] [W]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import event_utils_amd as E  # noqa: E402
from event_utils_amd.contrast_max import events_cmax as C  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 480
W = int(sys.argv[3]) if len(sys.argv) > 3 else 640
rng = np.random.default_rng(2)
x = rng.uniform(1, W - 1, n).astype(np.float32); y = rng.uniform(1, H - 1, n).astype(np.float32)
t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32); p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
ev = E.DeviceEvents.from_arrays(x, y, t, p)
obj = E.variance_objective(minimum_events=1)
obj.sensor_size = (H, W)
w = E.linvel_warp()
flows = [np.array([a * 20 - 200., b * 20 - 200.]) for a in range(20) for b in range(20)]
obj.evaluate_function(flows[0], ev, None, None, None, w, (H, W), 0)
for name, fn in (("single", lambda: [obj.evaluate_function(q, ev, None, None, None, w, (H, W), 0) for q in flows]),
                 ("batch", lambda: obj.evaluate_function_batch(flows, ev, None, None, None, w, (H, W), 0))):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vals = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-6s %d evaluations of %d events (%dx%d): %.1f ms = %.3f ms/eval, %.1f Gev/s  (checksum %.6f)" %
          (name, len(flows), n, W, H, dt * 1e3, dt * 1e3 / len(flows), len(flows) * n / dt / 1e9,
           float(np.sum(np.asarray(vals, dtype=np.float64)))))
t0 = time.perf_counter()
r = C.grid_search_optimisation(ev, None, None, None, w, obj, (H, W), log_scale=False)
print("grid_search_optimisation: %.1f ms -> %s (f = %.5f)" % ((time.perf_counter() - t0) * 1e3, r["min_params"],
                                                               r["min_func_eval"]))
