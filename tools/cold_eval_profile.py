"""The FIRST evaluation of a new event set (columns already on the device): wall time against a repeated evaluation, and a
cProfile of where the host spends the difference (bucketing, read-backs, allocations).  python tools/cold_eval_profile.py [N H W]"""
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

n, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (100_000, 180, 240)
x, y, t, p = bench.structured_scene(3, n, H, W)
ev0 = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
w = E.linvel_warp()
o = E.variance_objective()
o.sensor_size, o.reference_exact = (H, W), False
q = np.array([38.0, -24.0])


def fresh():
    e = ev0.fresh_view()
    e.many_evaluations = True
    return e


def cold():
    return o.evaluate_function_and_gradient(q, fresh(), None, None, None, w, (H, W), 1.0)


for _ in range(5):
    cold()
ts = []
for _ in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter(); cold(); ts.append(time.perf_counter() - t0)
e = fresh(); cold_e = o.evaluate_function_and_gradient(q, e, None, None, None, w, (H, W), 1.0)
tw = []
for _ in range(30):
    t0 = time.perf_counter(); o.evaluate_function_and_gradient(q, e, None, None, None, w, (H, W), 1.0); tw.append(time.perf_counter() - t0)
print("n=%d %dx%d: first evaluation of a new set %.1f us (median), repeated %.1f us" % (n, W, H, np.median(ts) * 1e6, np.median(tw) * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    cold()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)

# stage timers: the bucketing call (enqueue only), the 12-byte read-back that ends it (waits for the kernels), the rest
from event_utils_amd import tiled  # noqa: E402
acc = {"bucket_events": 0.0, "_tail": 0.0, "compact": 0.0, "total": 0.0}
orig_b, orig_t, orig_c = tiled.bucket_events, tiled.Buckets._tail, tiled.Buckets.compact


def timed(name, fn):
    def wrap(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc[name] += time.perf_counter() - t0
        return r
    return wrap


tiled.bucket_events = timed("bucket_events", orig_b)
tiled.Buckets._tail = timed("_tail", orig_t)
tiled.Buckets.compact = timed("compact", orig_c)
R = 50
for _ in range(R):
    torch.cuda.synchronize(); t0 = time.perf_counter(); cold(); acc["total"] += time.perf_counter() - t0
print({k: round(v / R * 1e6, 1) for k, v in acc.items()}, "(us per cold evaluation; _tail and compact nest: compact calls _tail)")
