"""Pass-to-pass time of the repeated objective evaluation (value + gradient, one flow) through three host paths, to size what a
native optimiser loop can save: (a) the bound closure evk_bfgs uses, (b) the bare ctypes call with its cached arguments (no
Python arithmetic -- the upper bound of a loop written in C, plus ~5 us of ctypes marshalling), against (c) the kernels' own time
(the same call without host delivery, back to back between two HIP events).      python tools/again_gap.py [N H W]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import event_utils_amd as E  # noqa: E402
from event_utils_amd import tiled  # noqa: E402

n, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (10_000_000, 480, 640)
x, y, t, p = bench.structured_scene(3, n, H, W)
ev = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
o = E.variance_objective()
o.sensor_size, o.reference_exact = (H, W), False
fg, f3 = o.bind_fast(ev, None, None, None, E.linvel_warp(), (H, W), 1.0)
q = [38.0, -24.0]
pts = [[38.0, -24.0], [39.0, -24.5], [40.0, -25.0]]
fg(q); fg(q); f3(pts); f3(pts)
c1 = ev.__dict__["_cmax_last_single"]
c3 = ev.__dict__["_cmax_last_b3"]


def loop(fn, reps=300):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e6


def raw(c):
    args, st, fn = c["args"], c["spill"], c["fn"]

    def call():
        args[c["i_parity"]] = st[1] ^ 1
        rc = fn(*args)
        assert rc == 0
        st[1] ^= 1
    return call


print("n=%d %dx%d" % (n, W, H))
print("value+gradient: closure %.1f us   bare ctypes call %.1f us" % (loop(lambda: fg(q)), loop(raw(c1))))
print("three values  : closure %.1f us   bare ctypes call %.1f us" % (loop(lambda: f3(pts)), loop(raw(c3))))
# device time: the same calls without host delivery, enqueued back to back
for name, c, ih in (("value+gradient", c1, -2), ("three values", c3, -2)):
    args = list(c["args"])
    args[ih] = None
    st, fn = c["spill"], c["fn"]

    def call():
        args[c["i_parity"]] = st[1] ^ 1
        assert fn(*args) == 0
        st[1] ^= 1
    print("%s: kernels back to back %.1f us" % (name, tiled._time_ms(call, 100) * 1e3))
