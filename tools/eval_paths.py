"""One objective evaluation on resident events through every host path, wall time per call in a tight loop: the public
methods (what scipy calls back), the bound closures (objective.bind_fast), the bare library call, and the kernels alone
(back to back).    python tools/eval_paths.py [N H W]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import event_utils_amd as E  # noqa: E402
from event_utils_amd import tiled  # noqa: E402

n, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (100_000, 180, 240)
x, y, t, p = bench.structured_scene(3, n, H, W)
ev = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
ev.many_evaluations = True
w = E.linvel_warp()
o = E.variance_objective()
o.sensor_size, o.reference_exact = (H, W), False
args = (ev, None, None, None, w, (H, W), 1.0)
q = np.array([38.0, -24.0])
fg, f3 = o.bind_fast(*args)


def loop(fn, reps=400):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e6


ql = [38.0, -24.0]
pts = [[38.0, -24.0], [39.0, -24.0], [38.0, -23.0]]
fg(ql); f3(pts)
print("n=%d %dx%d" % (n, W, H))
print("f                    : public %.1f us" % loop(lambda: o.evaluate_function(q, *args)))
print("f + gradient         : public %.1f us   closure %.1f us" % (loop(lambda: o.evaluate_function_and_gradient(q, *args)), loop(lambda: fg(ql))))
print("f + numeric gradient : public %.1f us   closure %.1f us" % (loop(lambda: o.evaluate_function_and_numeric_gradient(q, *args)), loop(lambda: f3(pts))))
for name, c in (("f + gradient", ev.__dict__["_cmax_last_single"]), ("three flows", ev.__dict__["_cmax_last_b3"])):
    a, st, fn = c["args"], c["spill"], c["fn"]

    def call():
        a[c["i_parity"]] = st[1] ^ 1
        assert fn(*a) == 0
        st[1] ^= 1
    bare = loop(call)
    a2 = list(a)
    a2[-2] = None

    def call2():
        a2[c["i_parity"]] = st[1] ^ 1
        assert fn(*a2) == 0
        st[1] ^= 1
    print("%-21s: bare library call %.1f us   kernels back to back %.1f us" % (name, bare, tiled._time_ms(call2, 100) * 1e3))
