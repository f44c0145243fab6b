"""direct vs one-pass / tiled time as a function of the event count: sets the 'auto' thresholds of the voxel grid
(tiled.TILED_MIN_EVENTS) and of the objective evaluation (tiled.TILED_MIN_EVENTS_IWE).  Round 4: re-measured with the one-pass
voxel path (round 1 measured the three-pass sort).   usage: python tools/crossover.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import event_utils_amd as E  # noqa: E402
from event_utils_amd import tiled  # noqa: E402
from event_utils_amd.representations.voxel_grid import _voxel_f32_device  # noqa: E402

B = 5
rng = np.random.default_rng(1)
for H, W in ((180, 240), (480, 640)):
    for n in (1000, 5000, 20_000, 50_000, 100_000, 200_000, 350_000, 600_000):
        x = torch.from_numpy(rng.integers(0, W, n).astype(np.float32)).cuda()
        y = torch.from_numpy(rng.integers(0, H, n).astype(np.float32)).cuda()
        t = torch.from_numpy(np.sort(rng.uniform(0, 0.1, n)).astype(np.float32)).cuda()
        p = torch.from_numpy((rng.integers(0, 2, n) * 2 - 1).astype(np.float32)).cuda()
        out = torch.empty((B, H, W), device="cuda")
        r = {impl: tiled._time_ms(lambda: _voxel_f32_device(x, y, t, p, B, (H, W), None, None, out=out, check=False, impl=impl, fresh=True), 50)
             for impl in ("direct", "tiled")}
        print("voxel %dx%dx%d n=%7d  direct %.1f us  one-pass %.1f us" % (B, H, W, n, r["direct"] * 1e3, r["tiled"] * 1e3), flush=True)

# objective evaluation (value, and value + gradient in one pass) on resident events: wall time per evaluation, bucketing cached
obj, w = E.variance_objective(), E.linvel_warp()
prm = np.array([30.0, -20.0])
for H, W in ((180, 240), (480, 640)):
    obj.sensor_size = (H, W)
    for n in (2000, 10_000, 20_000, 50_000, 100_000, 150_000, 300_000):
        ev = E.DeviceEvents.from_arrays(rng.uniform(1, W - 1, n).astype(np.float32), rng.uniform(1, H - 1, n).astype(np.float32),
                                        np.sort(rng.uniform(0, 0.1, n)).astype(np.float32), (rng.integers(0, 2, n) * 2 - 1).astype(np.float32))
        r = {}
        for impl in ("direct", "tiled"):
            os.environ["EVK_IMPL"] = impl
            for what, fn in (("f", lambda: obj.evaluate_function(prm, ev, None, None, None, w, (H, W), 1.0)),
                             ("fg", lambda: obj.evaluate_function_and_gradient(prm, ev, None, None, None, w, (H, W), 1.0))):
                for _ in range(5):
                    fn()
                t0 = time.perf_counter()
                for _ in range(100):
                    fn()
                r[impl, what] = (time.perf_counter() - t0) * 1e4
        # one evaluation of a NEW event set (columns already on the device): bucketing included
        cols = (ev.x, ev.y, ev.t, ev.p)
        for impl in ("direct", "tiled"):
            os.environ["EVK_IMPL"] = impl
            for _ in range(3):
                obj.evaluate_function(prm, E.DeviceEvents(*cols), None, None, None, w, (H, W), 1.0)
            t0 = time.perf_counter()
            for _ in range(30):
                obj.evaluate_function(prm, E.DeviceEvents(*cols), None, None, None, w, (H, W), 1.0)
            r[impl, "cold"] = (time.perf_counter() - t0) / 30 * 1e6
        os.environ.pop("EVK_IMPL")
        print("objective %dx%d n=%7d  first evaluation of a new event set: direct %.1f us tiled %.1f us" % (H, W, n, r["direct", "cold"], r["tiled", "cold"]))
        print("objective %dx%d n=%7d  f: direct %.1f us tiled %.1f us   f+grad: direct %.1f us tiled %.1f us"
              % (H, W, n, r["direct", "f"], r["tiled", "f"], r["direct", "fg"], r["tiled", "fg"]), flush=True)
