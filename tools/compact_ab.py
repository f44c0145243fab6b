"""Compact (8-byte) against full (16-byte) bucketed records on sensor-pixel events (integer x, y; +-1 polarity): the
IWE / dIWE must be bit-identical, then f / grad evaluation times at C3 (10 M events, 640x480) and C4 (50 M, 1280x720)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import event_utils_amd as E  # noqa: E402
from event_utils_amd import tiled  # noqa: E402
from event_utils_amd.events import DeviceEvents  # noqa: E402


def evals(x, y, t, p, size, reps):
    res = {}
    for mode in ("full", "compact"):
        tiled.FORCE["iwe_records"] = mode
        ev = DeviceEvents.from_arrays(x, y, t, p, precision="f32")
        obj, w = E.variance_objective(), E.linvel_warp()
        obj.sensor_size = size
        prm = np.array([30.0, -20.0])
        iwe, d = E.get_iwe(prm, ev, None, None, None, w, size, sensor_size=size, compute_gradient=True)
        r = bench._time_evals(obj, w, ev, prm, size, reps=reps)
        bk = list(ev._buckets.values())
        res[mode] = (np.asarray(iwe), np.asarray(d), r, [b.iwe_flag for b in bk])
        print(mode, "flags", res[mode][3], {k: v for k, v in r.items() if k.endswith("_ms")}, flush=True)
        del ev
        torch.cuda.empty_cache()
    a, b = res["full"], res["compact"]
    print("IWE identical:", np.array_equal(a[0], b[0]), " dIWE identical:", np.array_equal(a[1], b[1]))
    assert b[3] and all(b[3]) and not any(a[3])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


if __name__ == "__main__":
    torch.cuda.set_device(0)
    x, y, t, p = bench.synth(2, 10_000_000, 0.0, 0.1, real_xy=False)
    evals(x, y, t, p, (480, 640), 20)
    if "--big" in sys.argv:
        x, y, t, p = bench.structured_scene(3, 50_000_000, 720, 1280)
        evals(np.floor(x), np.floor(y), t, p, (720, 1280), 5)
