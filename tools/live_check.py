"""EVK_VOXEL2_LIVE (evk_voxel_live.h): the voxel tiles accumulated while the partition sorts.  Correctness of the live call
against the two-launch call (bit for bit: the same integer sums) and the oracle on uniform events, structured scenes (hot tiles
are LEFT to the tile kernel proper), arbitrary polarities (everything is left), then timings from HBM (4 rotating streams).
  python tools/live_check.py [--quick] [--time-only]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_utils_amd import tiled  # noqa: E402
from event_utils_amd.representations.voxel_grid import _voxel_f32_device  # noqa: E402
from tools.voxel_sweep import scene, synth  # noqa: E402

H, W, B = 480, 640, 5


def grid(cols, live, **kw):
    tiled.FORCE["live"] = live
    try:
        return _voxel_f32_device(*cols, B, (H, W), None, None, impl="tiled", **kw)
    finally:
        tiled.FORCE["live"] = None


def check(n):
    from oracle import reference_np as R
    ok = True
    for seed, kind in ((1, "pm1"), (2, "pm1"), (3, "zero_one"), (4, "unsorted"), (5, "mixed"), (6, "wide"), (7, "edges"), (8, "blob"),
                       (9, "pm1")):
        if kind in ("edges", "blob"):
            x, y, t, p = scene(kind, n, H, W)
        else:
            x, y, t, p = synth(seed, n, H, W, kind)
        cols = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (x, y, t, p)]
        a = grid(cols, True)
        b = grid(cols, False)
        a2 = grid(cols, True)
        torch.cuda.synchronize()
        ref = R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64")
        ea = np.abs(a.cpu().numpy().astype(np.float64) - ref).max()
        tol = 1e-5 * np.abs(ref).max()
        same, again = torch.equal(a, b), torch.equal(a, a2)
        d = (a - b).abs().max().item()
        print("check %-8s n=%d: live vs oracle %.2e (tol %.2e)  live == two-launch: %s (max diff %.2e)  live twice: %s" % (
            kind, n, ea, tol, same, d, again), flush=True)
        ok &= ea <= tol and d <= tol and again
    return ok


def timing(n, reps=20):
    x, y, t, p = synth(1, n, H, W)
    sets = [[torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (x, y, t, p)]]
    for k in range(1, 4):
        xs, ys, ts_, ps = synth(1000 * k + 1, n, H, W)
        ts_[0], ts_[-1] = t[0], t[-1]
        sets.append([torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (xs, ys, ts_, ps)])
    out = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
    it = [0]

    def call(live):
        it[0] += 1
        tiled.FORCE["live"] = live
        _voxel_f32_device(*sets[it[0] % 4], B, (H, W), None, None, out=out, check=False, impl="tiled", fresh=True)
    alg = 16.0 * n + B * H * W * 4
    for rnd in range(3):
        for live in (False, True):
            ms = tiled._time_ms(lambda: call(live), reps)
            print("time n=%d live=%-5s: %.4f ms per call  (%.1f Gev/s, whole-call frac %.3f)" % (
                n, live, ms, n / ms / 1e6, alg / (ms * 1e-3) / 8e12), flush=True)
    tiled.FORCE["live"] = None


if __name__ == "__main__":
    torch.cuda.set_device(0)
    n = 3_000_000 if "--quick" in sys.argv else 10_000_000
    t0 = time.time()
    good = True
    if "--time-only" not in sys.argv:
        good = check(n)
    timing(10_000_000)
    print("live_check: %s  (%.1f s)" % ("OK" if good else "MISMATCH", time.time() - t0), flush=True)
    sys.exit(0 if good else 1)
