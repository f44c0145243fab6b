"""One-pass voxel path (evk_voxel2.hip): correctness against the oracle and the three-pass path, then stage timings of
one kernel-geometry variant (EVK_V2_THREADS / EVK_V2_EPT / EVK_V2_WG / EVK_V2_G are read when the library loads, so
every variant is its own process: tools/v2_sweep.sh)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_utils_amd import tiled, _lib  # noqa: E402
from event_utils_amd.representations.voxel_grid import _voxel_f32_device  # noqa: E402


def synth(seed, n, H, W, wide=False):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, W, n).astype(np.float32)
    y = rng.integers(0, H, n).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    if wide:
        p = (p * rng.uniform(0.1, 3.0, n)).astype(np.float32)
    return x, y, t, p


def check():
    from oracle import reference_np as R
    for (n, H, W, B, wide) in ((600_000, 480, 640, 5, False), (1_000_003, 180, 240, 9, True), (400_001, 720, 1280, 3, False)):
        x, y, t, p = synth(n, n, H, W, wide)
        cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
        ref = R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64")
        os.environ["EVK_VOXEL_PATH"] = "v2"
        a = _voxel_f32_device(*cols, B, (H, W), None, None, impl="tiled").cpu().numpy().astype(np.float64)
        b = _voxel_f32_device(*cols, B, (H, W), float(t[0]), float(t[-1]), impl="tiled").cpu().numpy().astype(np.float64)
        os.environ["EVK_VOXEL_PATH"] = "v1"
        c = _voxel_f32_device(*cols, B, (H, W), float(t[0]), float(t[-1]), impl="tiled").cpu().numpy().astype(np.float64)
        os.environ["EVK_VOXEL_PATH"] = "v2"
        tol = 1e-5 * np.abs(ref).max()
        print("check n=%d %dx%dx%d wide=%s: v2(dev t) %.2e  v2 %.2e  v1 %.2e  tol %.2e  mass %.6f/%.6f" % (
            n, H, W, B, wide, np.abs(a - ref).max(), np.abs(b - ref).max(), np.abs(c - ref).max(), tol, a.sum(), ref.sum()))
        assert np.abs(a - ref).max() <= tol and np.abs(b - ref).max() <= tol
    # repeated calls on the same persistent index (self-resetting counters) and a clustered scene (split hot tiles)
    n, H, W, B = 3_000_000, 480, 640, 5
    x, y, t, p = synth(7, n, H, W)
    x[: n // 2] = 100 + (x[: n // 2] % 8)
    y[: n // 2] = 50 + (y[: n // 2] % 8)
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    ref = R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64")
    for k in range(3):
        a = _voxel_f32_device(*cols, B, (H, W), None, None, impl="tiled").cpu().numpy().astype(np.float64)
        print("clustered run %d: err %.2e tol %.2e" % (k, np.abs(a - ref).max(), 1e-5 * np.abs(ref).max()))
        assert np.abs(a - ref).max() <= 1e-5 * np.abs(ref).max()


def timing(n, H, W, B, reps=20):
    x, y, t, p = synth(1, n, H, W)
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    res = {}
    for path in (("v2",) if "--v2only" in sys.argv else ("v2", "v1")):
        os.environ["EVK_VOXEL_PATH"] = path
        k = tiled.time_voxel_kernels(*cols, float(t[0]), float(t[-1]), B, H, W, impl="tiled", reps=reps)
        res[path] = k
        alg = 16.0 * n + B * H * W * 4
        print("%s n=%d %dx%dx%d: total %.4f ms (%.1f Gev/s, whole-call frac %.3f)  %s  [%s]" % (
            path, n, H, W, B, k["total_ms"], n / k["total_ms"] / 1e6, alg / (k["total_ms"] * 1e-3) / 8e12,
            k["kernels_ms"], k["impl"]), flush=True)
    os.environ["EVK_VOXEL_PATH"] = "v2"
    return res


if __name__ == "__main__":
    torch.cuda.set_device(0)
    print("variant: THREADS=%s EPT=%s WG=%s U=%s LIB=%s" % tuple(os.environ.get(k, "-") for k in
          ("EVK_V2_PART", "EVK_V2_PART", "EVK_V2_WG", "EVK_V2_U", "EVK_LIB_PATH")), flush=True)
    if "--check" in sys.argv:
        check()
    timing(10_000_000, 480, 640, 5)
    if "--big" in sys.argv:
        timing(50_000_000, 720, 1280, 5, reps=10)
