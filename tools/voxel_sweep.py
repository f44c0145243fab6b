"""One-pass voxel path (evk_voxel2.hip): correctness against the oracle on the cases that stress the record formats (escapes
of the 4-byte records: arbitrary polarities, unsorted / sparse time stamps; hot tiles; deterministic mode), then stage
timings (REC=4|8 in the environment of this script forces a record size through tiled.FORCE)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_utils_amd import tiled  # noqa: E402
from event_utils_amd.representations.voxel_grid import _voxel_f32_device  # noqa: E402


def synth(seed, n, H, W, kind="pm1"):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, W, n).astype(np.float32)
    y = rng.integers(0, H, n).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    if kind == "wide":
        p = (p * rng.uniform(0.1, 3.0, n)).astype(np.float32)
    elif kind == "zero_one":
        p = rng.integers(0, 2, n).astype(np.float32)
    elif kind == "unsorted":
        t = rng.permutation(t)
        t[0], t[-1] = 0.0, 0.1
    elif kind == "mixed":
        p[::7] = 0.25
        t[n // 2: n // 2 + 1000] = t[n // 2: n // 2 + 1000][::-1]
    return x, y, t, p


def check():
    from oracle import reference_np as R
    cases = [(600_000, 480, 640, 5, "pm1"), (1_000_003, 180, 240, 9, "wide"), (400_001, 720, 1280, 3, "zero_one"),
             (500_000, 480, 640, 5, "unsorted"), (700_001, 260, 346, 5, "mixed"), (360_000, 480, 640, 1, "pm1")]
    for (n, H, W, B, kind) in cases:
        x, y, t, p = synth(n, n, H, W, kind)
        cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
        ref = R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64")
        a = _voxel_f32_device(*cols, B, (H, W), None, None, impl="tiled").cpu().numpy().astype(np.float64)
        b = _voxel_f32_device(*cols, B, (H, W), float(t[0]), float(t[-1]), impl="tiled").cpu().numpy().astype(np.float64)
        tol = 1e-5 * np.abs(ref).max()
        ea, eb = np.abs(a - ref).max(), np.abs(b - ref).max()
        print("check n=%d %dx%dx%d %-8s: device ts[0]/ts[-1] %.2e  host %.2e  tol %.2e  mass %.6f/%.6f  nan %d/%d" % (
            n, H, W, B, kind, ea, eb, tol, np.nansum(a), np.nansum(ref), np.isnan(a).sum(), np.isnan(ref).sum()), flush=True)
        assert ea <= tol and eb <= tol
    # repeated calls on the same persistent index (self-resetting counters) and a clustered scene (split hot tiles)
    n, H, W, B = 3_000_000, 480, 640, 5
    x, y, t, p = synth(7, n, H, W)
    x[: n // 2] = 100 + (x[: n // 2] % 8)
    y[: n // 2] = 50 + (y[: n // 2] % 8)
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    ref = R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64")
    for k in range(3):
        a = _voxel_f32_device(*cols, B, (H, W), None, None, impl="tiled").cpu().numpy().astype(np.float64)
        print("clustered run %d: err %.2e tol %.2e" % (k, np.abs(a - ref).max(), 1e-5 * np.abs(ref).max()), flush=True)
        assert np.abs(a - ref).max() <= 1e-5 * np.abs(ref).max()
    # deterministic mode: bit-identical from run to run, and within the bar
    os.environ["EVK_VOXEL_DETERMINISTIC"] = "1"
    g = [_voxel_f32_device(*cols, B, (H, W), None, None, impl="tiled").cpu().numpy() for _ in range(3)]
    os.environ["EVK_VOXEL_DETERMINISTIC"] = "0"
    print("deterministic: equal %s %s err %.2e" % (np.array_equal(g[0], g[1]), np.array_equal(g[0], g[2]),
                                                 np.abs(g[0].astype(np.float64) - ref).max()), flush=True)
    assert np.array_equal(g[0], g[1]) and np.array_equal(g[0], g[2])


def scene(kind, n, H, W):
    import bench
    if kind == "edges":
        x, y, t, p = bench.structured_scene(3, n, H, W)
        return np.floor(x), np.floor(y), t, p
    rng = np.random.default_rng(2)   # half of the events inside a 100 x 100 px blob
    x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
    hot = rng.random(n) < 0.5
    x[hot] = (W // 3 + rng.integers(0, 100, hot.sum())).astype(np.float32)
    y[hot] = (H // 3 + rng.integers(0, 100, hot.sum())).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32); p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    return x, y, t, p


def timing(n, H, W, B, reps=20, paths=("v2",), kind="uniform"):
    x, y, t, p = synth(1, n, H, W) if kind == "uniform" else scene(kind, n, H, W)
    cols = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (x, y, t, p)]
    sets = [cols]
    if "--rotate" in sys.argv and kind == "uniform":      # 4 distinct streams: every call reads its events from HBM
        for k in range(1, 4):
            xs, ys, ts_, ps = synth(1000 * k + 1, n, H, W)
            ts_[0], ts_[-1] = t[0], t[-1]
            sets.append([torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (xs, ys, ts_, ps)])
    for path in paths:
        k = tiled.time_voxel_kernels(sets, float(t[0]), float(t[-1]), B, H, W, impl="tiled", reps=reps)
        alg = 16.0 * n + B * H * W * 4
        print("%s %-7s n=%d %dx%dx%d: total %.4f ms (%.1f Gev/s, whole-call frac %.3f)  %s  [%s]" % (
            path, kind, n, H, W, B, k["total_ms"], n / k["total_ms"] / 1e6, alg / (k["total_ms"] * 1e-3) / 8e12,
            k["kernels_ms"], k["impl"]), flush=True)


def native_timing(n, H, W, B, reps=20):
    """13 B/event (int16 x, y; float64 epoch-second t; uint8 p) against the same events as four float32 columns."""
    from event_utils_amd.events import DeviceEvents
    x, y, t, p = synth(1, n, H, W)
    ev = DeviceEvents.from_native(x.astype(np.int16), y.astype(np.int16), 1.6e9 + t.astype(np.float64), ((p + 1) / 2).astype(np.uint8))
    nat = ev.native
    out = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
    t_first, t_last = ev.t_at(0), ev.t_at(-1)
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    f32 = lambda: _voxel_f32_device(*cols, B, (H, W), t_first, t_last, out=out, check=False, impl="tiled", fresh=True)
    nat_fn = lambda: _voxel_f32_device(None, None, None, None, B, (H, W), t_first, t_last, out=out, check=False, impl="tiled", fresh=True, native=nat)
    a = tiled._time_ms(f32, reps); b = tiled._time_ms(nat_fn, reps); a2 = tiled._time_ms(f32, reps); b2 = tiled._time_ms(nat_fn, reps)
    print("native n=%d %dx%dx%d: float32 columns (16 B/ev) %.4f / %.4f ms, on-disk dtypes (13 B/ev) %.4f / %.4f ms" % (n, H, W, B, a, a2, b, b2), flush=True)


if __name__ == "__main__":
    torch.cuda.set_device(0)
    if os.environ.get("REC") in ("4", "8"):
        tiled.FORCE["rec"] = int(os.environ["REC"])
    if os.environ.get("COUNT2") == "0":
        tiled.FORCE["count2"] = False
    print("variant: REC=%s COUNT2=%s LIB=%s" % (os.environ.get("REC", "-"), os.environ.get("COUNT2", "-"), os.environ.get("EVK_LIB_PATH", "-")), flush=True)
    if "--check" in sys.argv:
        check()
    paths = ("v2", "v1") if "--v1" in sys.argv else ("v2",)
    timing(10_000_000, 480, 640, 5, paths=paths)
    if "--scenes" in sys.argv:
        timing(10_000_000, 480, 640, 5, paths=paths, kind="edges")
        timing(10_000_000, 480, 640, 5, paths=paths, kind="blob")
    if "--native" in sys.argv:
        native_timing(10_000_000, 480, 640, 5)
    if "--big" in sys.argv:
        timing(50_000_000, 720, 1280, 5, reps=10, paths=paths)
        if "--native" in sys.argv:
            native_timing(50_000_000, 720, 1280, 5, reps=10)
