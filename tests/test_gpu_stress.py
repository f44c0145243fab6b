"""GPU (-m gpu): randomized tiled-vs-direct equivalence (both are HIP paths; the direct one is the simplest possible
kernel and is itself pinned to the oracle in test_gpu_parity.py) over odd sensor sizes, bin counts, event counts, flows,
clustered and degenerate inputs."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(a, b, tol=1e-5):
    a, b = a.double(), b.double()
    scale = max(b.abs().max().item(), 1e-30)
    assert (a - b).abs().max().item() <= tol * scale


@pytest.mark.parametrize("seed", range(12))
def test_random_voxel_tiled_equals_direct(seed):
    from event_utils_amd.representations.voxel_grid import _voxel_f32_device
    rng = np.random.default_rng(1000 + seed)
    H, W = int(rng.integers(5, 300)), int(rng.integers(5, 400))
    B = int(rng.integers(1, 10))
    n = int(rng.choice([17, 300, 4099, 70_001, 333_333]))
    kind = seed % 4
    x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
    if kind == 1:       # clustered
        hot = rng.random(n) < 0.8
        x[hot] = W // 2; y[hot] = H // 2
    if kind == 2:       # fractional + a few negative (wrap) coordinates
        x = np.clip(x + rng.random(n).astype(np.float32), 0, W - 1).astype(np.float32)
        x[:3] = -1.0
    t = np.sort(rng.uniform(5.0, 5.1, n)).astype(np.float32)
    p = rng.normal(size=n).astype(np.float32) if kind == 3 else (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    a = _voxel_f32_device(*cols, B, (H, W), float(t[0]), float(t[-1]), impl="tiled")
    b = _voxel_f32_device(*cols, B, (H, W), float(t[0]), float(t[-1]), impl="direct")
    # the direct kernel sums in float32 atomics: with 80 % of the events on ONE pixel its own rounding reaches ~1e-5 of
    # the hot cell (the tiled path accumulates in float64), so the clustered scenes get a wider band
    _close(a, b, 1e-4 if kind == 1 else 1e-5)
    assert abs(a.double().sum().item() - float(p.astype(np.float64).sum())) <= 1e-4 * max(1.0, np.abs(p).sum())


@pytest.mark.parametrize("seed", range(12))
def test_random_iwe_tiled_equals_direct(seed):
    from event_utils_amd.contrast_max.objectives import iwe_device
    from event_utils_amd.events import DeviceEvents
    rng = np.random.default_rng(2000 + seed)
    H, W = int(rng.integers(20, 300)), int(rng.integers(20, 400))
    n = int(rng.choice([33, 2000, 50_001, 250_000]))
    x = rng.uniform(-2, W + 2, n).astype(np.float32); y = rng.uniform(-2, H + 2, n).astype(np.float32)
    if seed % 3 == 1:
        hot = rng.random(n) < 0.85
        x[hot] = (W / 3 + rng.random(hot.sum())).astype(np.float32); y[hot] = (H / 2 + rng.random(hot.sum())).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.2, n)).astype(np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32) * (1 if seed % 4 else 37.5)
    prm = rng.normal(0, [10, 100, 1000][seed % 3], 2)
    ev = DeviceEvents.from_arrays(x, y, t, p)
    if seed % 5 == 0:
        ev = ev.scaled(100.0)
    img_size = (H, W) if seed % 2 else (H + 7, W + 11)
    for grad in (False, True):
        for pol in (True, False):
            a = iwe_device(prm, ev, img_size, grad, pol, (H, W), impl="tiled")
            b = iwe_device(prm, ev, img_size, grad, pol, (H, W), impl="direct")
            _close(a[0], b[0])
            if grad:
                _close(a[1], b[1])


def test_objective_values_are_reproducible_run_to_run():
    """Fixed-point LDS windows + sorted gather lists + two-stage reductions: repeated evaluations are bit-identical."""
    import event_utils_amd as E
    from event_utils_amd.events import DeviceEvents
    rng = np.random.default_rng(7)
    H, W, n = 240, 320, 600_000
    x = rng.uniform(1, W - 1, n).astype(np.float32); y = rng.uniform(1, H - 1, n).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32); p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    ev = DeviceEvents.from_arrays(x, y, t, p)
    obj, w = E.variance_objective(), E.linvel_warp()
    obj.sensor_size = (H, W)
    prm = np.array([33.3, -21.7])
    fs = {float(obj.evaluate_function(prm, ev, None, None, None, w, (H, W), 1.0)) for _ in range(5)}
    gs = {tuple(obj.evaluate_gradient(prm, ev, None, None, None, w, (H, W), 1.0).tolist()) for _ in range(5)}
    assert len(fs) == 1 and len(gs) == 1


@pytest.mark.parametrize("cfg", [("configs[2]", 10_000_000, 480, 640), ("configs[3]", 50_000_000, 720, 1280)])
def test_full_size_iwe_properties(cfg, monkeypatch):
    """BASELINE.json's contrast-maximisation sizes, where the oracle would take minutes: size-independent properties.
    (1) mass: the four bilinear weights of an accepted event sum to 1, so sum(IWE) = sum of the accepted polarities
    (here: all of them, the flow keeps every event inside), and sum(dIWE) = 0 (a derivative of weights that sum to a
    constant); (2) additivity: IWE(first half) + IWE(second half) = IWE(all), each half warped to the global reference
    time (what the multi-GPU all-reduce relies on); (3) the tile-bucketed and the direct global-atomic kernels agree."""
    import event_utils_amd as E
    from event_utils_amd.contrast_max.objectives import iwe_device
    _, n, H, W = cfg
    rng = np.random.default_rng(7)
    x = rng.uniform(8, W - 8, n).astype(np.float32); y = rng.uniform(8, H - 8, n).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32); p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    prm = np.array([30., -20.])                         # at most 3 px of displacement: nothing leaves the sensor
    ev = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
    monkeypatch.setenv("EVK_IMPL", "tiled")
    iwe, diwe = iwe_device(prm, ev, (H, W), True, True, (H, W))
    total = float(p.astype(np.float64).sum())
    assert abs(iwe.double().sum().item() - total) <= 1e-6 * n ** 0.5 + 1e-3
    assert abs(diwe[0].double().sum().item()) <= 1e-4 * n ** 0.5 and abs(diwe[1].double().sum().item()) <= 1e-4 * n ** 0.5
    half = n // 2
    parts = [iwe_device(prm, ev.slice(a, b), (H, W), True, True, (H, W), t_ref=ev.t_at(-1))
             for a, b in ((0, half), (half, n))]
    _close(parts[0][0] + parts[1][0], iwe)
    _close(parts[0][1] + parts[1][1], diwe)
    monkeypatch.setenv("EVK_IMPL", "direct")
    iwe_d, diwe_d = iwe_device(prm, ev, (H, W), True, True, (H, W))
    _close(iwe, iwe_d)
    _close(diwe, diwe_d)
    # the objective of the whole evaluation chain, tiled (one library call) vs direct kernels
    obj, w = E.variance_objective(), E.linvel_warp()
    obj.sensor_size = (H, W)
    fd, gd = obj.evaluate_function(prm, ev, None, None, None, w, (H, W), 1.0), obj.evaluate_gradient(prm, ev, None, None, None, w, (H, W), 1.0)
    monkeypatch.setenv("EVK_IMPL", "tiled")
    ft, gt = obj.evaluate_function_and_gradient(prm, ev, None, None, None, w, (H, W), 1.0)
    assert abs(float(ft) - float(fd)) <= 1e-5 * abs(float(fd))
    assert np.max(np.abs(np.asarray(gt, float) - np.asarray(gd, float))) <= 1e-5 * np.max(np.abs(np.asarray(gd, float))) + 1e-9


def test_full_size_compact_records_and_balanced_plan(monkeypatch):
    """50 M sensor events (integer pixels) at 1280x720: beyond the Infinity Cache the bucketing compacts its records by
    itself (tiled.FORCE["iwe_records"] = "auto") and, on the moving-edge scene, balances the plan.  Mass conservation, and IWE / dIWE
    equal to those from the 16-byte records (same plan, same sums: bit-identical up to the float atomics of the few events
    that leave their windows)."""
    import bench
    import event_utils_amd as E
    from event_utils_amd import tiled
    from event_utils_amd.contrast_max.objectives import iwe_device
    n, H, W = 50_000_000, 720, 1280
    x, y, t, p = bench.structured_scene(3, n, H, W)
    x, y = np.floor(x), np.floor(y)
    prm = np.array([30., -20.])
    monkeypatch.setenv("EVK_IMPL", "tiled")
    out = {}
    for mode in ("auto", "full"):
        monkeypatch.setitem(tiled.FORCE, "iwe_records", mode)
        ev = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
        iwe, diwe = iwe_device(prm, ev, (H, W), True, True, (H, W))
        bk = list(ev._buckets.values())
        assert len(bk) == 1 and bool(bk[0].iwe_flag) == (mode == "auto") and bk[0].structured
        out[mode] = (iwe, diwe)
        del ev
    inside = (x + 3.0 < W) & (y - 2.0 > 0)          # the flow moves an event by at most (+3, -2) px
    assert inside.all()
    assert abs(out["auto"][0].double().sum().item() - float(p.astype(np.float64).sum())) <= 1e-6 * n ** 0.5 + 1e-3
    _close(out["auto"][0], out["full"][0])
    _close(out["auto"][1], out["full"][1])


def test_full_size_voxel_per_gpu_share_of_configs4(monkeypatch):
    """One rank's share of configs[4]: 50 M events, 1280x720, 5 bins.  Mass conservation (the two temporal weights of
    an event sum to 1: sum(grid) = sum(p)), tiled == direct, and additivity of two half streams voxelised against the
    GLOBAL time range (the all-reduce identity)."""
    from event_utils_amd.representations.voxel_grid import _voxel_f32_device
    H, W, B, n = 720, 1280, 5, 50_000_000
    rng = np.random.default_rng(9)
    cols = [torch.from_numpy(a).cuda() for a in (rng.integers(0, W, n).astype(np.float32),
                                                rng.integers(0, H, n).astype(np.float32),
                                                np.sort(rng.uniform(0, 0.1, n)).astype(np.float32),
                                                (rng.integers(0, 2, n) * 2 - 1).astype(np.float32))]
    t0, t1 = cols[2][0].item(), cols[2][-1].item()
    monkeypatch.setenv("EVK_IMPL", "tiled")
    vt = _voxel_f32_device(*cols, B, (H, W), t0, t1)
    assert abs(vt.double().sum().item() - cols[3].double().sum().item()) <= 1e-3 * n ** 0.5
    half = (n // 2) & ~3                                  # keeps the second half's columns 16-byte aligned
    va = _voxel_f32_device(*(c[:half] for c in cols), B, (H, W), t0, t1)
    vb = _voxel_f32_device(*(c[half:] for c in cols), B, (H, W), t0, t1)
    _close(va + vb, vt)
    monkeypatch.setenv("EVK_IMPL", "direct")
    _close(vt, _voxel_f32_device(*cols, B, (H, W), t0, t1))


@pytest.mark.parametrize("impl", ["auto", "tiled"])
@pytest.mark.parametrize("seed", range(16))
def test_random_event_image_variants_equal_the_oracle(seed, impl, monkeypatch):
    """events_to_image_torch over its whole option space -- nearest / bilinear, padding, clip_out_of_range, default,
    integer or real coordinates, coordinates beyond the sensor (clipped ones pile up at (0, 0) with their weight, Q8) --
    and events_to_image (numpy path, meanval / bilinear) against the oracle on random inputs."""
    import event_utils_amd as E
    from oracle import reference_np as R
    monkeypatch.setenv("EVK_IMPL", impl)   # "tiled": one-pass partition + LDS tiles at any size (evk_image2.hip)
    rng = np.random.default_rng(4000 + seed)
    H, W = int(rng.integers(4, 90)), int(rng.integers(4, 120))
    n = int(rng.choice([1, 33, 5000, 60_000]))
    interp = [None, "bilinear"][seed % 2]
    padding = bool((seed >> 1) & 1)
    clip = bool((seed >> 2) & 1) or interp == "bilinear"          # un-clipped bilinear out-of-range raises upstream
    default = float(rng.choice([0.0, 0.5]))
    real = bool((seed >> 3) & 1)
    hi_x, hi_y = (W + 3, H + 3) if clip else (W - 1, H - 1)        # beyond the sensor only when clipping is on
    if real:
        x = rng.uniform(0, hi_x, n).astype(np.float32); y = rng.uniform(0, hi_y, n).astype(np.float32)
        if not clip:
            x = np.minimum(x, np.float32(W - 1.001)); y = np.minimum(y, np.float32(H - 1.001))
    else:
        x = rng.integers(0, hi_x, n).astype(np.int64); y = rng.integers(0, hi_y, n).astype(np.int64)
    p = rng.normal(size=n).astype(np.float32)
    ref = R.events_to_image_torch(x, y, p, sensor_size=(H, W), clip_out_of_range=clip, interpolation=interp,
                                  padding=padding, default=default, accum="f64")
    got = E.events_to_image_torch(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(p), sensor_size=(H, W),
                                  clip_out_of_range=clip, interpolation=interp, padding=padding, default=default)
    assert tuple(got.shape) == ref.shape
    # float32 accumulation (as the reference's index_put_): the clipped events pile up on one pixel, thousands of N(0, 1)
    # weights whose sum cancels -- its rounding error scales with the accumulated magnitude, not with the cancelled sum
    mag = R.events_to_image_torch(x, y, np.abs(p), sensor_size=(H, W), clip_out_of_range=clip, interpolation=interp,
                                  padding=padding, default=0.0, accum="f64")
    tol = 1e-5 * max(np.max(np.abs(ref)), 1e-30) + 2e-7 * np.max(np.abs(mag))
    assert np.max(np.abs(got.numpy().astype(np.float64) - ref)) <= tol
    # numpy entry point: integer coordinates inside the (H+1, W+1) canvas, nearest, with and without meanval
    xi, yi = rng.integers(0, W + 1, n), rng.integers(0, H + 1, n)
    pi = rng.integers(-3, 4, n)
    for meanval in (False, True):
        a = E.events_to_image(xi, yi, pi, sensor_size=(H, W), meanval=meanval, default=default)
        b = R.events_to_image(xi, yi, pi, sensor_size=(H, W), meanval=meanval, default=default)
        assert a.dtype == np.float64 and np.array_equal(a, b)


@pytest.mark.parametrize("seed", range(12))
def test_random_get_iwe_variants_equal_the_oracle(seed):
    """get_iwe over its option space: polarity on / off, gradient on / off, the hard-wired (181, 241) canvas (Q1) or an
    explicit sensor_size, img_size different from the canvas, flows that push events out of bounds (Q2, Q3), float64
    inputs that are not float32-representable (float64 device columns), both kernel families."""
    import event_utils_amd as E
    from oracle import reference_np as R
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([40, 3000, 200_000]))
    sensor = None if seed % 3 == 0 else (int(rng.integers(20, 200)), int(rng.integers(20, 260)))
    canvas = (180, 240) if sensor is None else sensor
    img_size = canvas if seed % 2 else (canvas[0] + int(rng.integers(-10, 30)), canvas[1] + int(rng.integers(-10, 30)))
    x = rng.uniform(-2, canvas[1] + 2, n); y = rng.uniform(-2, canvas[0] + 2, n)
    t = np.sort(rng.uniform(0, 0.2, n)); p = rng.choice([-1.0, 1.0], n)
    if seed % 4 != 3:       # float32-representable columns -> float32 device columns; else float64 ones
        x, y, t = (a.astype(np.float32).astype(np.float64) for a in (x, y, t))
        t = np.sort(t)
    prm = rng.uniform(-400, 400, 2)
    grad, pol = bool((seed >> 1) & 1), bool(seed & 1) or seed % 5 == 0
    ri, rd = R.get_iwe(prm, x, y, t, p, R.linvel_warp(), img_size, compute_gradient=grad, use_polarity=pol,
                       sensor_size=sensor, accum="f64")
    for impl in ("direct", "tiled"):
        os.environ["EVK_IMPL"] = impl
        try:
            iwe, diwe = E.get_iwe(prm, x, y, t, p, E.linvel_warp(), img_size, compute_gradient=grad, use_polarity=pol,
                                  sensor_size=sensor)
        finally:
            os.environ.pop("EVK_IMPL", None)
        assert iwe.shape == ri.shape and iwe.dtype == np.float32
        assert np.max(np.abs(iwe.astype(np.float64) - ri)) <= 1e-5 * max(np.max(np.abs(ri)), 1e-30), impl
        if grad:
            assert np.max(np.abs(diwe.astype(np.float64) - rd)) <= 1e-5 * max(np.max(np.abs(rd)), 1e-30), impl
        else:
            assert diwe is None


def test_optimize_at_configs3_size_converges_and_matches_the_direct_kernels():
    """configs[3]: 50 M events, 1280x720, the full optimize() loop -- both gradient modes reach the scene's true flow
    (40, -25) px/s, and at that size the reference-exact function value / gradient of the tiled path equal the direct
    (global-atomic) kernels'."""
    import warnings
    import bench
    import event_utils_amd as E
    from event_utils_amd.contrast_max.events_cmax import optimize_contrast
    from event_utils_amd.events import DeviceEvents
    H, W, n = 720, 1280, 50_000_000
    x, y, t, p = bench.structured_scene(3, n, H, W)
    ev = DeviceEvents.from_arrays(x, y, t, p, precision="f32")
    del x, y, t, p
    w = E.linvel_warp()
    for numeric, exact in ((True, True), (False, False)):      # the reference's default path; the consistent analytic gradient
        obj = E.variance_objective()
        obj.sensor_size, obj.reference_exact = (H, W), exact
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            argmax = optimize_contrast(ev, None, None, None, w, obj, numeric_grads=numeric, blur_sigma=1.0, img_size=(H, W))
        assert np.abs(np.asarray(argmax, dtype=float) - np.array([40.0, -25.0])).max() <= 0.7, argmax
    prm = np.array([38.0, -23.5])
    vals = {}
    for impl in ("tiled", "direct"):
        obj = E.variance_objective()
        obj.sensor_size, obj.impl = (H, W), impl
        vals[impl] = (float(obj.evaluate_function(prm, ev, None, None, None, w, (H, W), 1.0)),
                      np.asarray(obj.evaluate_gradient(prm, ev, None, None, None, w, (H, W), 1.0), dtype=np.float64))
    assert abs(vals["tiled"][0] - vals["direct"][0]) <= 1e-5 * abs(vals["direct"][0])
    assert np.abs(vals["tiled"][1] - vals["direct"][1]).max() <= 1e-5 * np.abs(vals["direct"][1]).max()


# ---------------------------------------------------------------------------------------------------------------------
# Full-size comparisons with the oracle (round 4): BASELINE.json's sizes on the DEFAULT dispatch (the code paths that are
# switched by size -- 4-byte voxel records above 16 M events, compact IWE records beyond the Infinity Cache, the tile
# kernels' workgroup shapes -- run as a user gets them), against oracle/reference_np.py with float64 accumulation.
# Bar: 1e-5 of the reference's maximum.  The oracle needs 0.5-12 s per case on one host core.
# ---------------------------------------------------------------------------------------------------------------------
def _rel(a, ref):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return np.max(np.abs(a - ref)) / max(np.max(np.abs(ref)), 1e-30)


def test_full_size_voxel_configs1_against_the_oracle():
    """configs[1]: 10 M events, 640x480, 5 bins."""
    import event_utils_amd as E
    from oracle import reference_np as R
    rng = np.random.default_rng(1)
    H, W, B, n = 480, 640, 5, 10_000_000
    x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32); p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    got = E.events_to_voxel_torch(*(torch.from_numpy(a).cuda() for a in (x, y, t, p)), B, sensor_size=(H, W)).cpu().numpy()
    ref = R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64")
    assert _rel(got, ref) <= 1e-5
    # weights that are not +-1: the float64 two-atomic path of the tile kernel
    pw = (p * rng.uniform(0.1, 3.0, n)).astype(np.float32)
    got = E.events_to_voxel_torch(*(torch.from_numpy(a).cuda() for a in (x, y, t, pw)), B, sensor_size=(H, W)).cpu().numpy()
    assert _rel(got, R.events_to_voxel_torch(x, y, t, pw, B, sensor_size=(H, W), accum="f64")) <= 1e-5


def test_full_size_voxel_configs4_share_against_the_oracle():
    """One rank's share of configs[4]: 50 M events, 1280x720, 5 bins (4-byte records, HBM-resident)."""
    import event_utils_amd as E
    from oracle import reference_np as R
    rng = np.random.default_rng(9)
    H, W, B, n = 720, 1280, 5, 50_000_000
    x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32); p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    got = E.events_to_voxel_torch(*(torch.from_numpy(a).cuda() for a in (x, y, t, p)), B, sensor_size=(H, W)).cpu().numpy()
    ref = R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64")
    assert _rel(got, ref) <= 1e-5


@pytest.mark.parametrize("cfg", [("configs[2]", 10_000_000, 480, 640, "uniform"), ("configs[3]", 50_000_000, 720, 1280, "edges")])
def test_full_size_iwe_and_objective_against_the_oracle(cfg):
    """configs[2] (10 M events, 640x480): IWE, dIWE, objective and gradient; configs[3] (50 M events, 1280x720, the
    moving-edge scene of the optimize() benchmark): IWE + dIWE, objective and gradient at one flow."""
    import bench
    import event_utils_amd as E
    from oracle import reference_np as R
    from event_utils_amd.contrast_max.objectives import iwe_device
    _, n, H, W, scene = cfg
    if scene == "uniform":
        rng = np.random.default_rng(2)
        x = rng.uniform(1, W - 1, n).astype(np.float32); y = rng.uniform(1, H - 1, n).astype(np.float32)
        t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32); p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    else:
        x, y, t, p = bench.structured_scene(3, n, H, W)
    prm = np.array([30., -20.])
    ev = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
    iwe, diwe = iwe_device(prm, ev, (H, W), True, True, (H, W))
    d = [a.astype(np.float64) for a in (x, y, t, p)]
    ri, rd = R.get_iwe(prm, *d, R.linvel_warp(), (H, W), compute_gradient=True, sensor_size=(H, W), accum="f64")
    assert _rel(iwe.cpu().numpy(), ri) <= 1e-5
    assert _rel(diwe.cpu().numpy(), rd) <= 1e-5
    obj, w = E.variance_objective(), E.linvel_warp()
    obj.sensor_size = (H, W)
    robj = R.variance_objective(); robj.sensor_size = (H, W); robj.accum = "f64"
    f = float(obj.evaluate_function(prm, ev, None, None, None, w, (H, W), 1.0))
    g = np.asarray(obj.evaluate_gradient(prm, ev, None, None, None, w, (H, W), 1.0), np.float64)
    fr = float(robj.evaluate_function(prm, *d, R.linvel_warp(), (H, W), 1.0, iwe=ri))
    gr = np.asarray(robj.evaluate_gradient(prm, *d, R.linvel_warp(), (H, W), 1.0, iwe=ri, d_iwe=rd), np.float64)
    assert abs(f - fr) <= 1e-5 * abs(fr), (f, fr)
    assert np.max(np.abs(g - gr)) <= 1e-5 * np.max(np.abs(gr)) + 1e-9, (g, gr)
