"""GPU (-m gpu): the parameter-space samplers of events_cmax.py (grid search, objective landscape, grid_cmax,
optimize_r2), the batched many-flows evaluation behind them, and rms_objective -- against the golden vectors the real
reference produced (tests/golden/f14_search.npz) and against single evaluations of the same HIP path."""
import numpy as np
import pytest
import torch

from oracle import reference_np as R

pytestmark = pytest.mark.gpu
TOL = 1e-5


def f64(a):
    return np.asarray(a, dtype=np.float64)


@pytest.fixture(scope="module")
def E():
    import event_utils_amd as E
    from event_utils_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available()
    return E


@pytest.fixture(scope="module")
def C():
    from event_utils_amd.contrast_max import events_cmax
    return events_cmax


@pytest.fixture(scope="module")
def scene(golden):
    g8 = golden("f8_objective")
    return f64(g8["xs"]), f64(g8["ys"]), f64(g8["ts"]), f64(g8["ps"]), tuple(int(v) for v in g8["img_size"])


def test_grid_search_initial_matches_reference(E, C, golden, scene):
    g = golden("f14_search")
    x, y, t, p, img_size = scene
    for tag, kw in (("log5", dict(log_scale=True, num_samples_per_param=5)),
                    ("lin7", dict(log_scale=False, num_samples_per_param=7, param_ranges=[[-60, 60], [-90, 30]]))):
        r = C.grid_search_initial(x, y, t, p, E.linvel_warp(), E.variance_objective(), img_size, **kw)
        ref = g["gsi_%s_eval" % tag]
        assert np.array_equal(np.array(r["params"]), g["gsi_%s_params" % tag])
        assert np.max(np.abs(f64(r["eval"]) - ref)) <= TOL * np.max(np.abs(ref))
        assert np.array_equal(np.array(r["min_params"]), g["gsi_%s_min_params" % tag])
        assert abs(r["min_func_eval"] - g["gsi_%s_min_eval" % tag]) <= TOL * abs(g["gsi_%s_min_eval" % tag])


def test_grid_search_optimisation_follows_the_reference_levels(E, C, golden, scene):
    g = golden("f14_search")
    x, y, t, p, img_size = scene
    levels = []
    gsi = C.grid_search_initial

    def recording(*a, **k):
        r = gsi(*a, **k)
        levels.append(np.array(r["min_params"], dtype=np.float64))
        return r
    C.grid_search_initial = recording
    try:
        r = C.grid_search_optimisation(x, y, t, p, E.linvel_warp(), E.variance_objective(), img_size, log_scale=False)
    finally:
        C.grid_search_initial = gsi
    assert len(levels) == len(g["gso_level_min_params"])
    assert np.array_equal(np.array(levels), g["gso_level_min_params"])      # same best sample at every level
    assert np.array_equal(np.array(r["min_params"]), g["gso_min_params"])
    assert abs(r["min_func_eval"] - g["gso_min_eval"]) <= TOL * abs(g["gso_min_eval"])


def test_objective_landscape_and_draw(E, C, golden, scene):
    import matplotlib
    matplotlib.use("Agg")
    g = golden("f14_search")
    x, y, t, p, img_size = scene
    a = g["landscape_args"]
    kw = dict(x_range=(a[0], a[1]), y_range=(a[2], a[3]), resolution=a[4], img_size=img_size)
    img = C.objective_landscape(x, y, t, p, E.variance_objective(minimum_events=1), E.linvel_warp(), **kw)
    assert img.shape == g["landscape"].shape
    assert np.max(np.abs(img - g["landscape"])) <= TOL              # normalised to [0, 1]
    img2 = C.draw_objective_function(x, y, t, p, show=False, gt=(40, -25), **kw)
    assert np.max(np.abs(img - img2)) <= TOL      # float atomics: two runs agree to rounding, not bitwise
    import matplotlib.pyplot as plt
    plt.close("all")


@pytest.mark.parametrize("n,shape", [(400_000, (480, 640)), (30_000, (180, 240))])
def test_batched_evaluation_equals_single_evaluations(E, monkeypatch, n, shape):
    """K flows through evaluate_function_batch (three flows per pass on the tiled path, each with its own LDS window origin;
    flows too large for the windows and the direct-kernel regime fall back to single passes) == K separate
    evaluate_function calls."""
    H, W = shape
    rng = np.random.default_rng(5)
    x = rng.uniform(1, W - 1, n).astype(np.float32); y = rng.uniform(1, H - 1, n).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32); p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    ev = E.DeviceEvents.from_arrays(x, y, t, p)
    obj = E.variance_objective()
    obj.sensor_size = (H, W)
    flows = [(0., 0.), (20., 0.), (40., 0.), (40., -25.), (41., -25.), (40., -24.), (-300., 200.), (-300., 230.),
             (900., 900.), (30., -20.), (35., 10.)]                       # K = 11: near trios, far trios, a ragged tail
    w = E.linvel_warp()
    single = [obj.evaluate_function(np.array(q), ev, None, None, None, w, (H, W), 1.0) for q in flows]
    launches = []
    from event_utils_amd import tiled
    real = tiled.cmax_variance_batch3

    def counting(*a, **k):
        ok = real(*a, **k)
        launches.append(ok)
        return ok
    monkeypatch.setattr(tiled, "cmax_variance_batch3", counting)
    batch = obj.evaluate_function_batch(flows, ev, None, None, None, w, (H, W), 1.0)
    assert len(batch) == len(flows)
    assert np.max(np.abs(f64(batch) - f64(single))) <= 2e-6 * np.max(np.abs(f64(single)))
    if n >= 150_000:
        assert sum(launches) >= 2            # the near trios really shared a pass
    ref = R.variance_objective(); ref.sensor_size = (H, W); ref.accum = "f64"
    for k in (3, 6, 10):
        r = ref.evaluate_function(np.array(flows[k]), f64(x), f64(y), f64(t), f64(p), R.linvel_warp(), (H, W), 1.0)
        assert abs(batch[k] - r) <= TOL * abs(r)


def test_grid_search_initial_at_configs2_size_takes_nine_event_passes(E, C):
    """configs[2]'s events (10 M, 640x480): the 25 samples of one grid-search level over the reference's default range
    (+-150 px/s: flows up to 30 px apart) are 9 passes over the events -- every flow of a three-flow pass has its own LDS
    window origin (round 4; with shared windows only flows within 6 px shared a pass and this level took 25) -- and equal
    the single evaluations; three of them are checked against the oracle."""
    rng = np.random.default_rng(2)
    n, H, W = 10_000_000, 480, 640
    x = rng.uniform(1, W - 1, n).astype(np.float32); y = rng.uniform(1, H - 1, n).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32); p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    ev = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
    obj, w = E.variance_objective(), E.linvel_warp()
    obj.sensor_size = (H, W)
    calls = []
    real = obj.evaluate_function_batch

    def counting(*a, **k):
        r = real(*a, **k)
        calls.append(obj.batch_passes)
        return r
    obj.evaluate_function_batch = counting
    r = C.grid_search_initial(ev, None, None, None, w, obj, (H, W))
    assert len(r["params"]) == 25 and calls == [9]
    single = [float(obj.evaluate_function(np.array(q), ev, None, None, None, w, (H, W), 1.0)) for q in r["params"]]
    assert np.max(np.abs(f64(r["eval"]) - f64(single))) <= 2e-6 * np.max(np.abs(f64(single)))
    ref = R.variance_objective(); ref.sensor_size = (H, W); ref.accum = "f64"
    d = [f64(a) for a in (x, y, t, p)]
    for k in (0, 12, 23):
        fr = float(ref.evaluate_function(np.array(r["params"][k]), *d, R.linvel_warp(), (H, W), 1.0))
        assert abs(r["eval"][k] - fr) <= TOL * abs(fr)


def test_rms_objective_matches_reference(E, golden, scene):
    g = golden("f14_search")
    x, y, t, p, img_size = scene
    from event_utils_amd.contrast_max.objectives import rms_objective
    rms, w = rms_objective(), E.linvel_warp()
    assert (rms.name, rms.use_polarity, rms.has_derivative, rms.default_blur) == ("rms", True, True, 1.0)
    k = 0
    for q in g["rms_params"]:
        for s in (None, 0.0):
            f = rms.evaluate_function(q, x, y, t, p, w, img_size, blur_sigma=s)
            assert abs(f - g["rms_f"][k]) <= TOL * abs(g["rms_f"][k])
            gr = rms.evaluate_gradient(q, x, y, t, p, w, img_size, blur_sigma=s)
            assert np.max(np.abs(f64(gr) - g["rms_g"][k])) <= TOL * max(np.max(np.abs(g["rms_g"][k])), 1e-3)
            k += 1


def test_grid_search_initialised_bfgs_and_r2(E, C, scene):
    x, y, t, p, img_size = scene
    w = E.linvel_warp()
    a = C.optimize_contrast(x, y, t, p, w, E.variance_objective(), numeric_grads=False, blur_sigma=1.0,
                            img_size=img_size, grid_search_init=True)
    assert np.linalg.norm(np.asarray(a) - np.array([40., -25.])) < 1.0
    ev = E.DeviceEvents.from_arrays(x, y, t, p)
    r2 = C.optimize_r2(ev, None, None, None, w, E.variance_objective(), numeric_grads=False)
    first = C.optimize_contrast(ev, None, None, None, w, E.variance_objective(), numeric_grads=False, blur_sigma=None)
    from event_utils_amd.contrast_max.objectives import soe_objective
    manual = C.optimize_contrast(ev, None, None, None, w, soe_objective(), x0=first, numeric_grads=False, blur_sigma=1.0)
    assert np.allclose(r2, manual, rtol=0, atol=0.05)     # float-atomic rounding differs run to run; BFGS amplifies it
    assert np.linalg.norm(np.asarray(r2) - np.array([40., -25.])) < 3.0


def test_grid_cmax_cells(E, C, scene):
    """Two halves of the sensor moving with different flows: every cell recovers its own."""
    rng = np.random.default_rng(11)
    n, H, W = 40_000, 64, 128
    t = np.sort(rng.uniform(0, 0.2, n))
    left = rng.random(n) < 0.5
    flow = np.where(left[:, None], np.array([[60., 0.]]), np.array([[-40., 30.]]))
    x0 = np.where(left, rng.choice(np.arange(12, 52, 8), n), rng.choice(np.arange(76, 116, 8), n)) + rng.normal(0, .3, n)
    y0 = rng.uniform(14, H - 14, n)
    horiz = rng.random(n) < 0.5                      # half of the events sit on horizontal edges instead
    y0 = np.where(horiz, rng.choice(np.arange(16, 48, 8), n) + rng.normal(0, .3, n), y0)
    x0 = np.where(horiz, np.where(left, rng.uniform(10, 54, n), rng.uniform(74, 118, n)), x0)
    x = np.rint(x0 + (t - t[-1]) * flow[:, 0]); y = np.rint(y0 + (t - t[-1]) * flow[:, 1])
    keep = (x >= 0) & (x < W) & (y >= 0) & (y < H)
    x, y, t = x[keep], y[keep], t[keep]
    p = np.where(horiz[keep], -1.0, 1.0)
    x[-1], y[-1] = W - 1, H - 1                      # pins the inferred resolution
    params, rois, fevals = C.grid_cmax(x, y, t, p, roi_size=(H, W // 2))
    assert rois == [[0, 0, H, W // 2], [0, W // 2, H, W // 2]] and len(params) == len(fevals) == 2
    assert np.linalg.norm(params[0] - np.array([60., 0.])) < 6.0
    assert np.linalg.norm(params[1] - np.array([-40., 30.])) < 6.0
    assert all(f < 0 for f in fevals)


@pytest.mark.parametrize("shape", [(181, 241), (241, 181), (481, 641), (721, 1281), (7, 300), (64, 64)])
def test_spectral_norm_kernel_against_numpy(E, shape):
    """evk_spectral_norm_sq_f32 (Lanczos on the Gram operator, one workgroup, float64): sigma_max^2 of random, smooth,
    sign-alternating, rank-one, rank-two and zero images against np.linalg.norm(a.astype(float64), 2)^2."""
    from event_utils_amd import _device as D, _lib, tiled
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    h, w = shape
    yy, xx = np.mgrid[0:h, 0:w]
    cases = {
        "noise": rng.normal(size=shape),
        "positive": rng.poisson(3.0, size=shape).astype(np.float64),
        "smooth": np.sin(xx / 17.0) * np.cos(yy / 11.0) + 0.3 * rng.normal(size=shape),
        "checker": ((xx + yy) % 2 * 2.0 - 1.0) * (1.0 + 0.01 * rng.normal(size=shape)),
        "rank1": np.outer(rng.normal(size=h), rng.normal(size=w)),
        "rank2": np.outer(rng.normal(size=h), rng.normal(size=w)) + 0.999 * np.outer(rng.normal(size=h), rng.normal(size=w)),
        "one pixel": np.where((yy == h // 2) & (xx == w // 3), 5.0, 0.0),
        "zero": np.zeros(shape),
    }
    if min(shape) >= 181:
        # a SLOWLY converging spectrum (round 6): 150 singular values within 3e-3 of the largest -- one sweep of 96 Lanczos steps
        # stops short of 1e-9 here; the kernel measures the Ritz pair's true residual and restarts from the Ritz vector
        k = min(shape)
        sv = np.concatenate([1.0 - 2e-5 * np.arange(150), rng.uniform(0.0, 0.5, k - 150)])
        qa, _ = np.linalg.qr(rng.normal(size=(h, k)))
        qb, _ = np.linalg.qr(rng.normal(size=(w, k)))
        cases["clustered top"] = (qa * sv) @ qb.T
    dev = torch.device("cuda", 0)
    nbytes = int(_lib.lib().evk_spectral_scratch_bytes(h, w))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out = torch.zeros(4, dtype=torch.float64, device=dev)
    for name, a in cases.items():
        a32 = np.ascontiguousarray(a, dtype=np.float32)
        img = torch.from_numpy(a32).cuda()
        _lib.call("evk_spectral_norm_sq_f32", D.ptr(img), h, w, D.ptr(out), D.ptr(scratch), nbytes, D.stream())
        got, resid = (float(v) for v in out[:2].cpu().numpy())
        want = float(np.linalg.norm(a32.astype(np.float64), 2)) ** 2
        assert got <= want * (1.0 + 1e-12), (name, shape, got, want)                # a Ritz value never exceeds the truth
        assert abs(got - want) <= max(1e-9, 1.01 * resid) * max(want, 1e-300), (name, shape, got, want, resid)
        if name != "clustered top":
            assert resid <= 1e-9 and abs(got - want) <= 1e-9 * max(want, 1e-300), (name, shape, got, want, resid)
        else:
            assert abs(got - want) <= 2e-6 * want, (name, shape, got, want, resid)  # (the reported bound is what a caller checks)
