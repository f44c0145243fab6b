"""CPU: the numpy restatement (oracle/reference_np.py) against the golden vectors produced by the real
reference (oracle/make_golden.py).  Integer and f32-sequential paths must be bit-exact."""
import numpy as np
import pytest

from oracle import reference_np as R


def f64(a):
    return np.asarray(a, dtype=np.float64)


def test_f1_image_nearest_int_bit_exact(golden):
    g = golden("f1_image_nearest_int")
    xs, ys, ps = g["xs"].astype(np.int64), g["ys"].astype(np.int64), g["ps"].astype(np.int64)
    ss = tuple(g["sensor_size"])
    assert np.array_equal(R.events_to_image(xs, ys, ps, sensor_size=ss), g["img_pm"])
    assert np.array_equal(R.events_to_image(xs, ys, np.ones_like(ps), sensor_size=ss), g["img_cnt"])
    assert np.array_equal(R.events_to_image(xs, ys, ps, sensor_size=ss, meanval=True), g["img_mean"])
    assert np.array_equal(R.events_to_image(xs, ys, ps, sensor_size=ss, meanval=True, default=7), g["img_mean_default"])
    assert np.array_equal(R.events_to_image(xs, ys, g["wf"], sensor_size=ss), g["img_wf"])
    assert R.events_to_image(xs, ys, ps, sensor_size=ss).dtype == np.float64


def test_f1_errors():
    xs = np.array([0, 241]); ys = np.array([0, 0]); ps = np.array([1, 1])
    with pytest.raises(ValueError):
        R.events_to_image(xs, ys, ps)
    with pytest.raises(TypeError):
        R.events_to_image(xs.astype(float), ys.astype(float), ps)


@pytest.mark.parametrize("tag,Bs", [("small", (1, 2, 5, 9)), ("dvs", (5,))])
def test_f2_voxel_numpy(golden, tag, Bs):
    g = golden("f2_voxel_numpy")
    xs, ys = g[tag + "_xs"].astype(np.int64), g[tag + "_ys"].astype(np.int64)
    ts, ps = g[tag + "_ts"], f64(g[tag + "_ps"])
    for B in Bs:
        v = R.events_to_voxel(xs, ys, ts, ps, B, sensor_size=tuple(g[tag + "_sensor_size"]))
        ref = g["%s_voxel_B%d" % (tag, B)]
        assert v.dtype == np.float64 and v.shape == ref.shape
        assert np.array_equal(v, ref)


@pytest.mark.parametrize("tag,Bs", [("small", (1, 2, 5, 9)), ("dvs", (5,))])
def test_f3_voxel_torch(golden, tag, Bs):
    g = golden("f3_voxel_torch")
    xs, ys, ts = g[tag + "_xs"], g[tag + "_ys"], g[tag + "_ts"]
    ps = g[tag + "_ps"].astype(np.float32)
    ss = tuple(g[tag + "_sensor_size"])
    for B in Bs:
        ref = g["%s_voxel_B%d" % (tag, B)]
        v = R.events_to_voxel_torch(xs, ys, ts, ps, B, sensor_size=ss)
        assert v.dtype == np.float32
        assert np.array_equal(v, ref)
        v64 = R.events_to_voxel_torch(xs, ys, ts, ps, B, sensor_size=ss, accum="f64")
        assert np.max(np.abs(f64(v64) - f64(ref))) <= 1e-5 * np.max(np.abs(ref))
    if tag == "dvs":
        v = R.events_to_voxel_torch(xs.astype(np.int64), ys.astype(np.int64), ts, ps, 5, sensor_size=ss)
        assert np.array_equal(v, g["dvs_voxel_B5_long"])


def test_f4_image_torch(golden):
    g = golden("f4_image_torch")
    xs, ys, ps = g["xs"], g["ys"], g["ps"]
    ss = tuple(g["sensor_size"])
    assert np.array_equal(R.events_to_image_torch(xs, ys, ps, sensor_size=ss, interpolation='bilinear', padding=True), g["bil_pad"])
    assert np.array_equal(R.events_to_image_torch(xs, ys, ps, sensor_size=ss, interpolation='bilinear', padding=False), g["bil_nopad"])
    assert np.array_equal(R.events_to_image_torch(xs, ys, ps, sensor_size=ss, interpolation=None, padding=True), g["near_pad"])
    assert np.array_equal(R.events_to_image_torch(xs, ys, ps, sensor_size=ss, interpolation=None, padding=False), g["near_nopad"])
    assert np.array_equal(R.events_to_image_torch(xs, ys, ps, sensor_size=ss, interpolation=None, padding=False, default=3), g["near_default3"])
    assert np.array_equal(R.events_to_image(f64(xs), f64(ys), f64(ps), sensor_size=ss, interpolation='bilinear', padding=False), g["np_bil"])
    assert np.array_equal(R.events_to_image(f64(xs), f64(ys), f64(ps), sensor_size=ss, interpolation='bilinear', padding=True), g["np_bil_pad"])


def test_f5_warp(golden):
    g = golden("f5_warp")
    x, y, t, p = f64(g["xs"]), f64(g["ys"]), f64(g["ts"]), f64(g["ps"])
    w = R.linvel_warp()
    assert (w.name, w.dims) == ("linvel_warp", 2)
    for i, prm in enumerate(g["params"]):
        xp, yp, jx, jy = w.warp(x, y, t, p, t[-1], prm, compute_grad=True)
        for a, k in ((xp, "xp"), (yp, "yp"), (jx, "jx"), (jy, "jy")):
            assert np.array_equal(a, g["%s%d" % (k, i)])
        assert np.array_equal(R.events_bounds_mask(xp, yp, 0, 240, 0, 180), g["mask%d" % i])
    xp, yp, jx, jy = w.warp(x, y, t, p, t[-1], g["params"][1])
    assert jx is None and jy is None


def test_f6_get_iwe_verbatim(golden):
    g = golden("f6_get_iwe")
    x, y, t, p = f64(g["xs"]), f64(g["ys"]), f64(g["ts"]), f64(g["ps"])
    w = R.linvel_warp()
    img_size = tuple(g["img_size"])
    for i, prm in enumerate(g["params"]):
        iwe, diwe = R.get_iwe(prm, x, y, t, p, w, img_size, compute_gradient=True)
        assert iwe.shape == (181, 241) and diwe.shape == (2, 181, 241) and iwe.dtype == np.float32
        assert np.array_equal(iwe, g["iwe%d" % i])
        assert np.array_equal(diwe, g["diwe%d" % i])
    iwe, d = R.get_iwe(g["params"][0], x, y, t, p, w, img_size, compute_gradient=False, use_polarity=False)
    assert d is None and np.array_equal(iwe, g["iwe_nopol0"])
    # Q1: img_size only sets the bounds mask, the canvas stays (181, 241)
    x, y, t, p = f64(g["q1_xs"]), f64(g["q1_ys"]), f64(g["q1_ts"]), f64(g["q1_ps"])
    iwe, diwe = R.get_iwe(np.array([30., -20.]), x, y, t, p, w, tuple(g["q1_img_size"]), compute_gradient=True)
    assert np.array_equal(iwe, g["q1_iwe"]) and np.array_equal(diwe, g["q1_diwe"])


@pytest.mark.parametrize("tag", ["s48", "vga"])
def test_f7_iwe_sized(golden, tag):
    g = golden("f7_iwe_sized")
    x, y, t, p = f64(g[tag + "_xs"]), f64(g[tag + "_ys"]), f64(g[tag + "_ts"]), f64(g[tag + "_ps"])
    ss = tuple(g[tag + "_sensor_size"])
    iwe, diwe = R.get_iwe(g["params"], x, y, t, p, R.linvel_warp(), ss, compute_gradient=True, sensor_size=ss)
    assert np.array_equal(iwe, g[tag + "_iwe"]) and np.array_equal(diwe, g[tag + "_diwe"])
    iwe64, diwe64 = R.get_iwe(g["params"], x, y, t, p, R.linvel_warp(), ss, compute_gradient=True, sensor_size=ss, accum="f64")
    assert np.max(np.abs(f64(iwe64) - f64(iwe))) <= 1e-5 * np.max(np.abs(iwe))
    assert np.max(np.abs(f64(diwe64) - f64(diwe))) <= 1e-5 * np.max(np.abs(diwe))


def test_f10_blur(golden):
    from scipy.ndimage import gaussian_filter
    g = golden("f10_blur")
    a3 = g["a3"]
    for s in (1.0, 2.0, 0.5):
        assert np.array_equal(R.gaussian_filter_reflect(a3, s), g["blur3_s%g" % s])
        assert np.array_equal(R.gaussian_filter_reflect(a3[0], s), g["blur2_s%g" % s])
        assert np.array_equal(R.gaussian_filter_reflect(a3, s), gaussian_filter(a3, s))   # scipy on this box
    assert np.array_equal(R.gaussian_filter_reflect(g["small"], 1.0), g["small_blur_s1"])
    # Q4: channel mixing of the 2-channel axis at sigma=1
    z = np.zeros((2, 21, 21), dtype=np.float32); z[0, 10, 10] = 1
    b = R.gaussian_filter_reflect(z, 1.0)
    assert abs(b[0].sum() - 0.64561) < 1e-4 and abs(b[1].sum() - 0.35439) < 1e-4


def test_f8_objective(golden):
    g = golden("f8_objective")
    x, y, t, p = f64(g["xs"]), f64(g["ys"]), f64(g["ts"]), f64(g["ps"])
    w, obj = R.linvel_warp(), R.variance_objective()
    img_size = tuple(g["img_size"])
    for i, prm in enumerate(g["params"]):
        for j, s in enumerate(g["sigmas"]):
            f = obj.evaluate_function(prm, x, y, t, p, w, img_size, blur_sigma=s)
            gr = obj.evaluate_gradient(prm, x, y, t, p, w, img_size, blur_sigma=s)
            assert np.float64(f) == g["f"][i, j]
            assert np.array_equal(f64(gr), g["grad"][i, j])
    al = R.variance_objective(adaptive_lifespan=True, minimum_events=5000)
    al.iter_update(np.array([400., -250.]))
    f = al.evaluate_function(np.array([40., -25.]), x, y, t, p, w, img_size, blur_sigma=1.0)
    assert al.s_idx == g["al_s_idx"] and np.float64(f) == g["al_f"]
    al.iter_update(np.array([400., -250.]))
    assert np.array_equal(f64(al.evaluate_gradient(np.array([40., -25.]), x, y, t, p, w, img_size, blur_sigma=1.0)), g["al_grad"])


@pytest.mark.parametrize("mode", ["numeric", "analytic"])
def test_f9_optimize_trace(golden, mode):
    g8, g = golden("f8_objective"), golden("f9_optimize_trace")
    x, y, t, p = f64(g8["xs"]), f64(g8["ys"]), f64(g8["ts"]), f64(g8["ps"])
    obj = R.variance_objective()
    trace = []
    f0, g0 = obj.evaluate_function, obj.evaluate_gradient
    obj.evaluate_function = lambda prm, *a, **k: (trace.append(("f", np.array(prm, float))), f0(prm, *a, **k))[1]
    obj.evaluate_gradient = lambda prm, *a, **k: (trace.append(("g", np.array(prm, float))), g0(prm, *a, **k))[1]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        argmax = R.optimize_contrast(x, y, t, p, R.linvel_warp(), obj, numeric_grads=(mode == "numeric"),
                                     blur_sigma=1.0, img_size=tuple(g8["img_size"]))
    assert np.allclose(argmax, g[mode + "_argmax"], rtol=0, atol=1e-9)
    assert [k for k, _ in trace] == list(g[mode + "_kind"])
    assert np.allclose(np.array([q for _, q in trace]), g[mode + "_params"], rtol=0, atol=1e-9)


def test_f11_gather_contrast_timestamp_images(golden):
    g = golden("f11_gather_timestamp")
    x, y, t, p = f64(g["xs"]), f64(g["ys"]), f64(g["ts"]), f64(g["ps"])
    r = R.get_iwe(g["params"], x, y, t, p, R.linvel_warp(), (180, 240), return_events=True, return_per_event_contrast=True)
    assert np.array_equal(r[0], g["iwe"]) and r[1] is None
    assert np.array_equal(r[2][0], g["ev_x"]) and np.array_equal(r[2][1], g["ev_y"])
    assert np.array_equal(r[3], g["contrast"])
    assert np.array_equal(R.image_to_event_weights(g["g_x"], g["g_y"], g["g_img"]), g["g_w"])
    xi, yi = f64(g["ti_x"]), f64(g["ti_y"])
    a, b = R.events_to_timestamp_image(xi, yi, g["ti_ts64"], p)
    assert a.dtype == np.float32 and np.array_equal(a, g["ti_np_pos"]) and np.array_equal(b, g["ti_np_neg"])
    a, b = R.events_to_timestamp_image(xi, yi, g["ti_ts64"], p, padding=False, normalize_timestamps=False)
    assert np.array_equal(a, g["ti_np_nopad_pos"]) and np.array_equal(b, g["ti_np_nopad_neg"])
    for rev in (False, True):
        a, b = R.events_to_timestamp_image_torch(g["ti_x"], g["ti_y"], g["ti_ts64"].astype(np.float32), p.astype(np.float32),
                                                 timestamp_reverse=rev)
        assert np.array_equal(a, g["ti_t_pos_rev%d" % rev]) and np.array_equal(b, g["ti_t_neg_rev%d" % rev])


def test_f12_other_objectives(golden):
    g8, g = golden("f8_objective"), golden("f12_other_objectives")
    x, y, t, p = f64(g8["xs"]), f64(g8["ys"]), f64(g8["ts"]), f64(g8["ps"])
    w = R.linvel_warp()
    objs = {"sos": R.sos_objective(), "soe": R.soe_objective(), "moa": R.moa_objective(), "isoa": R.isoa_objective(),
            "sosa": R.sosa_objective()}
    for name, ob in objs.items():
        k = 0
        for prm in g["params"]:
            for s in (None, 0.0):
                assert np.float64(ob.evaluate_function(prm, x, y, t, p, w, (180, 240), blur_sigma=s)) == g[name + "_f"][k], name
                if ob.has_derivative:
                    assert np.array_equal(f64(ob.evaluate_gradient(prm, x, y, t, p, w, (180, 240), blur_sigma=s)), g[name + "_g"][k]), name
                k += 1
    r1 = R.r1_objective()
    P = g["params"]
    vals = [np.float64(r1.evaluate_function(q, x, y, t, p, w, (180, 240))) for q in (P[0], P[1], P[1], P[2])]
    assert np.array_equal(np.array(vals), g["r1_f"])


def test_f13_dense_flow_warp(golden):
    """torch's vectorised grid_sample contracts multiplies and adds differently from a plain float32 evaluation, so this
    row is pinned to 1 ulp of the coordinates (2e-7 relative), not bit-exactly."""
    g = golden("f13_flow_warp")
    xw, yw = R.warp_events_flow_torch(g["xs"], g["ys"], g["ts"], None, g["flow"])
    assert np.max(np.abs(xw - g["xw"])) <= 2e-7 * np.max(np.abs(g["xw"]))
    assert np.max(np.abs(yw - g["yw"])) <= 2e-7 * np.max(np.abs(g["yw"]))
    xw, yw = R.warp_events_flow_torch(g["xs"], g["ys"], g["ts"], None, g["flow"], t0=0.02)
    assert np.max(np.abs(xw - g["xw_t0"])) <= 2e-7 * np.max(np.abs(g["xw_t0"]))


def _f8_events(golden):
    g8 = golden("f8_objective")
    return f64(g8["xs"]), f64(g8["ys"]), f64(g8["ts"]), f64(g8["ps"]), tuple(int(v) for v in g8["img_size"])


def test_f14_grid_search_and_landscape(golden):
    """grid_search_initial / grid_search_optimisation / find_new_range / draw_objective_function's image
    (events_cmax.py:103-311) -- the parameter-space samplers of SURVEY 8(f) rank 1."""
    g = golden("f14_search")
    x, y, t, p, img_size = _f8_events(golden)
    w = R.linvel_warp()
    for tag, kw in (("log5", dict(log_scale=True, num_samples_per_param=5)),
                    ("lin7", dict(log_scale=False, num_samples_per_param=7, param_ranges=[[-60, 60], [-90, 30]]))):
        r = R.grid_search_initial(x, y, t, p, w, R.variance_objective(), img_size, **kw)
        assert np.array_equal(np.array(r["search_axes"]), g["gsi_%s_axes" % tag])
        assert np.array_equal(np.array(r["params"]), g["gsi_%s_params" % tag])
        assert np.array_equal(np.array(r["eval"], dtype=np.float64), g["gsi_%s_eval" % tag])
        assert np.array_equal(np.array(r["min_params"]), g["gsi_%s_min_params" % tag])
    for q, rng in zip(g["fnr_params"], g["fnr_ranges"]):
        assert np.array_equal(np.array(R.find_new_range(g["fnr_axes"], q)), rng)
    r = R.grid_search_optimisation(x, y, t, p, w, R.variance_objective(), img_size, log_scale=False)
    assert np.array_equal(np.array(r["min_params"]), g["gso_min_params"])
    a = g["landscape_args"]
    obj = R.variance_objective(minimum_events=1)
    img = R.objective_landscape(x, y, t, p, obj, w, x_range=(a[0], a[1]), y_range=(a[2], a[3]), resolution=a[4],
                                img_size=img_size)
    assert img.shape == g["landscape"].shape
    assert np.array_equal(img, g["landscape"])


def test_f14_rms_lifespan_cut_segmentation(golden):
    g = golden("f14_search")
    x, y, t, p, img_size = _f8_events(golden)
    w = R.linvel_warp()
    rms = R.rms_objective()
    k = 0
    for q in g["rms_params"]:
        for s in (None, 0.0):
            assert np.float64(rms.evaluate_function(q, x, y, t, p, w, img_size, blur_sigma=s)) == g["rms_f"][k]
            assert np.array_equal(f64(rms.evaluate_gradient(q, x, y, t, p, w, img_size, blur_sigma=s)), g["rms_g"][k])
            k += 1
    for i, prm in enumerate(([400., -250.], [4000., -2500.])):
        cut = R.cut_events_to_lifespan(x, y, t, p, np.array(prm), 5, minimum_events=5000)
        assert len(cut[0]) == g["cut_len"][i] and cut[2][0] == g["cut_first_t"][i]
    assert np.array_equal(R.segmentation_mask_from_d_iwe(g["seg_d_iwe"]), g["seg_mask"])
    assert np.array_equal(R.segmentation_mask_from_d_iwe(g["seg_d_iwe"], th=0.05), g["seg_mask_th"])


def test_f15_native_dtypes(golden):
    """Events in the reference's on-disk dtypes (int16 / float64 epoch seconds / bool): the loaders' widening restated
    (widen_native_events) + the torch path == what the reference computes; the numpy path takes them as stored."""
    g = golden("f15_native_dtypes")
    xs, ys, ts, ps = g["xs"], g["ys"], g["ts"], g["ps"]
    ss, B = tuple(int(v) for v in g["sensor_size"]), int(g["B"])
    assert xs.dtype == np.int16 and ts.dtype == np.float64 and ps.dtype == np.bool_
    assert np.array_equal(R.events_to_voxel(xs, ys, ts, ps * 2.0 - 1.0, B, sensor_size=ss), g["voxel_numpy_f64"])
    cols = R.widen_native_events(xs, ys, ts, ps)
    assert all(c.dtype == np.float32 for c in cols) and set(np.unique(cols[3])) == {-1.0, 1.0} and cols[2][0] == 0
    assert np.array_equal(R.events_to_voxel_torch(*cols, B, sensor_size=ss), g["voxel_torch_widened"])
    xy = np.stack((xs, ys), axis=1)
    assert all(np.array_equal(a, b) for a, b in zip(R.widen_native_events(xy, None, ts, ps), cols))
    lit = R.widen_native_events(xs, ys, cols[2], ps.astype(np.uint8), t_offset=0.0, polarity="literal")
    assert np.array_equal(lit[2], cols[2]) and set(np.unique(lit[3])) == {0.0, 1.0}
    assert np.array_equal(R.events_to_voxel_torch(*lit, B, sensor_size=ss), g["voxel_torch_narrow_literal"])
    # float32 of the ABSOLUTE epoch timestamps would be useless: every event collapses onto one value
    assert len(np.unique(ts.astype(np.float32))) == 1
    # the float32 torch grid agrees with the float64 numpy grid to float32 rounding of t_norm
    assert np.max(np.abs(g["voxel_torch_widened"] - g["voxel_numpy_f64"])) <= 1e-5 * np.max(np.abs(g["voxel_numpy_f64"]))
    obj, w = R.variance_objective(), R.linvel_warp()
    xf, yf, tf, pf = xs.astype(np.float64), ys.astype(np.float64), ts - ts[0], ps * 2.0 - 1.0
    for q, f, gr in zip(g["cmax_params"], g["cmax_f"], g["cmax_g"]):
        assert np.float64(obj.evaluate_function(q, xf, yf, tf, pf, w, ss, blur_sigma=1.0)) == f
        assert np.array_equal(f64(obj.evaluate_gradient(q, xf, yf, tf, pf, w, ss, blur_sigma=1.0)), gr)


def test_torch_cpu_restatement_matches_f3(golden):
    """oracle/reference_torch_cpu.py (the multi-threaded CPU baseline of bench.py) against the reference's own torch
    output; one thread so that index_put_ accumulates in a fixed order."""
    import torch
    from oracle import reference_torch_cpu as T
    g = golden("f3_voxel_torch")
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        for tag, Bs in (("small", (1, 2, 5, 9)), ("dvs", (5,))):
            cols = [torch.from_numpy(np.asarray(g[tag + k], dtype=np.float32)) for k in ("_xs", "_ys", "_ts", "_ps")]
            for B in Bs:
                if B == 1:
                    continue                      # dt * 0: the reference's NaN grid is covered by the numpy oracle
                v = T.events_to_voxel_torch(*cols, B, sensor_size=tuple(g[tag + "_sensor_size"]))
                ref = g["%s_voxel_B%d" % (tag, B)]
                assert np.max(np.abs(v.numpy() - ref)) <= 1e-6 * np.max(np.abs(ref))
    finally:
        torch.set_num_threads(nthreads)


def test_torch_cpu_restatement_of_the_objective_matches_f8(golden):
    """The CPU baseline of bench.py's contrast-maximisation numbers (numpy warp + torch splat + scipy blur, as the
    reference composes them) against the reference's own objective values and gradients."""
    import torch
    from oracle import reference_torch_cpu as T
    g = golden("f8_objective")
    x, y, t, p = (f64(g[k]) for k in ("xs", "ys", "ts", "ps"))
    size = tuple(int(v) for v in g["img_size"])
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        for i, prm in enumerate(g["params"]):
            for j, s in enumerate(g["sigmas"]):
                assert np.isclose(T.variance_f(prm, x, y, t, p, size, size, float(s)), g["f"][i, j], rtol=1e-6, atol=0)
                assert np.allclose(T.variance_grad(prm, x, y, t, p, size, size, float(s)), g["grad"][i, j], rtol=1e-5, atol=1e-9)
    finally:
        torch.set_num_threads(nthreads)


def test_f16_windowed_and_split_voxel_conventions(golden):
    """The window / split conventions of voxel_grid.py:37-112,155-243 as the reference itself produced them:
    range(0, len - n, n) windows (the last full one dropped when n divides len), np.arange(t0, t1 - T, T) time windows
    located with searchsorted, timesync [t0, t1), ps > 0 / ps <= 0 (torch) and truthiness (numpy) splits -- each window
    being one oracle voxelisation."""
    g = golden("f16_voxel_windows")
    x, y, t, p = g["xs"], g["ys"], g["ts"], g["ps"]
    ss, B = tuple(int(v) for v in g["sensor_size"]), int(g["B"])
    vox = lambda a, b, w=None: R.events_to_voxel_torch(x[a:b], y[a:b], t[a:b], (p if w is None else w)[a:b], B, sensor_size=ss)
    for key, n in (("fixed_n", int(g["fixed_n_n"])), ("fixed_n_div", int(g["fixed_n_div_n"]))):
        starts = list(range(0, len(x) - n, n))
        assert len(starts) == len(g[key])
        for k, s in enumerate(starts):
            assert np.array_equal(vox(s, s + n), g[key][k])
    T = float(g["fixed_t_t"])
    t_starts = np.arange(t[0].item(), t[-1].item() - T, T)
    assert len(t_starts) == len(g["fixed_t"])
    for k, ts0 in enumerate(t_starts):
        a, b = np.searchsorted(t, ts0), np.searchsorted(t, ts0 + T)
        assert np.array_equal(vox(a, b), g["fixed_t"][k])
    a, b = np.searchsorted(t, 0.2), np.searchsorted(t, 0.5)
    assert np.array_equal(vox(a, b), g["timesync"])
    assert np.array_equal(vox(0, len(x), (p > 0).astype(np.float32)), g["neg_pos_torch_pos"])
    assert np.array_equal(vox(0, len(x), (p <= 0).astype(np.float32)), g["neg_pos_torch_neg"])
    xi, yi, t64 = x.astype(np.int64), y.astype(np.int64), t.astype(np.float64)
    assert np.array_equal(R.events_to_voxel(xi, yi, t64, np.where(p, 1, 0), B, sensor_size=ss), g["neg_pos_numpy_pos"])
    assert np.array_equal(R.events_to_voxel(xi, yi, t64, np.where(p, 0, 1), B, sensor_size=ss), g["neg_pos_numpy_neg"])


def _drive_image_classes(M, g):
    """The sequence of calls oracle/make_golden_classes.py made on the real reference, on module / namespace M."""
    xs, ys, ts, ps = g["xs"], g["ys"], g["ts"], g["ps"]
    H, W = (int(v) for v in g["sensor_size"])
    n = len(xs)
    out = {}
    ti = M.TimestampImage((H, W))
    out["ts_init_image"] = np.asarray(ti.get_image())
    ti.set_init(-1.0)
    half = n // 2
    ti.add_events(xs[:half], ys[:half], ts[:half], ps[:half])
    out["ts_image_half"] = np.array(ti.image)
    out["ts_get_half"] = np.asarray(ti.get_image())
    ti.add_events(xs[half:], ys[half:], ts[half:], ps[half:])
    ti.add_event(3.7, 5.2, 0.123, 1)
    out["ts_image_full"] = np.array(ti.image)
    out["ts_get_full"] = np.asarray(ti.get_image())
    ts2 = M.TimestampImage((H, W))
    ts2.add_events(xs[:300], ys[:300], ts[:300] + 2.0, ps[:300])
    out["ts_sparse_image"] = np.array(ts2.image)
    out["ts_sparse_get"] = np.asarray(ts2.get_image())
    ei = M.EventImage((H, W))
    ei.add_events(xs, ys, ts, ps)
    out["ev_image_after_add_events"] = np.array(ei.image)
    for k in range(0, 2000):
        ei.add_event(xs[k], ys[k], ts[k], ps[k])
    out["ev_image"] = np.array(ei.image)
    out["ev_get"] = np.asarray(ei.get_image())
    return out


def test_f17_timestamp_image_and_event_image_classes(golden):
    """image.py:355-396: last-writer-wins time-stamp image with dense-rank normalisation, and the event image whose
    add_events adds zeros upstream -- every intermediate image and get_image() of the reference, bit for bit."""
    g = golden("f17_image_classes")
    got = _drive_image_classes(R, g)
    for k, v in got.items():
        assert v.dtype == np.float64 and np.array_equal(v, g[k], equal_nan=True), k
    with pytest.raises(IndexError):
        R.TimestampImage((4, 4)).add_events(np.array([1.0, 4.0]), np.array([1.0, 1.0]), np.array([0.1, 0.2]), None)
    ei = R.EventImage((4, 4))
    ei.add_events(np.array([1.0, 1.0]), np.array([2.0, 2.0]), np.array([0.1, 0.2]), np.array([1.0, 1.0]), use_polarity=True)
    assert ei.image[2, 1] == 3.0
