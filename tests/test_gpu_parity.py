"""GPU parity tests (-m gpu): the HIP path (through the C ABI, via the reference-signature Python API) against
(i) the golden vectors produced by the real reference and (ii) the numpy oracle on seeded inputs.
Bars: bit-exact for integer event images; float paths |a-b| <= 1e-5 * max|ref| (BASELINE.json north_star) -- float
atomics sum in arbitrary order, so bit-equality is only demanded where the arithmetic is order-free."""
import numpy as np
import pytest
import torch

from oracle import reference_np as R

pytestmark = pytest.mark.gpu

TOL = 1e-5


def f64(a):
    return np.asarray(a, dtype=np.float64)


def close(a, ref, tol=TOL):
    a, ref = f64(a), f64(ref)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    scale = max(np.max(np.abs(ref)), 1e-30)
    err = np.max(np.abs(a - ref))
    assert err <= tol * scale, "max err %.3e vs tol %.3e (scale %.3e)" % (err, tol * scale, scale)


@pytest.fixture(scope="module")
def E():
    import event_utils_amd as E
    from event_utils_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available()
    return E


# ------------------------------------------------------------------------------------------------ F1 / C1
@pytest.mark.parametrize("impl", ["auto", "tiled"])
def test_f1_image_nearest_int_bit_exact(E, golden, impl, monkeypatch):
    monkeypatch.setenv("EVK_IMPL", impl)   # "tiled": the one-pass partition + LDS tiles (evk_image2.hip) at any size
    g = golden("f1_image_nearest_int")
    xs, ys, ps = g["xs"].astype(np.int64), g["ys"].astype(np.int64), g["ps"].astype(np.int64)
    ss = tuple(g["sensor_size"])
    out = E.events_to_image(xs, ys, ps, sensor_size=ss)
    assert out.dtype == np.float64 and np.array_equal(out, g["img_pm"])
    assert np.array_equal(E.events_to_image(xs, ys, np.ones_like(ps), sensor_size=ss), g["img_cnt"])
    assert np.array_equal(E.events_to_image(xs, ys, ps, sensor_size=ss, meanval=True), g["img_mean"])
    assert np.array_equal(E.events_to_image(xs, ys, ps, sensor_size=ss, meanval=True, default=7), g["img_mean_default"])
    close(E.events_to_image(xs, ys, g["wf"], sensor_size=ss), g["img_wf"], 1e-12)


def test_image_nearest_errors(E):
    xs = np.array([0, 241]); ys = np.array([0, 0]); ps = np.array([1, 1])
    with pytest.raises(ValueError):
        E.events_to_image(xs, ys, ps)
    with pytest.raises(ValueError):
        E.events_to_image(np.array([-1, 3]), ys, ps)
    with pytest.raises(TypeError):
        E.events_to_image(xs.astype(float), ys.astype(float), ps)


def test_c1_image_1m_events_bit_exact(E):
    """BASELINE.json configs[0]: 1M events, 240x180, nearest, integer counts."""
    rng = np.random.default_rng(0)
    n, H, W = 1_000_000, 180, 240
    xs, ys = rng.integers(0, W, n), rng.integers(0, H, n)
    ps = rng.integers(0, 2, n) * 2 - 1
    assert np.array_equal(E.events_to_image(xs, ys, ps, sensor_size=(H, W)), R.events_to_image(xs, ys, ps, sensor_size=(H, W)))
    cnt = E.events_to_image(xs, ys, np.ones_like(ps), sensor_size=(H, W))
    assert np.array_equal(cnt, R.events_to_image(xs, ys, np.ones_like(ps), sensor_size=(H, W)))
    assert cnt.sum() == n


def test_image_empty_and_ragged(E):
    e = np.array([], dtype=np.int64)
    assert np.array_equal(E.events_to_image(e, e, e), np.zeros((180, 240)))
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 5, 63, 64, 65, 1023):
        xs, ys, ps = rng.integers(0, 240, n), rng.integers(0, 180, n), rng.integers(-3, 4, n)
        assert np.array_equal(E.events_to_image(xs, ys, ps), R.events_to_image(xs, ys, ps))


# ------------------------------------------------------------------------------------------------ F2 / F3 voxel
@pytest.mark.parametrize("tag,Bs", [("small", (1, 2, 5, 9)), ("dvs", (5,))])
def test_f2_voxel_numpy(E, golden, tag, Bs):
    g = golden("f2_voxel_numpy")
    xs, ys = g[tag + "_xs"].astype(np.int64), g[tag + "_ys"].astype(np.int64)
    ts, ps = g[tag + "_ts"], f64(g[tag + "_ps"])
    for B in Bs:
        v = E.events_to_voxel(xs, ys, ts, ps, B, sensor_size=tuple(g[tag + "_sensor_size"]))
        assert v.dtype == np.float64
        close(v, g["%s_voxel_B%d" % (tag, B)], 1e-12)


@pytest.mark.parametrize("impl", ["auto", "direct", "tiled", "tiled-rec4"])
@pytest.mark.parametrize("tag,Bs", [("small", (1, 2, 5, 9)), ("dvs", (5,))])
def test_f3_voxel_torch(E, golden, tag, Bs, impl, monkeypatch):
    # every kernel family against the reference's own outputs: global atomics, one-pass partition (8-byte records, and the
    # 4-byte records a call takes above 16 M events)
    from event_utils_amd import tiled
    monkeypatch.setenv("EVK_IMPL", impl.split("-")[0])
    monkeypatch.setitem(tiled.FORCE, "rec", 4 if impl.endswith("rec4") else 8)
    g = golden("f3_voxel_torch")
    xs, ys, ts = (torch.from_numpy(g[tag + k]) for k in ("_xs", "_ys", "_ts"))
    ps = torch.from_numpy(g[tag + "_ps"].astype(np.float32))
    ss = tuple(g[tag + "_sensor_size"])
    for B in Bs:
        v = E.events_to_voxel_torch(xs, ys, ts, ps, B, sensor_size=ss)
        assert v.dtype == torch.float32 and v.device.type == "cpu"       # device=None -> xs.device
        close(v.numpy(), g["%s_voxel_B%d" % (tag, B)])
        vd = E.events_to_voxel_torch(xs.cuda(), ys.cuda(), ts.cuda(), ps.cuda(), B, sensor_size=ss)
        assert vd.is_cuda
        close(vd.cpu().numpy(), g["%s_voxel_B%d" % (tag, B)])
    if tag == "dvs":
        v = E.events_to_voxel_torch(xs.long(), ys.long(), ts, ps, 5, sensor_size=ss)
        close(v.numpy(), g["dvs_voxel_B5_long"])


def test_voxel_errors_and_edges(E):
    x = torch.tensor([1., 2., 300.]); y = torch.tensor([1., 2., 3.]); t = torch.tensor([0., .5, 1.]); p = torch.ones(3)
    with pytest.raises(IndexError):
        E.events_to_voxel_torch(x, y, t, p, 3)
    with pytest.raises(RuntimeError):
        E.events_to_voxel_torch(y, y, t.double(), p, 3)
    with pytest.raises(AssertionError):
        E.events_to_voxel_torch(y, y, t[:2], p, 3)
    # negative indices wrap like torch's index_put_
    v = E.events_to_voxel_torch(torch.tensor([-1., 2.]), torch.tensor([-1., 0.]), torch.tensor([0., 1.]), torch.ones(2), 2,
                                sensor_size=(4, 5))
    ref = R.events_to_voxel_torch(np.array([-1., 2.], np.float32), np.array([-1., 0.], np.float32),
                                  np.array([0., 1.], np.float32), np.ones(2, np.float32), 2, sensor_size=(4, 5))
    assert np.array_equal(v.numpy(), ref)
    # dt == 0 -> NaN everywhere the events land (Q9)
    v = E.events_to_voxel_torch(y, y, torch.ones(3), p, 3, sensor_size=(8, 8)).numpy()
    ref = R.events_to_voxel_torch(y.numpy(), y.numpy(), np.ones(3, np.float32), p.numpy(), 3, sensor_size=(8, 8))
    assert np.array_equal(np.isnan(v), np.isnan(ref)) and np.isnan(v).sum() == 9
    with pytest.raises(ValueError):
        E.events_to_voxel(np.array([1, 500]), np.array([1, 1]), np.array([0., 1.]), np.array([1., 1.]), 2)
    with pytest.raises(TypeError):
        E.events_to_voxel(np.array([1., 5.]), np.array([1., 1.]), np.array([0., 1.]), np.array([1., 1.]), 2)


@pytest.mark.parametrize("n", [1, 7, 64, 1001, 200_003])
def test_voxel_vs_oracle_ragged(E, n):
    rng = np.random.default_rng(n)
    H, W, B = 48, 64, 5
    x = rng.uniform(0, W, n).astype(np.float32); y = rng.uniform(0, H, n).astype(np.float32)
    x[x >= W] = W - 1; y[y >= H] = H - 1
    t = np.sort(rng.uniform(0, 1, n)).astype(np.float32) if n > 1 else np.array([0.5], np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    if n == 1:
        return   # dt == 0: covered above
    ref = R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64")
    v = E.events_to_voxel_torch(*(torch.from_numpy(a) for a in (x, y, t, p)), B, sensor_size=(H, W)).numpy()
    close(v, ref)
    # unaligned views (slices start at odd offsets): the scalar-load kernel variant
    xs, ys, ts_, ps = (torch.from_numpy(a).cuda()[1:] for a in (x, y, t, p))
    if n > 2:
        ref = R.events_to_voxel_torch(x[1:], y[1:], t[1:], p[1:], B, sensor_size=(H, W), accum="f64")
        close(E.events_to_voxel_torch(xs, ys, ts_, ps, B, sensor_size=(H, W)).cpu().numpy(), ref)


def test_c2_voxel_vga_mass_and_oracle(E):
    """BASELINE.json configs[1] shape (640x480, 5 bins): 2M events vs the oracle, and at the full 10M events the
    size-independent properties: mass conservation (sum(voxel) == sum(ps)) and shard additivity f(A u B) = f(A)+f(B)."""
    from event_utils_amd.representations.voxel_grid import _voxel_f32_device
    rng = np.random.default_rng(1)
    H, W, B, n = 480, 640, 5, 10_000_000
    x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    m = 2_000_000
    ref = R.events_to_voxel_torch(x[:m], y[:m], t[:m], p[:m], B, sensor_size=(H, W), accum="f64")
    xd, yd, td, pd = (torch.from_numpy(a).cuda() for a in (x, y, t, p))
    close(E.events_to_voxel_torch(xd[:m], yd[:m], td[:m], pd[:m], B, sensor_size=(H, W)).cpu().numpy(), ref)
    full = E.events_to_voxel_torch(xd, yd, td, pd, B, sensor_size=(H, W))
    assert abs(full.double().sum().item() - float(p.astype(np.float64).sum())) <= 1e-3 * np.sqrt(n)
    t0, t1 = float(t[0]), float(t[-1])
    half = n // 2
    a = _voxel_f32_device(xd[:half], yd[:half], td[:half], pd[:half], B, (H, W), t0, t1)
    b = _voxel_f32_device(xd[half:], yd[half:], td[half:], pd[half:], B, (H, W), t0, t1)
    close((a + b).cpu().numpy(), full.cpu().numpy())
    # all-positive weights: every bin plane is non-negative and the count is conserved
    cnt = E.events_to_voxel_torch(xd, yd, td, torch.ones_like(pd), B, sensor_size=(H, W))
    assert cnt.min().item() >= 0 and abs(cnt.double().sum().item() - n) <= 1e-6 * n


# ------------------------------------------------------------------------------------------------ F4 image torch
@pytest.mark.parametrize("impl", ["auto", "tiled"])
def test_f4_image_torch(E, golden, impl, monkeypatch):
    monkeypatch.setenv("EVK_IMPL", impl)
    g = golden("f4_image_torch")
    xs, ys, ps = (torch.from_numpy(g[k]) for k in ("xs", "ys", "ps"))
    ss = tuple(g["sensor_size"])
    for key, kw in (("bil_pad", dict(interpolation='bilinear', padding=True)),
                    ("bil_nopad", dict(interpolation='bilinear', padding=False)),
                    ("near_pad", dict(interpolation=None, padding=True)),
                    ("near_nopad", dict(interpolation=None, padding=False)),
                    ("near_default3", dict(interpolation=None, padding=False, default=3))):
        out = E.events_to_image_torch(xs, ys, ps, sensor_size=ss, **kw)
        assert out.dtype == torch.float32
        close(out.numpy(), g[key])
    # interpolate_to_image (image.py:102-115) on the pixels, fractions and masked weights events_to_image_torch hands it upstream
    # (image.py:79-86): the reference's own padded bilinear image, through either kernel family (round 6: the one-pass path too)
    clipx, clipy = ss[1], ss[0]                                   # padded image (H+1, W+1): thresholds img_size - 1
    mask = ((xs < clipx) & (ys < clipy)).float()
    pxs, pys = xs.floor(), ys.floor()
    dxs, dys = xs - pxs, ys - pys
    img = torch.zeros(ss[0] + 1, ss[1] + 1)
    E.interpolate_to_image((pxs * mask).long(), (pys * mask).long(), dxs, dys, ps * mask, img)
    close(img.numpy(), g["bil_pad"])
    close(E.events_to_image(f64(g["xs"]), f64(g["ys"]), f64(g["ps"]), sensor_size=ss, interpolation='bilinear', padding=False), g["np_bil"])
    close(E.events_to_image(f64(g["xs"]), f64(g["ys"]), f64(g["ps"]), sensor_size=ss, interpolation='bilinear', padding=True), g["np_bil_pad"])
    with pytest.raises(IndexError):
        E.events_to_image_torch(xs, ys, ps, sensor_size=ss, clip_out_of_range=False, padding=False)
    with pytest.raises(RuntimeError):
        E.events_to_image_torch(xs, ys, ps.double(), sensor_size=ss)


def test_interpolate_primitives(E):
    rng = np.random.default_rng(7)
    n, H, W = 5000, 33, 47
    px = rng.integers(0, W - 1, n); py = rng.integers(0, H - 1, n)
    dx = rng.random(n).astype(np.float32); dy = rng.random(n).astype(np.float32)
    w = rng.normal(size=n).astype(np.float32)
    w1 = rng.normal(size=(2, n)).astype(np.float32); w2 = rng.normal(size=(2, n)).astype(np.float32)
    ref = R.interpolate_to_image(px, py, dx, dy, w, np.zeros((H, W), np.float32), accum="f64")
    img = torch.zeros(H, W)
    out = E.interpolate_to_image(*(torch.from_numpy(a) for a in (px, py, dx, dy, w)), img)
    assert out is img
    close(img.numpy(), ref)
    refd = R.interpolate_to_derivative_img(px, py, dx, dy, np.zeros((2, H, W), np.float32), w1, w2, accum="f64")
    dimg = torch.zeros(2, H, W, device="cuda")
    E.interpolate_to_derivative_img(*(torch.from_numpy(a).cuda() for a in (px, py, dx, dy)), dimg,
                                    torch.from_numpy(w1).cuda(), torch.from_numpy(w2).cuda())
    close(dimg.cpu().numpy(), refd)
    with pytest.raises(IndexError):
        E.interpolate_to_image(torch.tensor([W - 1]), torch.tensor([0]), torch.tensor([.5]), torch.tensor([.5]),
                               torch.tensor([1.]), torch.zeros(H, W))


def test_normalised_time_is_bit_identical_to_float32_division():
    """voxel_grid.py:134, t_norm = (ts - ts[0]) / dt * (B - 1) in float32: evk_normalise_time_f32 runs the very function the
    partition kernel calls for every event (evk_part.h, time_norm).  10^8 samples over 40 (ts[0], dt) pairs -- ordinary streams,
    tiny and huge dt, negative dt (unsorted ends), dt == 0, time stamps outside [ts[0], ts[-1]], subnormal quotients,
    infinities and NaN -- against numpy's float32 arithmetic, bit for bit."""
    from event_utils_amd import _lib, _device as D
    rng = np.random.default_rng(11)
    pairs = [(0.0, 0.1), (0.0, 1.0), (1.6e9, 1.6e9 + 64.0), (3.25, 3.25 + 2.0 ** -20), (0.0, 3e38), (0.0, 1e-38), (5.0, 5.0),
             (1.0, -1.0), (-7.5, 12.25), (0.0, 2.0 ** -126), (0.0, float(np.float32(1) / np.float32(3)))]
    while len(pairs) < 40:
        a = float(rng.uniform(-10, 10)); pairs.append((a, a + float(10.0 ** rng.uniform(-6, 6))))
    n = 2_500_000
    bad = 0
    for k, (t0, t1) in enumerate(pairs):
        t0, t1 = np.float32(t0), np.float32(t1)
        t = (t0 + (t1 - t0) * rng.random(n).astype(np.float32)).astype(np.float32)
        t[::97] = (t0 + (t1 - t0) * np.float32(3.0) * rng.standard_normal(len(t[::97])).astype(np.float32))   # outside the range
        t[1::1009] = t0; t[2::1009] = t1
        t[3::5003] = t0 + np.float32(1e-45) * rng.integers(0, 50, len(t[3::5003])).astype(np.float32)            # tiny numerators
        t[5:9] = [np.inf, -np.inf, np.nan, 0.0]
        for B in (5, 2, 1, 16)[: 1 + (k % 4)]:
            with np.errstate(all="ignore"):
                ref = (t - t0) / (t1 - t0) * np.float32(B - 1)
            td = torch.from_numpy(t).cuda(); out = torch.empty_like(td)
            _lib.call("evk_normalise_time_f32", D.ptr(td), n, float(t0), float(t1), B, D.ptr(out), D.stream())
            got = out.cpu().numpy()
            same = (got.view(np.uint32) == ref.astype(np.float32).view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
            bad += int(np.count_nonzero(~same))
    assert bad == 0, "%d of 10^8 normalised time stamps differ from the float32 division" % bad


# ------------------------------------------------------------------------------------------------ F5 warp
def test_f5_warp_bit_exact(E, golden):
    g = golden("f5_warp")
    x, y, t, p = f64(g["xs"]), f64(g["ys"]), f64(g["ts"]), f64(g["ps"])
    w = E.linvel_warp()
    for i, prm in enumerate(g["params"]):
        xp, yp, jx, jy = w.warp(x, y, t, p, t[-1], prm, compute_grad=True)
        for a, k in ((xp, "xp"), (yp, "yp"), (jx, "jx"), (jy, "jy")):
            assert a.dtype == np.float64 and np.array_equal(a, g["%s%d" % (k, i)])
        assert np.array_equal(E.events_bounds_mask(xp, yp, 0, 240, 0, 180), g["mask%d" % i])
    xp, yp, jx, jy = E.warp_events(x, y, t, p, t[-1], g["params"][1])
    assert jx is None and jy is None and np.array_equal(xp, g["xp1"])


# ------------------------------------------------------------------------------------------------ F6 / F7 IWE
@pytest.mark.parametrize("impl", ["auto", "direct", "tiled"])
def test_f6_get_iwe_verbatim(E, golden, impl, monkeypatch):
    monkeypatch.setenv("EVK_IMPL", impl)
    g = golden("f6_get_iwe")
    x, y, t, p = f64(g["xs"]), f64(g["ys"]), f64(g["ts"]), f64(g["ps"])
    w = E.linvel_warp()
    img_size = tuple(g["img_size"])
    for i, prm in enumerate(g["params"]):
        iwe, diwe = E.get_iwe(prm, x, y, t, p, w, img_size, compute_gradient=True)
        assert iwe.shape == (181, 241) and diwe.shape == (2, 181, 241) and iwe.dtype == np.float32
        close(iwe, g["iwe%d" % i]); close(diwe, g["diwe%d" % i])
    iwe, d = E.get_iwe(g["params"][0], x, y, t, p, w, img_size, compute_gradient=False, use_polarity=False)
    assert d is None
    close(iwe, g["iwe_nopol0"])
    x, y, t, p = f64(g["q1_xs"]), f64(g["q1_ys"]), f64(g["q1_ts"]), f64(g["q1_ps"])
    iwe, diwe = E.get_iwe(np.array([30., -20.]), x, y, t, p, w, tuple(g["q1_img_size"]), compute_gradient=True)
    close(iwe, g["q1_iwe"]); close(diwe, g["q1_diwe"])


@pytest.mark.parametrize("impl", ["auto", "direct", "tiled"])
@pytest.mark.parametrize("tag", ["s48", "vga"])
def test_f7_iwe_sized_and_generic_plugin_path(E, golden, tag, impl, monkeypatch):
    monkeypatch.setenv("EVK_IMPL", impl)
    g = golden("f7_iwe_sized")
    x, y, t, p = f64(g[tag + "_xs"]), f64(g[tag + "_ys"]), f64(g[tag + "_ts"]), f64(g[tag + "_ps"])
    ss = tuple(g[tag + "_sensor_size"])
    iwe, diwe = E.get_iwe(g["params"], x, y, t, p, E.linvel_warp(), ss, compute_gradient=True, sensor_size=ss)
    close(iwe, g[tag + "_iwe"]); close(diwe, g[tag + "_diwe"])

    class host_warp(E.warp_function):        # a user plugin: numpy in, numpy out -> generic mask + splat kernels
        def __init__(self):
            super().__init__("host_linvel", 2)

        def warp(self, xs, ys, ts, ps, t0, params, compute_grad=False):
            return R.linvel_warp().warp(xs, ys, ts, ps, t0, params, compute_grad)
    iwe2, diwe2 = E.get_iwe(g["params"], x, y, t, p, host_warp(), ss, compute_gradient=True, sensor_size=ss)
    close(iwe2, g[tag + "_iwe"]); close(diwe2, g[tag + "_diwe"])
    # f64 columns (exact for arbitrary float64 input)
    from event_utils_amd.events import DeviceEvents
    ev = DeviceEvents.from_arrays(x, y, t, p, precision="f64")
    iwe3, diwe3 = E.get_iwe(g["params"], ev, None, None, None, E.linvel_warp(), ss, compute_gradient=True, sensor_size=ss)
    close(iwe3, g[tag + "_iwe"]); close(diwe3, g[tag + "_diwe"])
    out = E.get_iwe(g["params"], x, y, t, p, E.linvel_warp(), ss, return_events=True, sensor_size=ss)
    xw, yw, _, _ = R.linvel_warp().warp(x, y, t, p, t[-1], g["params"])
    m = R.events_bounds_mask(xw, yw, 0, ss[1], 0, ss[0])
    assert np.array_equal(out[2][0], xw * m) and np.array_equal(out[2][1], yw * m)


def test_c3_iwe_vga_vs_oracle(E):
    """BASELINE.json configs[2] shape: 640x480 warp + IWE (+dIWE), 1M events vs the oracle's f64 sum."""
    rng = np.random.default_rng(2)
    H, W, n = 480, 640, 1_000_000
    x = rng.uniform(1, W - 1, n).astype(np.float32); y = rng.uniform(1, H - 1, n).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    prm = np.array([30., -20.])
    ref_iwe, ref_d = R.get_iwe(prm, f64(x), f64(y), f64(t), f64(p), R.linvel_warp(), (H, W), compute_gradient=True,
                               sensor_size=(H, W), accum="f64")
    iwe, diwe = E.get_iwe(prm, x, y, t, p, E.linvel_warp(), (H, W), compute_gradient=True, sensor_size=(H, W))
    close(iwe, ref_iwe); close(diwe, ref_d)
    # large flow: many events leave the sensor
    prm = np.array([-3000., 2500.])
    ref_iwe, ref_d = R.get_iwe(prm, f64(x), f64(y), f64(t), f64(p), R.linvel_warp(), (H, W), compute_gradient=True,
                               sensor_size=(H, W), accum="f64")
    iwe, diwe = E.get_iwe(prm, x, y, t, p, E.linvel_warp(), (H, W), compute_gradient=True, sensor_size=(H, W))
    close(iwe, ref_iwe); close(diwe, ref_d)


# ------------------------------------------------------------------------------------------------ F10 blur
def test_f10_blur(E, golden):
    from event_utils_amd.contrast_max.objectives import gaussian_filter_device
    g = golden("f10_blur")
    a3 = torch.from_numpy(g["a3"]).cuda()
    for s in (1.0, 2.0, 0.5):
        assert np.array_equal(gaussian_filter_device(a3, s).cpu().numpy(), g["blur3_s%g" % s])
        assert np.array_equal(gaussian_filter_device(a3[0], s).cpu().numpy(), g["blur2_s%g" % s])
    assert np.array_equal(gaussian_filter_device(torch.from_numpy(g["small"]).cuda(), 1.0).cpu().numpy(), g["small_blur_s1"])


# ------------------------------------------------------------------------------------------------ F8 / F9 objective
def test_f8_objective(E, golden):
    g = golden("f8_objective")
    x, y, t, p = f64(g["xs"]), f64(g["ys"]), f64(g["ts"]), f64(g["ps"])
    w, obj = E.linvel_warp(), E.variance_objective()
    img_size = tuple(g["img_size"])
    from event_utils_amd.events import DeviceEvents
    ev = DeviceEvents.from_arrays(x, y, t, p)
    assert ev.dtype == torch.float32            # fixture columns are float32-lossless
    for i, prm in enumerate(g["params"]):
        for j, s in enumerate(g["sigmas"]):
            f = obj.evaluate_function(prm, ev, None, None, None, w, img_size, blur_sigma=s)
            gr = obj.evaluate_gradient(prm, ev, None, None, None, w, img_size, blur_sigma=s)
            assert isinstance(f, np.float32) and gr.shape == (2,) and gr.dtype == np.float32
            assert abs(f - g["f"][i, j]) <= TOL * abs(g["f"][i, j])
            assert np.max(np.abs(f64(gr) - g["grad"][i, j])) <= TOL * np.max(np.abs(g["grad"][i, j])) + 1e-9
    # numpy front door, precomputed iwe / d_iwe front door
    f = obj.evaluate_function(g["params"][1], x, y, t, p, w, img_size, blur_sigma=1.0)
    assert abs(f - g["f"][1, 1]) <= TOL * abs(g["f"][1, 1])
    iwe, diwe = E.get_iwe(g["params"][1], x, y, t, p, w, img_size, compute_gradient=True)
    assert abs(obj.evaluate_function(iwe=iwe, blur_sigma=1.0) - g["f"][1, 1]) <= TOL * abs(g["f"][1, 1])
    gr = obj.evaluate_gradient(iwe=iwe, d_iwe=diwe, blur_sigma=1.0)
    assert np.max(np.abs(f64(gr) - g["grad"][1, 1])) <= TOL * np.max(np.abs(g["grad"][1, 1])) + 1e-9
    # adaptive lifespan (Q10)
    al = E.variance_objective(adaptive_lifespan=True, minimum_events=5000)
    al.iter_update(np.array([400., -250.]))
    f = al.evaluate_function(np.array([40., -25.]), ev, None, None, None, w, img_size, blur_sigma=1.0)
    assert al.s_idx == g["al_s_idx"] and abs(f - g["al_f"]) <= TOL * abs(g["al_f"])
    al.iter_update(np.array([400., -250.]))
    gr = al.evaluate_gradient(np.array([40., -25.]), ev, None, None, None, w, img_size, blur_sigma=1.0)
    assert np.max(np.abs(f64(gr) - g["al_grad"])) <= TOL * np.max(np.abs(g["al_grad"]))


def test_consistent_gradient_mode(E, golden):
    """reference_exact=False: the analytic gradient equals the finite-difference gradient of evaluate_function."""
    g = golden("f8_objective")
    x, y, t, p = f64(g["xs"]), f64(g["ys"]), f64(g["ts"]), f64(g["ps"])
    from event_utils_amd.events import DeviceEvents
    ev = DeviceEvents.from_arrays(x, y, t, p)
    w, obj = E.linvel_warp(), E.variance_objective()
    obj.reference_exact = False
    prm = np.array([35., -22.])
    gr = f64(obj.evaluate_gradient(prm, ev, None, None, None, w, (180, 240), blur_sigma=1.0))
    h = 0.05
    fd = np.zeros(2)
    for k in range(2):
        e = np.zeros(2); e[k] = h
        fd[k] = (f64(obj.evaluate_function(prm + e, ev, None, None, None, w, (180, 240), blur_sigma=1.0)) -
                 f64(obj.evaluate_function(prm - e, ev, None, None, None, w, (180, 240), blur_sigma=1.0))) / (2 * h)
    assert np.max(np.abs(gr - fd)) <= 0.05 * np.max(np.abs(fd)) + 1e-4


@pytest.mark.parametrize("mode", ["numeric", "analytic"])
def test_f9_optimize(E, golden, mode):
    g8, g = golden("f8_objective"), golden("f9_optimize_trace")
    x, y, t, p = f64(g8["xs"]), f64(g8["ys"]), f64(g8["ts"]), f64(g8["ps"])
    obj = E.variance_objective()
    trace = []
    f0, g0 = obj.evaluate_function, obj.evaluate_gradient

    def frec(prm, *a, **k):
        v = f0(prm, *a, **k)
        trace.append(("f", np.array(prm, float), float(v), None))
        return v

    def grec(prm, *a, **k):
        v = g0(prm, *a, **k)
        trace.append(("g", np.array(prm, float), None, f64(v)))
        return v
    obj.evaluate_function, obj.evaluate_gradient = frec, grec
    fg0 = obj.evaluate_function_and_gradient

    def fgrec(prm, *a, **k):        # one pass yields the value and the gradient the line search asks for next
        fv, gv = fg0(prm, *a, **k)
        trace.append(("f", np.array(prm, float), float(fv), None))
        trace.append(("g", np.array(prm, float), None, f64(gv)))
        return fv, gv
    obj.evaluate_function_and_gradient = fgrec
    ng0, ngrads = obj.evaluate_function_and_numeric_gradient, []

    def ngrec(prm, *a, **k):        # f(x) and the forward-difference gradient at x from one batched pass
        fv, gv = ng0(prm, *a, **k)
        trace.append(("f", np.array(prm, float), float(fv), None))
        ngrads.append((np.array(prm, float), f64(gv)))
        return fv, gv
    obj.evaluate_function_and_numeric_gradient = ngrec
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        argmax = E.optimize(x, y, t, p, E.linvel_warp(), obj, numeric_grads=(mode == "numeric"), img_size=tuple(g8["img_size"]))
    # first evaluations are pinned tightly (same params in, same values out); the rest of the BFGS trajectory
    # amplifies 1e-7-level summation-order noise, so the end point is compared loosely
    k = 0
    if mode == "numeric":
        # the reference's trace starts f(0,0), f(1,0), f(0,1) (scipy's internal forward differences, epsilon=1);
        # here the three values come from ONE batched pass: f(0,0) and the first gradient estimate must agree
        gf = g["numeric_f"]
        assert np.allclose(g["numeric_params"][:3], [[0, 0], [1, 0], [0, 1]])
        assert trace[0][0] == "f" and abs(trace[0][2] - gf[0]) <= TOL * abs(gf[0])
        assert np.allclose(ngrads[0][0], [0, 0])
        assert np.max(np.abs(ngrads[0][1] - np.array([gf[1] - gf[0], gf[2] - gf[0]]))) <= 2 * TOL * abs(gf[0])
        trace = []
    for kind, prm, fv, gv in trace[:4]:
        assert kind == str(g[mode + "_kind"][k]) and np.allclose(prm, g[mode + "_params"][k], atol=1e-6)
        if kind == "f":
            assert abs(fv - g[mode + "_f"][k]) <= TOL * abs(g[mode + "_f"][k])
        else:
            assert np.max(np.abs(gv - g[mode + "_g"][k])) <= TOL * np.max(np.abs(g[mode + "_g"][k])) + 1e-9
        k += 1
    if mode == "analytic":
        assert len(trace) >= 4
        assert np.linalg.norm(np.asarray(argmax, float) - g[mode + "_argmax"]) < 1.5
        assert np.linalg.norm(np.asarray(argmax, float) - np.array([40., -25.])) < 2.0
    fa = f64(f0(np.asarray(argmax, float), x, y, t, p, E.linvel_warp(), tuple(g8["img_size"]), 1.0))
    fr = f64(f0(g[mode + "_argmax"], x, y, t, p, E.linvel_warp(), tuple(g8["img_size"]), 1.0))
    assert fa <= fr + 0.02 * abs(fr)        # at least as good an optimum as the reference found


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("n", [30_000, 400_000])
def test_value_and_gradient_in_one_pass_equal_the_separate_calls(E, golden, exact, n):
    """evaluate_function_and_gradient (evk_objective_variance_fg_f32 / EVK_POST_VALUE) == evaluate_function +
    evaluate_gradient, on the direct-kernel regime (30 k events) and the tile-bucketed one (400 k), with and without
    blur, reference-exact and consistent gradient; and the reference's own values on the golden scene."""
    H, W = 180, 240
    rng = np.random.default_rng(21)
    x = rng.uniform(1, W - 1, n).astype(np.float32); y = rng.uniform(1, H - 1, n).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32); p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    ev = E.DeviceEvents.from_arrays(x, y, t, p)
    obj, w = E.variance_objective(), E.linvel_warp()
    obj.reference_exact = exact
    for prm in ([0., 0.], [30., -20.], [-250., 400.]):
        for s in (1.0, 0.0, 2.0):
            fv, gv = obj.evaluate_function_and_gradient(np.array(prm), ev, None, None, None, w, (H, W), s)
            f1 = obj.evaluate_function(np.array(prm), ev, None, None, None, w, (H, W), s)
            g1 = obj.evaluate_gradient(np.array(prm), ev, None, None, None, w, (H, W), s)
            assert abs(float(fv) - float(f1)) <= 2e-6 * abs(float(f1))
            # (floor: a gradient of ~1e-6 is the rounding residue of sums of O(1e-2) terms, and below the tiled crossover
            # the float atomics of the direct kernel make two evaluations differ in the last bits)
            assert np.max(np.abs(f64(gv) - f64(g1))) <= 2e-6 * np.max(np.abs(f64(g1))) + 1e-10
    if exact and n == 30_000:
        g8 = golden("f8_objective")
        xs, ys, ts, ps = f64(g8["xs"]), f64(g8["ys"]), f64(g8["ts"]), f64(g8["ps"])
        for i, prm in enumerate(g8["params"]):
            for j, s in enumerate(g8["sigmas"]):
                fv, gv = obj.evaluate_function_and_gradient(prm, xs, ys, ts, ps, w, tuple(g8["img_size"]), float(s))
                assert abs(float(fv) - g8["f"][i, j]) <= TOL * abs(g8["f"][i, j])
                assert np.max(np.abs(f64(gv) - g8["grad"][i, j])) <= TOL * np.max(np.abs(g8["grad"][i, j])) + 1e-9


# ------------------------------------------------------------------------------------------------ windowed voxels
def test_voxel_windows_fixed_n_and_fixed_t(E, monkeypatch):
    """voxel_grids_fixed_n_torch / voxel_grids_fixed_t_torch (voxel_grid.py:37-80): all windows in one launch must
    equal the reference's per-window calls (each window normalises time with its own first / last event)."""
    from event_utils_amd.representations import voxel_grid as V
    rng = np.random.default_rng(11)
    n, H, W, B = 50_000, 40, 56, 4
    x = rng.uniform(0, W, n).astype(np.float32); y = rng.uniform(0, H, n).astype(np.float32)
    x[x >= W] = W - 1; y[y >= H] = H - 1
    t = np.sort(rng.uniform(0, 1.0, n)).astype(np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    tx, ty, tt, tp = (torch.from_numpy(a) for a in (x, y, t, p))
    win = 7000
    got = V.voxel_grids_fixed_n_torch(tx, ty, tt, tp, B, win, sensor_size=(H, W))
    starts = list(range(0, n - win, win))
    assert len(got) == len(starts) == 7
    for g, s0 in zip(got, starts):
        sl = slice(s0, s0 + win)
        close(g.numpy(), R.events_to_voxel_torch(x[sl], y[sl], t[sl], p[sl], B, sensor_size=(H, W), accum="f64"))
    tw = 0.13
    got = V.voxel_grids_fixed_t_torch(tx.cuda(), ty.cuda(), tt.cuda(), tp.cuda(), B, tw, sensor_size=(H, W))
    t_starts = np.arange(float(t[0]), float(t[-1]) - tw, tw)
    assert len(got) == len(t_starts) and got[0].is_cuda
    for g, ts0 in zip(got, t_starts):
        a, b = np.searchsorted(t, ts0), np.searchsorted(t, ts0 + tw)
        close(g.cpu().numpy(), R.events_to_voxel_torch(x[a:b], y[a:b], t[a:b], p[a:b], B, sensor_size=(H, W), accum="f64"))
    close(V.events_to_voxel_timesync_torch(tx, ty, tt, tp, B, 0.2, 0.5, sensor_size=(H, W)).numpy(),
          R.events_to_voxel_torch(*(a[np.searchsorted(t, 0.2):np.searchsorted(t, 0.5)] for a in (x, y, t, p)), B,
                                  sensor_size=(H, W), accum="f64"))
    vp, vn = V.events_to_neg_pos_voxel_torch(tx, ty, tt, tp, B, sensor_size=(H, W))
    close(vp.numpy(), R.events_to_voxel_torch(x, y, t, (p > 0).astype(np.float32), B, sensor_size=(H, W), accum="f64"))
    close(vn.numpy(), R.events_to_voxel_torch(x, y, t, (p <= 0).astype(np.float32), B, sensor_size=(H, W), accum="f64"))
    assert V.voxel_grids_fixed_n_torch(tx[:10], ty[:10], tt[:10], tp[:10], B, 10, sensor_size=(H, W)) == []
    # many short windows in bounded chunks (one launch holds <= 65535 windows and a bounded amount of memory)
    whole = V.voxel_grids_fixed_n_torch(tx, ty, tt, tp, B, 900, sensor_size=(H, W))
    monkeypatch.setattr(V, "_WINDOW_CHUNK_BYTES", 3 * B * H * W * 4)          # 3 windows per launch
    chunked = V.voxel_grids_fixed_n_torch(tx, ty, tt, tp, B, 900, sensor_size=(H, W))
    assert len(whole) == len(chunked) == 55        # (float atomics: the same sums, not necessarily the same bits)
    for a, b in zip(whole, chunked):
        close(a.numpy(), b.numpy())


def test_large_voxel_windows_take_the_one_pass_path(E, monkeypatch):
    """Windows of >= 350 k events are voxelised one by one by the partition + LDS-tile path (two launches per window instead of
    two global atomics per event); windows that start at any event index (fixed_t: unaligned slices) included."""
    from event_utils_amd import _lib
    from event_utils_amd.representations import voxel_grid as V
    rng = np.random.default_rng(12)
    n, H, W, B = 1_600_003, 180, 240, 5
    x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
    t = np.sort(rng.uniform(0, 1.0, n)).astype(np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    tt = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    calls = []
    real = _lib.call
    monkeypatch.setattr(_lib, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    win = 400_001                                             # odd: every window but the first starts off a 16-byte boundary
    got = V.voxel_grids_fixed_n_torch(*tt, B, win, sensor_size=(H, W))
    assert len(got) == 3 and calls.count("evk_voxel2_f32") == 3 and "evk_voxel_segments_f32" not in calls
    for k, g in enumerate(got):
        sl = slice(k * win, (k + 1) * win)
        close(g.cpu().numpy(), R.events_to_voxel_torch(x[sl], y[sl], t[sl], p[sl], B, sensor_size=(H, W), accum="f64"))
    del calls[:]
    got = V.voxel_grids_fixed_t_torch(*tt, B, 0.3, sensor_size=(H, W))
    t_starts = np.arange(float(t[0]), float(t[-1]) - 0.3, 0.3)
    assert len(got) == len(t_starts) == 3 and calls.count("evk_voxel2_f32") == 3
    for g, ts0 in zip(got, t_starts):
        a, b = np.searchsorted(t, ts0), np.searchsorted(t, ts0 + np.float32(0.3))
        close(g.cpu().numpy(), R.events_to_voxel_torch(x[a:b], y[a:b], t[a:b], p[a:b], B, sensor_size=(H, W), accum="f64"))
    del calls[:]
    small = V.voxel_grids_fixed_n_torch(*tt, B, 100_000, sensor_size=(H, W))     # short windows: one launch for all of them
    assert len(small) == 16 and "evk_voxel2_f32" not in calls and calls.count("evk_voxel_segments_f32") == 1
    E.check_errors()


# ------------------------------------------------------------------------------------------------ F11 next rows
def test_f11_gather_contrast_and_timestamp_images(E, golden):
    g = golden("f11_gather_timestamp")
    x, y, t, p = f64(g["xs"]), f64(g["ys"]), f64(g["ts"]), f64(g["ps"])
    r = E.get_iwe(g["params"], x, y, t, p, E.linvel_warp(), (180, 240), return_events=True, return_per_event_contrast=True)
    close(r[0], g["iwe"])
    assert r[1] is None and np.array_equal(r[2][0], g["ev_x"]) and np.array_equal(r[2][1], g["ev_y"])
    close(r[3], g["contrast"])
    w = E.image_to_event_weights(g["g_x"], g["g_y"], g["g_img"])
    assert w.dtype == np.float64 and np.array_equal(w, g["g_w"])       # pure gather: bit-exact
    xi, yi = f64(g["ti_x"]), f64(g["ti_y"])
    a, b = E.events_to_timestamp_image(xi, yi, g["ti_ts64"], p)
    assert a.dtype == np.float32
    close(a, g["ti_np_pos"]); close(b, g["ti_np_neg"])
    a, b = E.events_to_timestamp_image(xi, yi, g["ti_ts64"], p, padding=False, normalize_timestamps=False)
    close(a, g["ti_np_nopad_pos"]); close(b, g["ti_np_nopad_neg"])
    tt = [torch.from_numpy(v) for v in (g["ti_x"], g["ti_y"], g["ti_ts64"].astype(np.float32), p.astype(np.float32))]
    for rev in (False, True):
        a, b = E.events_to_timestamp_image_torch(*tt, timestamp_reverse=rev)
        assert a.dtype == torch.float32 and a.device.type == "cpu"
        close(a.numpy(), g["ti_t_pos_rev%d" % rev]); close(b.numpy(), g["ti_t_neg_rev%d" % rev])
    with pytest.raises(IndexError):
        E.image_to_event_weights(np.array([-500.0]), np.array([1.0]), g["g_img"])
    # a float64 image (upstream takes any numpy array; found by tools/fuzz_parity.py): its own values enter the products
    from oracle import reference_np as R
    rng = np.random.default_rng(3)
    img64 = rng.normal(size=(37, 53)); gx, gy = rng.uniform(0, 54, 5000), rng.uniform(0, 38, 5000)
    assert np.array_equal(E.image_to_event_weights(gx, gy, img64), R.image_to_event_weights(gx, gy, img64))
    imgi = rng.integers(-9, 9, (37, 53))
    assert np.array_equal(E.image_to_event_weights(gx, gy, imgi), R.image_to_event_weights(gx, gy, imgi))


def test_f12_other_objectives(E, golden):
    """sos / soe / moa / isoa / sosa / r1 (objectives.py:308-596): same IWE, different scalar reductions."""
    from event_utils_amd.contrast_max import objectives as O
    from event_utils_amd.events import DeviceEvents
    g8, g = golden("f8_objective"), golden("f12_other_objectives")
    ev = DeviceEvents.from_arrays(*(f64(g8[k]) for k in ("xs", "ys", "ts", "ps")))
    w = E.linvel_warp()
    objs = {"sos": O.sos_objective(), "soe": O.soe_objective(), "moa": O.moa_objective(), "isoa": O.isoa_objective(),
            "sosa": O.sosa_objective()}
    for name, ob in objs.items():
        k = 0
        for prm in g["params"]:
            for s in (None, 0.0):
                f = f64(ob.evaluate_function(prm, ev, None, None, None, w, (180, 240), blur_sigma=s))
                rf = g[name + "_f"][k]
                if name == "isoa":
                    assert abs(f - rf) <= 2, (name, f, rf)          # a count: pixels within 1e-7 of the threshold may flip
                else:
                    assert abs(f - rf) <= 2e-5 * abs(rf) + 1e-12, (name, k, f, rf)
                if ob.has_derivative:
                    gr = f64(ob.evaluate_gradient(prm, ev, None, None, None, w, (180, 240), blur_sigma=s))
                    rg = g[name + "_g"][k]
                    tol = 2e-5 * np.max(np.abs(rg)) + 1e-9
                    if name in ("soe", "sosa"):     # exp() weighting amplifies the 1e-7 summation-order noise of the IWE
                        tol = 2e-4 * np.max(np.abs(rg)) + 1e-9
                    if name == "isoa":
                        tol = 0.02 * np.max(np.abs(rg)) + 1e-3
                    assert np.max(np.abs(gr - rg)) <= tol, (name, k, gr, rg)
                k += 1
    r1 = O.r1_objective()
    P = g["params"]
    vals = [f64(r1.evaluate_function(q, ev, None, None, None, w, (180, 240))) for q in (P[0], P[1], P[1], P[2])]
    assert np.max(np.abs(np.array(vals) - g["r1_f"]) / np.abs(g["r1_f"])) <= 2e-5
    # these objectives now also work through optimize() (upstream several of them lack the base-class state)
    argmax = E.optimize(ev, None, None, None, w, O.sos_objective(), numeric_grads=False, img_size=(180, 240))
    assert np.all(np.isfinite(np.asarray(argmax, float)))


def test_rms_objective_of_a_handful_of_events(E):
    """objectives.py:266-306 on nearly rank-one images (one event, blurred): the spectral norm must come out as numpy's does
    (the device SVD does not converge there; the Gram matrix's largest eigenvalue does) -- found by tools/fuzz_parity.py."""
    from event_utils_amd.contrast_max import objectives as O
    from oracle import reference_np as R
    for n, sigma in ((1, 2.0), (2, 1.0), (5, 0.0)):
        rng = np.random.default_rng(n)
        x, y = rng.uniform(20, 200, n), rng.uniform(20, 150, n)
        t = np.sort(rng.uniform(0, 0.1, n)); p = np.ones(n)
        ro, eo = R.rms_objective(), O.rms_objective()
        ro.accum = "f64"
        rf = ro.evaluate_function(np.array([10., -5.]), x, y, t, p, R.linvel_warp(), (180, 240), blur_sigma=sigma)
        f = eo.evaluate_function(np.array([10., -5.]), x, y, t, p, E.linvel_warp(), (180, 240), blur_sigma=sigma)
        assert isinstance(f, np.float32) and abs(float(f) - float(rf)) <= 2e-5 * abs(float(rf)) + 1e-12, (n, f, rf)


def test_f13_dense_flow_warp(E, golden):
    from event_utils_amd.transforms.optic_flow import warp_events_flow_torch
    g = golden("f13_flow_warp")
    tx, ty, tt = (torch.from_numpy(g[k]) for k in ("xs", "ys", "ts"))
    xw, yw = warp_events_flow_torch(tx, ty, tt, torch.ones_like(tx), torch.from_numpy(g["flow"]))
    assert xw.dtype == torch.float32 and xw.device.type == "cpu"
    close(xw.numpy(), g["xw"], 1e-6); close(yw.numpy(), g["yw"], 1e-6)
    xw, yw = warp_events_flow_torch(tx.cuda(), ty.cuda(), tt.cuda(), None, torch.from_numpy(g["flow"]).cuda(), t0=0.02)
    close(xw.cpu().numpy(), g["xw_t0"], 1e-6); close(yw.cpu().numpy(), g["yw_t0"], 1e-6)
    # motion compensation = dense-flow warp + bilinear event image (draw_flow.py:15-21)
    img = E.events_to_image_torch(xw.cpu(), yw.cpu(), torch.ones_like(tx), sensor_size=(60, 80), interpolation='bilinear')
    xo, yo = R.warp_events_flow_torch(g["xs"], g["ys"], g["ts"], None, g["flow"], t0=0.02)
    ref = R.events_to_image_torch(xw.cpu().numpy(), yw.cpu().numpy(), np.ones(len(xo), np.float32), sensor_size=(60, 80),
                                  interpolation='bilinear', accum="f64")
    close(img.numpy(), ref)


def test_motion_compensate_is_the_composition_of_its_two_kernels(E, golden):
    """draw_flow.py:15-26 without the cv2 I/O: dense-flow warp (pinned by f13) + bilinear event image (pinned by f4) +
    flip, min-max normalisation and crop, against the same composition of the oracle's functions."""
    from event_utils_amd.lib.visualization.draw_flow import motion_compensate
    g = golden("f13_flow_warp")
    xs, ys, ts, flow = g["xs"], g["ys"], g["ts"], g["flow"]
    ps = np.where(np.arange(len(xs)) % 3 == 0, -1.0, 1.0).astype(np.float32)
    img = motion_compensate(xs, ys, ts, ps, flow, crop=(2, 40, 3, 50))
    xw, yw = R.warp_events_flow_torch(xs, ys, ts, None, flow)
    H, W = flow.shape[-2:]
    ref = R.events_to_image_torch(xw.astype(np.float32), yw.astype(np.float32), ps, sensor_size=(H, W), interpolation='bilinear')
    ref = np.flip(np.flip(ref, axis=0), axis=1)
    ref = ((ref - ref.min()) * (255.0 / (ref.max() - ref.min())))[2:40, 3:50]
    assert img.dtype == np.float32 and img.shape == ref.shape
    assert np.abs(img - ref).max() <= 1e-4 * 255.0


@pytest.mark.parametrize("impl", ["auto", "tiled"])
def test_f16_windowed_and_split_voxel_functions(E, golden, monkeypatch, impl):
    """voxel_grids_fixed_n_torch / voxel_grids_fixed_t_torch / events_to_voxel_timesync_torch /
    events_to_neg_pos_voxel[_torch] against what the REAL reference returned for the same stream (f16)."""
    from event_utils_amd.representations import voxel_grid as V
    monkeypatch.setenv("EVK_IMPL", impl)
    g = golden("f16_voxel_windows")
    ss, B = tuple(int(v) for v in g["sensor_size"]), int(g["B"])
    x, y, t, p = g["xs"], g["ys"], g["ts"], g["ps"]
    tx, ty, tt, tp = (torch.from_numpy(a).cuda() for a in (x, y, t, p))
    for key, n in (("fixed_n", int(g["fixed_n_n"])), ("fixed_n_div", int(g["fixed_n_div_n"]))):
        got = V.voxel_grids_fixed_n_torch(tx, ty, tt, tp, B, n, sensor_size=ss)
        assert len(got) == len(g[key])
        close(torch.stack(got).cpu().numpy(), g[key])
    got = V.voxel_grids_fixed_t_torch(tx, ty, tt, tp, B, float(g["fixed_t_t"]), sensor_size=ss)
    assert len(got) == len(g["fixed_t"])
    close(torch.stack(got).cpu().numpy(), g["fixed_t"])
    close(V.events_to_voxel_timesync_torch(tx, ty, tt, tp, B, 0.2, 0.5, sensor_size=ss).cpu().numpy(), g["timesync"])
    vp, vn = V.events_to_neg_pos_voxel_torch(tx, ty, tt, tp, B, sensor_size=ss)
    close(vp.cpu().numpy(), g["neg_pos_torch_pos"]); close(vn.cpu().numpy(), g["neg_pos_torch_neg"])
    vp, vn = V.events_to_neg_pos_voxel(x.astype(np.int64), y.astype(np.int64), t.astype(np.float64), p, B, sensor_size=ss)
    assert vp.dtype == np.float64
    close(vp, g["neg_pos_numpy_pos"], 1e-12); close(vn, g["neg_pos_numpy_neg"], 1e-12)


@pytest.mark.parametrize("mode", ["analytic", "numeric"])
def test_f9_evk_bfgs_reaches_the_reference_optimum(E, golden, mode):
    """optimize_contrast(optimizer='evk_bfgs') -- two event passes per iteration instead of scipy's ~12 -- on the golden
    scene of F9: the analytic mode ends at the reference's final argmax (fixture `analytic_argmax`, produced by the real
    reference with scipy's fmin_bfgs) and at scipy's on this path, with a third of the passes; with the reference's default
    numeric gradients (forward differences, epsilon = 1: a biased gradient whose zero is not the optimum) the end point must
    be at least as good an optimum as the reference's."""
    from event_utils_amd.contrast_max.events_cmax import optimize_contrast
    g8, g = golden("f8_objective"), golden("f9_optimize_trace")
    x, y, t, p = f64(g8["xs"]), f64(g8["ys"]), f64(g8["ts"]), f64(g8["ps"])
    size, w = tuple(g8["img_size"]), E.linvel_warp()
    numeric = mode == "numeric"
    res = {}
    for optimizer in ("evk_bfgs", "scipy"):
        obj = E.variance_objective()
        # (analytic mode: the CONSISTENT gradient -- the reference-exact one is not the gradient of the function, Q5, and a
        # line search that trusts it zigzags; scipy is given the same objective)
        obj.reference_exact = numeric
        passes = [0]
        for name in ("evaluate_function", "evaluate_gradient", "evaluate_function_and_gradient",
                     "evaluate_function_and_numeric_gradient", "evaluate_numeric_gradient"):
            fn = getattr(obj, name)

            def wrapped(*a, _fn=fn, **k):
                passes[0] += 1
                return _fn(*a, **k)
            setattr(obj, name, wrapped)
        fb = obj.evaluate_function_batch

        def fbw(*a, _fb=fb, _o=obj, **k):
            r = _fb(*a, **k)
            passes[0] += _o.batch_passes
            return r
        obj.evaluate_function_batch = fbw
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            kw = {"optimizer": "evk_bfgs"} if optimizer == "evk_bfgs" else {}
            a = optimize_contrast(x, y, t, p, w, obj, numeric_grads=numeric, blur_sigma=1.0, img_size=size, **kw)
        res[optimizer] = (np.asarray(a, dtype=np.float64), passes[0])
    plain = E.variance_objective()
    f_at = lambda q: float(plain.evaluate_function(np.asarray(q, float), x, y, t, p, w, size, 1.0))  # noqa: E731
    a, npass = res["evk_bfgs"]
    if not numeric:
        assert np.linalg.norm(a - g["analytic_argmax"]) < 0.5          # the reference's own end point (scipy, CPU)
        assert np.linalg.norm(a - res["scipy"][0]) < 0.5
    assert f_at(a) <= f_at(g[mode + "_argmax"]) + 1e-4 * abs(f_at(g[mode + "_argmax"]))
    with pytest.raises(ValueError):
        optimize_contrast(x, y, t, p, w, E.variance_objective(), optimizer="nelder-mead", img_size=size)
    if not numeric:
        # the tile-bucketed regime (600 k events; the golden scene's 50 k run the direct kernels, whose float32 global atomics
        # are noisier and where a three-point pass is three passes): same optimum with fewer event passes than scipy
        import bench
        xs, ys, ts, ps = bench.structured_scene(3, 600_000, 260, 346)
        ev = E.DeviceEvents.from_arrays(xs, ys, ts, ps, precision="f32")
        out = {}
        for optimizer in ("evk_bfgs", "scipy"):
            obj = E.variance_objective()
            obj.sensor_size, obj.reference_exact = (260, 346), False
            n_pass = [0]
            fg0, fb0 = obj.evaluate_function_and_gradient, obj.evaluate_function_batch

            def fgw(*a_, _f=fg0, **k):
                n_pass[0] += 1
                return _f(*a_, **k)

            def fbw2(*a_, _f=fb0, _o=obj, **k):
                r = _f(*a_, **k)
                n_pass[0] += _o.batch_passes
                return r
            obj.evaluate_function_and_gradient, obj.evaluate_function_batch = fgw, fbw2
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                kw = {"optimizer": "evk_bfgs"} if optimizer == "evk_bfgs" else {}
                out[optimizer] = (np.asarray(optimize_contrast(ev, None, None, None, w, obj, numeric_grads=False, blur_sigma=1.0,
                                                               img_size=(260, 346), **kw), dtype=np.float64), n_pass[0])
        assert np.linalg.norm(out["evk_bfgs"][0] - np.array([40., -25.])) < 0.05
        assert np.linalg.norm(out["evk_bfgs"][0] - out["scipy"][0]) < 0.05
        assert out["evk_bfgs"][1] < out["scipy"][1], out


def test_f17_timestamp_image_and_event_image_classes(golden):
    """TimestampImage / EventImage (image.py:355-396) on the GPU against the fixture the real reference produced: the
    last-writer-wins images bit for bit (float64 time stamps), the dense-rank and min-max normalised get_image() outputs bit
    for bit, upstream's no-op add_events; then 3 M events on a 480x640 image against the oracle (one atomicMax per event),
    an out-of-range event raising IndexError, and the opt-in polarity accumulation."""
    import event_utils_amd as E
    from test_oracle_golden import _drive_image_classes
    g = golden("f17_image_classes")
    got = _drive_image_classes(E, g)
    for k, v in got.items():
        assert v.dtype == np.float64 and np.array_equal(v, g[k], equal_nan=True), k
    rng = np.random.default_rng(171)
    n, H, W = 3_000_000, 480, 640
    xs = rng.uniform(-3, W, n); ys = rng.uniform(-2, H, n)
    ts = np.sort(rng.uniform(0, 1, n)); ps = rng.integers(0, 2, n) * 2.0 - 1.0
    a, b = E.TimestampImage((H, W)), R.TimestampImage((H, W))
    for obj in (a, b):
        obj.set_init(-0.5)
        obj.add_events(xs, ys, ts, ps)
    assert np.array_equal(a.image, b.image) and np.array_equal(a.get_image(), b.get_image())
    a.add_events(torch.from_numpy(xs[:1000]).cuda(), torch.from_numpy(ys[:1000]).cuda(), torch.from_numpy(ts[:1000] + 5).cuda(), None)
    b.add_events(xs[:1000], ys[:1000], ts[:1000] + 5, None)
    assert np.array_equal(a.image, b.image)
    with pytest.raises(IndexError):
        a.add_events([1.0, float(W)], [1.0, 1.0], [0.1, 0.2], None)
    e1, e2 = E.EventImage((H, W)), R.EventImage((H, W))
    for obj in (e1, e2):
        obj.add_events(xs, ys, ts, ps, use_polarity=True)
    assert np.array_equal(e1.image, e2.image) and np.array_equal(e1.get_image(), e2.get_image())
    with pytest.raises(IndexError):
        e1.add_events([1.0], [-float(H) - 1.0], [0.1], [1.0])
    e1.image = np.zeros((H, W)); e1.add_event(2.9, 3.1, 0.0, 4.0)
    assert e1.image[3, 2] == 4.0 and e1.image.sum() == 4.0
    assert np.all(np.isnan(E.EventImage((4, 5)).get_image()))


def test_device_searchsorted_equals_numpy(E):
    """evk_searchsorted_left (the window bounds of events_to_voxel_timesync_torch / voxel_grids_fixed_t_torch on a device time
    column, voxel_grid.py:104-105) against np.searchsorted: float32 and float64 columns, repeated values, keys below / above /
    between / equal, keys that are not float32 values, a NaN key, an empty column."""
    from event_utils_amd.representations import voxel_grid as V
    rng = np.random.default_rng(77)
    for dtype in (np.float32, np.float64):
        a = np.sort(rng.uniform(10.0, 11.0, 200_001)).astype(dtype)
        a[5000:5100] = a[5000]                                   # a run of equal values
        keys = np.concatenate([rng.uniform(9.5, 11.5, 4000), a[rng.integers(0, len(a), 500)].astype(np.float64),
                               [a[0], a[-1], -np.inf, np.inf, np.nan, float(a[5000])]])
        got = V._searchsorted_device(torch.from_numpy(a).cuda(), keys)
        assert got.dtype == np.int64 and np.array_equal(got, np.searchsorted(a, keys))
    assert np.array_equal(V._searchsorted_device(torch.empty(0, device="cuda"), [0.5, 2.0]), [0, 0])
    # the two public callers on a device column against their host-column selves
    n, H, W, B = 300_000, 60, 80, 3
    x = torch.from_numpy(rng.integers(0, W, n).astype(np.float32)); y = torch.from_numpy(rng.integers(0, H, n).astype(np.float32))
    t = torch.from_numpy(np.sort(rng.uniform(0, 1, n)).astype(np.float32)); p = torch.from_numpy(rng.choice([-1.0, 1.0], n).astype(np.float32))
    host = V.events_to_voxel_timesync_torch(x, y, t, p, B, 0.25, 0.7500001, sensor_size=(H, W))
    devc = V.events_to_voxel_timesync_torch(x.cuda(), y.cuda(), t.cuda(), p.cuda(), B, 0.25, 0.7500001, sensor_size=(H, W))
    assert devc.is_cuda
    close(devc.cpu().numpy(), host.numpy())          # (same events; a slice's alignment decides the kernel family, not the values)
    a = V.voxel_grids_fixed_t_torch(x, y, t, p, B, 0.13, sensor_size=(H, W))
    b = V.voxel_grids_fixed_t_torch(x.cuda(), y.cuda(), t.cuda(), p.cuda(), B, 0.13, sensor_size=(H, W))
    assert len(a) == len(b) == 7
    for u, v in zip(a, b):
        close(v.cpu().numpy(), u.numpy())
