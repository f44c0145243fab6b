"""GPU (-m gpu): the event images on the one-pass partition + LDS-tile design (evk_image2.hip) against the oracle --
events_to_image (integer image, bit-exact) and events_to_image_torch (nearest / bilinear, 1e-5 of the image's maximum) at
the full 10 M-event size of BASELINE.json configs[1]'s sensor, on structured scenes whose hot tiles are cut, and on the
inputs only the partition kernel's rare path can take (negative pixels that wrap, out-of-range ones that raise, weights
that need all 32 bits).  The golden vectors f1 / f4 and the 16 random option-space variants run with this path forced in
test_gpu_parity.py / test_gpu_stress.py (impl = "tiled")."""
import numpy as np
import pytest
import torch

from oracle import reference_np as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    import event_utils_amd as E
    from event_utils_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available()
    return E


@pytest.fixture(autouse=True)
def _tiled(monkeypatch):
    monkeypatch.setenv("EVK_IMPL", "tiled")


def close(a, ref, tol=1e-5):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    assert a.shape == ref.shape
    assert np.max(np.abs(a - ref)) <= tol * max(np.max(np.abs(ref)), 1e-30), np.max(np.abs(a - ref))


def test_integer_image_10m_events_bit_exact(E):
    """10 M events on the 640x480 sensor: +-1 polarities, the count image, and integer weights that do not fit the record's
    21 bits (side array) -- all equal to np.bincount bit for bit."""
    rng = np.random.default_rng(11)
    n, H, W = 10_000_000, 480, 640
    xs, ys = rng.integers(0, W + 1, n), rng.integers(0, H + 1, n)      # the (H+1, W+1) canvas: x == W, y == H are legal
    ps = rng.integers(0, 2, n) * 2 - 1
    assert np.array_equal(E.events_to_image(xs, ys, ps, sensor_size=(H, W)), R.events_to_image(xs, ys, ps, sensor_size=(H, W)))
    cnt = E.events_to_image(xs, ys, np.ones_like(ps), sensor_size=(H, W), meanval=False)
    assert np.array_equal(cnt, R.events_to_image(xs, ys, np.ones_like(ps), sensor_size=(H, W)))
    m = 1_000_000
    big = rng.integers(-2_000, 2_000, m)
    big[::7] = rng.integers(-(2 ** 20) - 5, 2 ** 20 + 5, big[::7].shape[0])       # around the 21-bit limit, both sides
    assert np.array_equal(E.events_to_image(xs[:m], ys[:m], big, sensor_size=(H, W)),
                          R.events_to_image(xs[:m], ys[:m], big, sensor_size=(H, W)))
    a = E.events_to_image(xs[:m], ys[:m], ps[:m], sensor_size=(H, W), meanval=True, default=3)
    assert np.array_equal(a, R.events_to_image(xs[:m], ys[:m], ps[:m], sensor_size=(H, W), meanval=True, default=3))


@pytest.mark.parametrize("weights", ["unit", "float"])
def test_float_images_10m_events_against_the_oracle(E, weights):
    rng = np.random.default_rng(12)
    n, H, W = 10_000_000, 480, 640
    x = rng.uniform(0, W - 1, n).astype(np.float32); y = rng.uniform(0, H - 1, n).astype(np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32) if weights == "unit" else rng.normal(size=n).astype(np.float32)
    xd, yd, pd = (torch.from_numpy(a).cuda() for a in (x, y, p))
    for kw in (dict(interpolation=None, padding=False), dict(interpolation='bilinear', padding=True),
               dict(interpolation='bilinear', padding=False, clip_out_of_range=True)):
        ref = R.events_to_image_torch(x, y, p, sensor_size=(H, W), accum="f64", **kw)
        got = E.events_to_image_torch(xd, yd, pd, sensor_size=(H, W), **kw)
        assert got.is_cuda and got.dtype == torch.float32
        # (float weights: sums of ~30 N(0, 1) terms per pixel; the bar is relative to the image's maximum)
        close(got.cpu().numpy(), ref)
    # unit weights are counted in integers: bit-reproducible from run to run, nearest and bilinear
    if weights == "unit":
        for kw in (dict(interpolation=None, padding=False), dict(interpolation='bilinear', padding=True)):
            a = E.events_to_image_torch(xd, yd, pd, sensor_size=(H, W), **kw)
            b = E.events_to_image_torch(xd, yd, pd, sensor_size=(H, W), **kw)
            if kw["interpolation"] is None:
                assert torch.equal(a, b)
            else:       # (the window rings meet in float32 global atomics: equal up to their order)
                close(a.cpu().numpy(), b.cpu().numpy(), 1e-6)


@pytest.mark.parametrize("scene", ["blob", "one_pixel", "edges"])
def test_structured_scenes_cut_hot_tiles(E, scene):
    """Half of the events in a 40x30 patch / on ONE pixel / on two moving edges: hot tiles are cut into pieces whose partial
    tiles the last piece sums; segments are long (streamed by whole waves)."""
    rng = np.random.default_rng(13)
    n, H, W = 3_000_000, 480, 640
    x = rng.uniform(0, W - 1, n).astype(np.float32); y = rng.uniform(0, H - 1, n).astype(np.float32)
    if scene == "blob":
        hot = rng.random(n) < 0.5
        x[hot] = rng.uniform(300, 340, hot.sum()).astype(np.float32); y[hot] = rng.uniform(200, 230, hot.sum()).astype(np.float32)
    elif scene == "one_pixel":
        hot = rng.random(n) < 0.5
        x[hot] = 123.25; y[hot] = 77.5
    else:
        t = np.linspace(0, 1, n)
        hot = rng.random(n) < 0.8
        x[hot] = (100 + 400 * t[hot] + rng.normal(0, 0.7, hot.sum())).astype(np.float32)
        x = np.clip(x, 0, W - 1.001).astype(np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    xd, yd, pd = (torch.from_numpy(a).cuda() for a in (x, y, p))
    for kw in (dict(interpolation=None, padding=False), dict(interpolation='bilinear', padding=True)):
        ref = R.events_to_image_torch(x, y, p, sensor_size=(H, W), accum="f64", **kw)
        close(E.events_to_image_torch(xd, yd, pd, sensor_size=(H, W), **kw).cpu().numpy(), ref)
        w = rng.normal(size=n).astype(np.float32)            # float64 accumulators
        ref = R.events_to_image_torch(x, y, w, sensor_size=(H, W), accum="f64", **kw)
        mag = R.events_to_image_torch(x, y, np.abs(w), sensor_size=(H, W), accum="f64", **kw)
        got = E.events_to_image_torch(xd, yd, torch.from_numpy(w).cuda(), sensor_size=(H, W), **kw).cpu().numpy()
        assert np.max(np.abs(got.astype(np.float64) - ref)) <= 1e-5 * np.max(np.abs(ref)) + 2e-7 * np.max(mag)
    xi, yi = x.astype(np.int64), y.astype(np.int64)
    pi = p.astype(np.int64) * 3
    assert np.array_equal(E.events_to_image(xi, yi, pi, sensor_size=(H, W)), R.events_to_image(xi, yi, pi, sensor_size=(H, W)))


def test_rare_events_take_the_direct_kernels_route(E):
    """Bilinear: pixels at -1 wrap to the last column / row (index_put_), masked events land on (0, 0) with weight 0,
    events whose right / lower neighbour is outside raise; nearest: clipped events pile up at (0, 0) WITH their weight (Q8),
    negative indices wrap.  A few thousand of them among 400 k ordinary events."""
    rng = np.random.default_rng(14)
    n, H, W = 400_000, 120, 160
    x = rng.uniform(0, W - 1, n).astype(np.float32); y = rng.uniform(0, H - 1, n).astype(np.float32)
    p = rng.normal(size=n).astype(np.float32)
    x[:3000] = rng.uniform(-1, 0, 3000).astype(np.float32)          # px = -1 -> wraps to W (padded image: W + 1 columns)
    y[3000:5000] = rng.uniform(-1, 0, 2000).astype(np.float32)
    x[5000:9000] = rng.uniform(W, W + 5, 4000).astype(np.float32)   # beyond clipx: masked
    y[9000:9500] = np.float32(H + 0.5)
    for padding in (True, False):
        kw = dict(interpolation='bilinear', padding=padding)
        ref = R.events_to_image_torch(x, y, p, sensor_size=(H, W), accum="f64", **kw)
        got = E.events_to_image_torch(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(p), sensor_size=(H, W), **kw)
        close(got.numpy(), ref)
    for padding in (True, False):
        kw = dict(interpolation=None, padding=padding)
        ref = R.events_to_image_torch(x, y, p, sensor_size=(H, W), accum="f64", **kw)
        mag = R.events_to_image_torch(x, y, np.abs(p), sensor_size=(H, W), accum="f64", **kw)
        got = E.events_to_image_torch(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(p), sensor_size=(H, W), **kw)
        assert np.max(np.abs(got.numpy().astype(np.float64) - ref)) <= 1e-5 * np.max(np.abs(ref)) + 2e-7 * np.max(mag)
    # a MASKED event whose other coordinate is NaN: its indices are x.long() * 0 = 0 (image.py:93-95) -- pixel (0, 0), with
    # its weight, no error; both kernel families
    import os
    import warnings
    xn, yn = x.copy(), y.copy()
    xn[9500:9510] = np.float32(W + 1); yn[9500:9510] = np.nan
    yn[9510:9520] = np.float32(H + 1); xn[9510:9520] = np.nan
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # (numpy's NaN -> int64 cast warns; the value is INT64_MIN as in torch)
        ref = R.events_to_image_torch(xn, yn, p, sensor_size=(H, W), accum="f64", interpolation=None, padding=False)
    mag = R.events_to_image_torch(x, y, np.abs(p), sensor_size=(H, W), accum="f64", interpolation=None, padding=False)
    for impl in ("tiled", "direct"):
        os.environ["EVK_IMPL"] = impl
        try:
            got = E.events_to_image_torch(torch.from_numpy(xn), torch.from_numpy(yn), torch.from_numpy(p), sensor_size=(H, W),
                                          interpolation=None, padding=False)
        finally:
            os.environ["EVK_IMPL"] = "tiled"
        assert np.max(np.abs(got.numpy().astype(np.float64) - ref)) <= 1e-5 * np.max(np.abs(ref)) + 2e-7 * np.max(mag), impl
    # without clipping the out-of-range events raise, as index_put_ does -- on both kernel families, with the same count
    xs, ys, ps = (torch.from_numpy(a) for a in (x, y, p))
    msgs = []
    for kw in (dict(interpolation='bilinear', padding=False), dict(interpolation=None, padding=False)):
        with pytest.raises(IndexError) as ei:
            E.events_to_image_torch(xs, ys, ps, sensor_size=(H, W), clip_out_of_range=False, **kw)
        msgs.append(str(ei.value))
    os.environ["EVK_IMPL"] = "direct"
    try:
        for i, kw in enumerate((dict(interpolation='bilinear', padding=False), dict(interpolation=None, padding=False))):
            with pytest.raises(IndexError) as ei:
                E.events_to_image_torch(xs, ys, ps, sensor_size=(H, W), clip_out_of_range=False, **kw)
            assert str(ei.value) == msgs[i]
    finally:
        os.environ["EVK_IMPL"] = "tiled"
    # a NaN weight poisons exactly the pixels the reference poisons
    q = p.copy(); q[20_000] = np.nan
    ref = R.events_to_image_torch(x, y, q, sensor_size=(H, W), accum="f64", interpolation='bilinear', padding=True)
    got = E.events_to_image_torch(xs, ys, torch.from_numpy(q), sensor_size=(H, W), interpolation='bilinear', padding=True).numpy()
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.isnan(ref).sum() == 4


def test_device_resident_image_calls_defer_their_error(E, monkeypatch):
    """Events and image on the device: the call only enqueues; an IndexError surfaces at the next call on the stream or at
    check_errors() (EVK_ERRORS=strict raises before returning)."""
    from event_utils_amd import _device as D
    n, H, W = 300_000, 100, 100
    rng = np.random.default_rng(15)
    x = torch.from_numpy(rng.uniform(0, W - 1, n).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.uniform(0, H - 1, n).astype(np.float32)).cuda()
    p = torch.ones(n, device="cuda")
    bad = x.clone(); bad[5] = 5000.0
    monkeypatch.setenv("EVK_ERRORS", "deferred")
    E.events_to_image_torch(bad, y, p, sensor_size=(H, W), clip_out_of_range=False, padding=False)   # enqueues
    with pytest.raises(IndexError):
        D.check_errors()
    good = E.events_to_image_torch(x, y, p, sensor_size=(H, W), padding=False)
    D.check_errors()
    assert abs(good.double().sum().item() - n) < 1e-3
    monkeypatch.setenv("EVK_ERRORS", "strict")
    with pytest.raises(IndexError):
        E.events_to_image_torch(bad, y, p, sensor_size=(H, W), clip_out_of_range=False, padding=False)


def test_odd_sizes_and_tiny_images(E):
    """Images smaller than a tile, one-pixel-wide images (bilinear falls back to the direct kernel), event counts that are
    not multiples of four."""
    rng = np.random.default_rng(16)
    for (H, W, n) in ((5, 7, 1001), (1, 50, 777), (33, 2, 4099), (9, 1000, 65_537), (300, 17, 123_457)):
        x = rng.uniform(0, max(W - 1, 0.5), n).astype(np.float32); y = rng.uniform(0, max(H - 1, 0.5), n).astype(np.float32)
        p = rng.normal(size=n).astype(np.float32)
        for kw in (dict(interpolation=None, padding=False), dict(interpolation='bilinear', padding=True)):
            ref = R.events_to_image_torch(x, y, p, sensor_size=(H, W), accum="f64", **kw)
            mag = R.events_to_image_torch(x, y, np.abs(p), sensor_size=(H, W), accum="f64", **kw)
            got = E.events_to_image_torch(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(p), sensor_size=(H, W), **kw)
            assert np.max(np.abs(got.numpy().astype(np.float64) - ref)) <= 1e-5 * np.max(np.abs(ref)) + 2e-7 * np.max(mag)
        xi, yi, pi = rng.integers(0, W + 1, n), rng.integers(0, H + 1, n), rng.integers(-9, 10, n)
        assert np.array_equal(E.events_to_image(xi, yi, pi, sensor_size=(H, W)), R.events_to_image(xi, yi, pi, sensor_size=(H, W)))


def test_images_at_50m_events_720p(E):
    """One rank's share of configs[4] as an image: 50 M events, 1280x720 -- 1020 tiles, sub-chunks of 12 K events (the
    partition geometry above 680 tiles), 6104 segments per tile.  Integer image bit-exact, float32 nearest and bilinear
    against the float64 oracle."""
    rng = np.random.default_rng(17)
    n, H, W = 50_000_000, 720, 1280
    x = rng.uniform(0, W - 1, n).astype(np.float32); y = rng.uniform(0, H - 1, n).astype(np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    xi, yi, pi = x.astype(np.int64), y.astype(np.int64), p.astype(np.int64)
    assert np.array_equal(E.events_to_image(xi, yi, pi, sensor_size=(H, W)), R.events_to_image(xi, yi, pi, sensor_size=(H, W)))
    del xi, yi, pi
    xd, yd, pd = (torch.from_numpy(a).cuda() for a in (x, y, p))
    for kw in (dict(interpolation=None, padding=False), dict(interpolation='bilinear', padding=True)):
        ref = R.events_to_image_torch(x, y, p, sensor_size=(H, W), accum="f64", **kw)
        close(E.events_to_image_torch(xd, yd, pd, sensor_size=(H, W), **kw).cpu().numpy(), ref)


# ---- average-timestamp images on the one-pass path (round 6: evk_timestamp_images2_f32, image.py:219-353) --------------------
def _ts_planes(E, x, y, t, p, img_size, mode, ta, tdiv, impl, clip=True, fixed=True):
    """The four raw planes (time+ | count+ | time- | count-; the counts start at ONE) through one kernel family."""
    import os
    from event_utils_amd import tiled
    from event_utils_amd.representations import image as I
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    old, tiled.FORCE["image_fixed"] = tiled.FORCE["image_fixed"], fixed
    os.environ["EVK_IMPL"] = impl
    try:
        out = I._timestamp_images_device(*cols, img_size, clip, 'bilinear', True, mode, ta, tdiv)
    finally:
        os.environ["EVK_IMPL"] = "tiled"
        tiled.FORCE["image_fixed"] = old
    return out.cpu().numpy().astype(np.float64)


def _ts_planes_oracle(x, y, nts, p, img_size, clip=True):
    """The same four planes from the oracle's splat (float64 accumulation), before the division."""
    mask = np.ones(x.shape, np.float32)
    if clip:
        mask = np.where(x >= img_size[1] - 1, np.float32(0), np.float32(1)) * np.where(y >= img_size[0] - 1, np.float32(0), np.float32(1))
    pxs, pys = np.floor(x), np.floor(y)
    dxs, dys = (x - pxs).astype(np.float32), (y - pys).astype(np.float32)
    pxs, pys = (pxs * mask).astype(np.int64), (pys * mask).astype(np.int64)
    pos, neg = (p > 0).astype(np.float32), (p <= 0).astype(np.float32)
    out = []
    for wts, init in ((nts * pos, 0), (pos, 1), (nts * neg, 0), (neg, 1)):
        img = np.full(img_size, init, dtype=np.float32)
        R.interpolate_to_image(pxs, pys, dxs, dys, wts.astype(np.float32), img, "f64")
        out.append(img.astype(np.float64))
    return np.stack(out)


@pytest.mark.parametrize("scene", ["uniform", "blob", "edges"])
def test_timestamp_images_one_pass_against_direct_kernel_and_oracle(E, scene):
    """2 M events at 640x480 (+ padding): the four planes of the one-pass path (fixed-point and float64 counts) against the
    direct global-atomic kernel and the oracle's float64 splat, for the three time modes; clipped events (they land on pixel
    (0, 0) WITH their weights, upstream quirk), negative pixels that wrap (rare path), NaN polarities (in neither class)."""
    from event_utils_amd import _lib
    rng = np.random.default_rng(31)
    n, H, W = 2_000_000, 480, 640
    img_size = (H + 1, W + 1)
    x = rng.uniform(0, W, n).astype(np.float32); y = rng.uniform(0, H, n).astype(np.float32)
    if scene == "blob":
        hot = rng.random(n) < 0.5
        x[hot] = rng.uniform(300, 340, hot.sum()).astype(np.float32); y[hot] = rng.uniform(200, 230, hot.sum()).astype(np.float32)
    elif scene == "edges":
        u = np.linspace(0, 1, n)
        hot = rng.random(n) < 0.8
        x[hot] = (100 + 400 * u[hot] + rng.normal(0, 0.7, hot.sum())).astype(np.float32)
    t = np.sort(rng.uniform(0.0, 0.1, n)).astype(np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    p[::1000] = 0.0                                                  # non-positive class
    p[7::5000] = np.nan                                              # neither class
    x[:2000] = rng.uniform(-1, 0, 2000).astype(np.float32)           # px = -1 wraps to the last column
    y[2000:3000] = rng.uniform(-1, 0, 1000).astype(np.float32)
    x[3000:6000] = rng.uniform(W, W + 7, 3000).astype(np.float32)    # clipped: pixel (0, 0), weights intact
    y[6000:6500] = np.float32(H + 0.25)
    calls = []
    orig = _lib.call
    _lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
    try:
        t_first, t_last = np.float32(t[0]), np.float32(t[-1])
        tdiv = np.float32(np.float32(t_last - t_first) + np.float32(1e-6))
        for mode, ta, td_, nts in ((0, t_first, tdiv, (t - t_first) / tdiv), (1, t_last, tdiv, (-t + t_last) / tdiv), (2, 0.0, 1.0, t)):
            ref = _ts_planes_oracle(x, y, nts.astype(np.float32), p, img_size)
            direct = _ts_planes(E, x, y, t, p, img_size, mode, ta, td_, "direct")
            assert calls[-1] == "evk_timestamp_images_f32"
            for fixed in (True, False):
                fast = _ts_planes(E, x, y, t, p, img_size, mode, ta, td_, "tiled", fixed=fixed)
                assert calls[-1] == "evk_timestamp_images2_f32"
                for k in range(4):
                    close(fast[k], ref[k])
                    close(fast[k], direct[k])
    finally:
        _lib.call = orig
    # the public functions on this stream (device tensors in, device tensors out) against the oracle's
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    for rev in (False, True):
        a, b = E.events_to_timestamp_image_torch(*cols, sensor_size=(H, W), timestamp_reverse=rev)
        ra, rb = R.events_to_timestamp_image_torch(x, y, t, p, sensor_size=(H, W), timestamp_reverse=rev, accum="f64")
        assert a.is_cuda and a.dtype == torch.float32
        close(a.cpu().numpy(), ra); close(b.cpu().numpy(), rb)


def test_timestamp_images_one_pass_errors_small_images_and_unsorted_time(E):
    """Without clipping an out-of-range event raises IndexError on both kernel families; images smaller than a tile, sizes that
    are not multiples of anything, unsorted time stamps (negative normalised times) and two events."""
    import os
    rng = np.random.default_rng(32)
    for n, H, W in ((60_000, 37, 53), (200_000, 180, 240), (2, 20, 20), (130_001, 719, 1279)):
        x = rng.uniform(0, W, n).astype(np.float32); y = rng.uniform(0, H, n).astype(np.float32)
        t = rng.uniform(0.0, 1.0, n).astype(np.float32)                  # NOT sorted: (t - t[0]) / (t[-1] - t[0] + eps) of any sign
        p = rng.normal(size=n).astype(np.float32)
        cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
        ra, rb = R.events_to_timestamp_image_torch(x, y, t, p, sensor_size=(H, W), accum="f64")
        a, b = E.events_to_timestamp_image_torch(*cols, sensor_size=(H, W))
        # (a ratio of two sums: where the count is tiny the quotient amplifies the 1e-7 relative error of the sums)
        scale = max(np.max(np.abs(ra)), np.max(np.abs(rb)), 1e-30)
        assert np.max(np.abs(a.cpu().numpy() - ra)) <= 2e-5 * scale and np.max(np.abs(b.cpu().numpy() - rb)) <= 2e-5 * scale
    x[5] = np.float32(W + 3)
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    msgs = []
    for impl in ("tiled", "direct"):
        os.environ["EVK_IMPL"] = impl
        try:
            with pytest.raises(IndexError) as ei:
                E.events_to_timestamp_image_torch(*cols, sensor_size=(H, W), clip_out_of_range=False)
            msgs.append(str(ei.value))
        finally:
            os.environ["EVK_IMPL"] = "tiled"
    assert msgs[0] == msgs[1]
    a, b = E.events_to_timestamp_image_torch(*cols, sensor_size=(H, W))           # the stream works on after the error
    assert torch.isfinite(a).all() and torch.isfinite(b).all()


# ---- interpolate_to_image on caller-computed pixels / fractions, one-pass path (round 6: evk_image2_splat_indexed_f32) ---------
def test_interpolate_to_image_one_pass_against_direct_kernel_and_oracle(E):
    """image.py:102-115 through the public function on device tensors: events whose px + dx is a float32 coordinate travel as
    bilinear records, the others -- fractions that are not the fraction of any float32 coordinate at that pixel, fractions
    outside [0, 1), NaN, pixels at -1 (wrap) -- take the direct kernel's code inside the partition kernel; the image is
    accumulated IN PLACE on top of what it held; out-of-range pixels raise IndexError on both kernel families."""
    import os
    from event_utils_amd import _lib
    from event_utils_amd.representations import image as I
    rng = np.random.default_rng(41)
    n, H, W = 1_500_000, 300, 400
    x = rng.uniform(0, W - 1, n).astype(np.float32); y = rng.uniform(0, H - 1, n).astype(np.float32)
    hot = rng.random(n) < 0.4                                   # a blob: cut tiles
    x[hot] = rng.uniform(100, 130, hot.sum()).astype(np.float32); y[hot] = rng.uniform(50, 70, hot.sum()).astype(np.float32)
    px, py = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)
    dx, dy = (x - np.floor(x)).astype(np.float32), (y - np.floor(y)).astype(np.float32)
    k = n // 10
    dx[:k] = rng.uniform(0, 1, k).astype(np.float32)            # 24 random mantissa bits: px + dx rounds -> rare path
    dy[k:2 * k] = rng.uniform(0, 1, k).astype(np.float32)
    dx[2 * k:2 * k + 500] = 1.5; dy[2 * k + 500:2 * k + 1000] = -0.25
    px[2 * k + 1000:2 * k + 2000] = -1; py[2 * k + 2000:2 * k + 3000] = -1          # wrap to the last column / row
    for weights in ("unit", "float"):
        w = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32) if weights == "unit" else rng.normal(size=n).astype(np.float32)
        base = rng.normal(size=(H, W)).astype(np.float32)
        ref = R.interpolate_to_image(px, py, dx, dy, w, base.copy(), "f64")
        mag = R.interpolate_to_image(px, py, np.abs(dx), np.abs(dy), np.abs(w), np.abs(base), "f64")
        cols = [torch.from_numpy(a).cuda() for a in (px, py, dx, dy, w)]
        calls = []
        orig = _lib.call
        _lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
        try:
            got = {}
            for impl in ("tiled", "direct"):
                os.environ["EVK_IMPL"] = impl
                img = torch.from_numpy(base.copy()).cuda()
                out = I.interpolate_to_image(*cols, img)
                assert out is img
                got[impl] = img.cpu().numpy().astype(np.float64)
                assert calls[-1] == ("evk_image2_splat_indexed_f32" if impl == "tiled" else "evk_splat_indexed_f32")
        finally:
            _lib.call = orig
            os.environ["EVK_IMPL"] = "tiled"
        for impl in got:
            assert np.max(np.abs(got[impl] - ref)) <= 1e-5 * np.max(np.abs(ref)) + 4e-7 * np.max(mag), impl
    # a NaN fraction poisons exactly the pixels the reference poisons; a CPU image is round-tripped and still edited in place
    dxn = dx.copy(); dxn[123_456] = np.nan
    w = np.ones(n, np.float32)
    ref = R.interpolate_to_image(px, py, dxn, dy, w, np.zeros((H, W), np.float32), "f64")
    img = torch.zeros(H, W)
    I.interpolate_to_image(*[torch.from_numpy(a) for a in (px, py, dxn, dy, w)], img)
    assert np.array_equal(np.isnan(img.numpy()), np.isnan(ref)) and np.isnan(ref).sum() == 4
    # pixels whose right neighbour is outside raise, with the same message on both kernel families
    pxb = px.copy(); pxb[77] = W - 1
    msgs = []
    for impl in ("tiled", "direct"):
        os.environ["EVK_IMPL"] = impl
        try:
            with pytest.raises(IndexError) as ei:
                I.interpolate_to_image(*[torch.from_numpy(a).cuda() for a in (pxb, py, dx, dy, w)], torch.zeros(H, W, device="cuda"))
            msgs.append(str(ei.value))
        finally:
            os.environ["EVK_IMPL"] = "tiled"
    assert msgs[0] == msgs[1]


# ---- the derivative splats on the one-pass path (round 6: evk_image2_splat_drv_indexed_f32, evk_image2_drv_f64) ---------------------
def _impls(fn):
    """fn() under EVK_IMPL=tiled and =direct -> {impl: result}, with the library entry points each one called."""
    import os
    from event_utils_amd import _lib
    out, names = {}, {}
    orig = _lib.call
    for impl in ("tiled", "direct"):
        calls = []
        _lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
        os.environ["EVK_IMPL"] = impl
        try:
            out[impl] = fn()
        finally:
            _lib.call = orig
            os.environ["EVK_IMPL"] = "tiled"
        names[impl] = calls
    return out, names


def test_interpolate_to_derivative_img_one_pass(E):
    """image.py:117-136 through the public function: records + the event's index, the tile kernel fetching the four weights;
    blob scene (cut tiles), fractions that are no float32 coordinate's (rare path), pixels at -1 (wrap), IndexError."""
    from event_utils_amd.representations import image as I
    rng = np.random.default_rng(61)
    n, H, W = 1_300_000, 260, 346
    x = rng.uniform(0, W - 1, n).astype(np.float32); y = rng.uniform(0, H - 1, n).astype(np.float32)
    hot = rng.random(n) < 0.4
    x[hot] = rng.uniform(100, 130, hot.sum()).astype(np.float32); y[hot] = rng.uniform(50, 70, hot.sum()).astype(np.float32)
    px, py = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)
    dx, dy = (x - np.floor(x)).astype(np.float32), (y - np.floor(y)).astype(np.float32)
    k = n // 10
    dx[:k] = rng.uniform(0, 1, k).astype(np.float32)
    px[k:k + 1000] = -1; py[k + 1000:k + 2000] = -1
    w1 = rng.normal(size=(2, n)).astype(np.float32); w2 = rng.normal(size=(2, n)).astype(np.float32)
    base = rng.normal(size=(2, H, W)).astype(np.float32)
    ref = R.interpolate_to_derivative_img(px, py, dx, dy, base.copy(), w1, w2, "f64")
    mag = R.interpolate_to_derivative_img(px, py, dx, dy, np.abs(base), np.abs(w1), np.abs(w2), "f64")   # (a bound of the summed magnitudes)
    cols = [torch.from_numpy(a).cuda() for a in (px, py, dx, dy)]
    wd1, wd2 = torch.from_numpy(w1).cuda(), torch.from_numpy(w2).cuda()

    def run():
        d = torch.from_numpy(base.copy()).cuda()
        assert I.interpolate_to_derivative_img(*cols, d, wd1, wd2) is d
        return d.cpu().numpy().astype(np.float64)
    got, names = _impls(run)
    assert "evk_image2_splat_drv_indexed_f32" in names["tiled"] and "evk_splat_drv_indexed_f32" in names["direct"]
    bound = 1e-5 * np.max(np.abs(ref)) + 2e-6 * (np.max(np.abs(mag)) + 4 * np.max(np.abs(ref)))
    for impl in got:
        assert np.max(np.abs(got[impl] - ref)) <= bound, (impl, np.max(np.abs(got[impl] - ref)), bound)
    pxb = px.copy(); pxb[5] = W - 1
    colsb = [torch.from_numpy(pxb).cuda()] + cols[1:]
    msgs, _ = _impls(lambda: str(pytest.raises(IndexError, I.interpolate_to_derivative_img, *colsb, torch.zeros(2, H, W, device="cuda"),
                                               wd1, wd2).value))
    assert msgs["tiled"] == msgs["direct"]


@pytest.mark.parametrize("grad", [True, False])
def test_events_to_image_drv_one_pass(E, grad):
    """image.py:162-217 through the public function (float64 numpy in, float32 numpy out): coordinates cast to float32 before
    floor (Q7), clipped events masked to weight 0, pixels at -1 wrapping, a hot blob; image and both derivative planes against
    the direct kernel and the oracle."""
    rng = np.random.default_rng(62)
    n, H, W = 1_100_000, 180, 240
    x = rng.uniform(0, W + 3, n); y = rng.uniform(0, H + 2, n)                 # beyond the padded image's clip: masked
    hot = rng.random(n) < 0.4
    x[hot] = rng.uniform(100, 130, hot.sum()); y[hot] = rng.uniform(50, 70, hot.sum())
    x[:1500] = rng.uniform(-1, 0, 1500); y[1500:2500] = rng.uniform(-1, 0, 1000)  # wrap to the last column / row
    p = rng.normal(size=n)
    jx, jy = (rng.normal(size=(2, n)), rng.normal(size=(2, n))) if grad else (None, None)
    ref_i, ref_d = R.events_to_image_drv(x, y, p, jx, jy, sensor_size=(H, W), compute_gradient=grad, accum="f64")
    mag_i, mag_d = R.events_to_image_drv(x, y, np.abs(p), None if jx is None else np.abs(jx), None if jy is None else np.abs(jy),
                                         sensor_size=(H, W), compute_gradient=grad, accum="f64")
    got, names = _impls(lambda: E.events_to_image_drv(x, y, p, jx, jy, sensor_size=(H, W), compute_gradient=grad))
    assert "evk_image2_drv_f64" in names["tiled"] and "evk_image_drv_f64" in names["direct"]
    for impl, (gi, gd) in got.items():
        assert gi.dtype == np.float32 and gi.shape == (H + 1, W + 1)
        assert np.max(np.abs(gi.astype(np.float64) - ref_i)) <= 1e-5 * np.max(np.abs(ref_i)) + 4e-7 * np.max(np.abs(mag_i)), impl
        if grad:
            bound = 1e-5 * np.max(np.abs(ref_d)) + 2e-6 * (np.max(np.abs(mag_d)) + 4 * np.max(np.abs(ref_d)))
            assert gd.shape == (2, H + 1, W + 1) and np.max(np.abs(gd.astype(np.float64) - ref_d)) <= bound, impl
        else:
            assert gd is None
