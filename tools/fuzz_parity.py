"""Differential fuzzing of the public voxel / event-image / get_iwe / objective calls against the CPU oracle (oracle/reference_np.py), beyond the
sizes and seeds of tests/: sensor sizes up to 1300 x 800, event counts around every boundary of the one-pass path (one wave,
one sub-chunk, the 'auto' thresholds, several sub-chunks per workgroup), scenes that cut hot tiles, polarities of every kind
(+-1, zeros, small integers, float32, huge, NaN / infinite), time stamps that are constant / few-valued / unsorted, every
EVK_IMPL.  Test infrastructure (imports the oracle): not part of the product.
usage: python tools/fuzz_parity.py [--seconds S] [--seed0 K] [--kinds voxel,image,native,iwe,objective,windows,misc,errors,prims,search]     exit code 1 on any mismatch"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import event_utils_amd as E  # noqa: E402
from event_utils_amd import tiled  # noqa: E402
from oracle import reference_np as R  # noqa: E402

N_CHOICES = [1, 2, 63, 64, 65, 1000, 8191, 8192, 8193, 12_289, 79_999, 80_001, 149_999, 150_001, 319_999, 320_001, 1_000_003, 2_500_000]


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def coords(rng, n, H, W, real, scene):
    if real:
        x = rng.uniform(0, W - 1, n); y = rng.uniform(0, H - 1, n)
    else:
        x = rng.integers(0, W, n).astype(np.float64); y = rng.integers(0, H, n).astype(np.float64)
    if scene == "blob" and n > 8:
        hot = rng.random(n) < 0.7
        x[hot] = np.clip(W // 2 + rng.integers(-2, 3, hot.sum()), 0, W - 1); y[hot] = np.clip(H // 3 + rng.integers(-2, 3, hot.sum()), 0, H - 1)
    elif scene == "pixel" and n > 8:
        hot = rng.random(n) < 0.9
        x[hot] = W - 1; y[hot] = H - 1
    elif scene == "edge" and n > 8:
        hot = rng.random(n) < 0.8
        x[hot] = np.clip(W * 0.37 + rng.normal(0, 0.6, hot.sum()), 0, W - 1)
        if not real:
            x = np.floor(x)
    return x.astype(np.float32), y.astype(np.float32)


def weights(rng, n, kind):
    if kind == "pm1":
        return (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    if kind == "pm1z":
        return rng.integers(-1, 2, n).astype(np.float32)
    if kind == "ones":
        return np.ones(n, np.float32)
    if kind == "ints":
        return rng.integers(-3, 4, n).astype(np.float32)
    if kind == "float":
        return rng.normal(size=n).astype(np.float32)
    if kind == "huge":
        return (rng.normal(size=n) * 1e30).astype(np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)        # "special": a few NaN / inf / -0 among unit polarities
    if n > 4:
        idx = rng.integers(0, n, 4)
        p[idx] = [np.nan, np.inf, -np.inf, -0.0]
    return p


def times(rng, n, kind):
    if kind == "sorted":
        return np.sort(rng.uniform(3.0, 3.2, n)).astype(np.float32)
    if kind == "const":
        return np.full(n, 1.5, np.float32)
    if kind == "few":
        return np.sort(rng.integers(0, 4, n)).astype(np.float32)
    if kind == "ends":          # everything at the two ends: t_norm exactly 0 or B - 1
        t = np.zeros(n, np.float32); t[n // 2:] = 1.0
        return t
    t = rng.uniform(0.0, 1.0, n).astype(np.float32)           # "unsorted": ts[0], ts[-1] are whatever they are (Q9)
    return t


def device_cols(rng, arrays):
    """The arrays as device tensors -- usually freshly allocated (16-byte aligned), in a third of the cases as SLICES that start
    1-3 elements into a larger buffer, per column, with readable storage behind them or (one case in four of those) ending with
    their storage: the one-pass paths read the former where they lie and copy the latter (tiled.column_ok)."""
    if rng.random() > 0.33:
        return [torch.from_numpy(a).cuda() for a in arrays]
    out = []
    for a in arrays:
        off = int(rng.integers(0, 4))
        tail = 0 if rng.random() < 0.25 else int(rng.integers(3, 9))
        buf = torch.empty(off + len(a) + tail, dtype=torch.from_numpy(a[:1]).dtype if len(a) else torch.float32, device="cuda")
        buf[off:off + len(a)] = torch.from_numpy(a).cuda()
        out.append(buf[off:off + len(a)])
    return out


def same(got, ref, mag, what, magf=4e-7):
    """float32 accumulation against the float64 oracle: 1e-5 of the reference's maximum + the float32 rounding of the summed
    magnitudes (cancelling sums); NaN / infinite cells in the same places."""
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    if got.shape != ref.shape:
        return "%s: shape %s vs %s" % (what, got.shape, ref.shape)
    bad_g, bad_r = ~np.isfinite(got), ~np.isfinite(ref)
    if not np.array_equal(bad_g, bad_r):
        return "%s: non-finite cells differ (%d vs %d)" % (what, bad_g.sum(), bad_r.sum())
    if bad_r.any() and not (np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(got[bad_r & ~np.isnan(ref)], ref[bad_r & ~np.isnan(ref)])):
        return "%s: NaN / infinity pattern differs" % what
    ok = ~bad_r
    if not ok.any():
        return None
    fin = np.abs(ref[ok])
    tol = 1e-5 * max(fin.max(), 1e-30) + magf * float(np.max(np.abs(mag)[ok])) if mag is not None else 1e-5 * max(fin.max(), 1e-30)
    err = np.max(np.abs(got[ok] - ref[ok]))
    return None if err <= tol else "%s: max error %.3e > %.3e (max |ref| %.3e)" % (what, err, tol, fin.max())


def case_voxel(rng):
    H, W = int(rng.integers(2, 800)), int(rng.integers(2, 1300))
    if rng.random() < 0.3:
        H, W = [(180, 240), (260, 346), (480, 640), (720, 1280)][int(rng.integers(0, 4))]
    B = int(rng.integers(1, 12))
    n = int(rng.choice(N_CHOICES))
    real = bool(rng.integers(0, 2))
    scene = str(rng.choice(["uniform", "uniform", "blob", "pixel", "edge"]))
    pk = str(rng.choice(["pm1", "pm1", "pm1z", "ones", "ints", "float", "huge", "special"]))
    tk = str(rng.choice(["sorted", "sorted", "sorted", "const", "few", "ends", "unsorted"]))
    impl = str(rng.choice(["auto", "tiled", "tiled", "direct"]))
    det = bool(rng.random() < 0.25)
    rec = [None, None, 4, 8][int(rng.integers(0, 4))] if impl == "tiled" else None
    if det and impl == "direct":      # a contradiction the call refuses (ValueError)
        impl = "auto"
    desc = "voxel %dx%dx%d n=%d real=%d %s p=%s t=%s impl=%s det=%d rec=%s" % (B, H, W, n, real, scene, pk, tk, impl, det, rec)
    x, y = coords(rng, n, H, W, real, scene)
    p, t = weights(rng, n, pk), times(rng, n, tk)
    with np.errstate(all="ignore"):
        ref = R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64")
        fin = np.where(np.isfinite(p), np.abs(p), 0).astype(np.float32)
        mag = R.events_to_voxel_torch(x, y, t, fin, B, sensor_size=(H, W), accum="f64")
    os.environ["EVK_IMPL"] = impl
    if det:
        os.environ["EVK_VOXEL_DETERMINISTIC"] = "1"
    tiled.FORCE["rec"] = rec
    try:
        cols = device_cols(rng, (x, y, t, p))
        got = E.events_to_voxel_torch(*cols, B, sensor_size=(H, W)).cpu().numpy()
        if det and pk in ("huge", "special"):   # not representable in the fixed-point cells: refused (documented), not compared
            return desc, None
    except Exception as e:  # noqa: BLE001
        if det and pk in ("huge", "special"):
            return desc, None
        return desc, "raised %s: %s" % (type(e).__name__, e)
    finally:
        os.environ.pop("EVK_IMPL", None); os.environ.pop("EVK_VOXEL_DETERMINISTIC", None)
        tiled.FORCE["rec"] = None
    if not np.isfinite(mag).all():
        mag = np.where(np.isfinite(mag), mag, 0.0)
    # the direct kernels add float32 atomics as the reference's index_put_ does: on a pixel that collects 10^5 events their own
    # rounding reaches 10^-4 of the cell (the one-pass path accumulates float64 / integers)
    direct_like = impl == "direct" or (impl == "auto" and n < tiled.TILED_MIN_EVENTS and not det)
    # (a small call on columns that cannot be read in place -- no slack behind a misaligned slice -- keeps the direct kernel too)
    direct_like = direct_like or (impl == "auto" and not det and n * 2 < tiled.REALIGN_ATOMICS and not all(tiled.column_ok(c) for c in cols))
    return desc, same(got, ref, mag, "grid", 1e-3 if direct_like and scene in ("pixel", "blob", "edge") else 4e-7)


def case_image(rng):
    H, W = int(rng.integers(2, 800)), int(rng.integers(2, 1300))
    if rng.random() < 0.3:
        H, W = [(180, 240), (260, 346), (480, 640), (720, 1280)][int(rng.integers(0, 4))]
    n = int(rng.choice(N_CHOICES))
    interp = [None, "bilinear"][int(rng.integers(0, 2))]
    padding = bool(rng.integers(0, 2))
    clip = bool(rng.integers(0, 2)) or interp == "bilinear"
    default = float(rng.choice([0.0, 0.0, 0.5]))
    real = bool(rng.integers(0, 2))
    scene = str(rng.choice(["uniform", "uniform", "blob", "pixel", "edge"]))
    pk = str(rng.choice(["pm1", "pm1", "pm1z", "ones", "ints", "float", "huge", "special"]))
    impl = str(rng.choice(["auto", "tiled", "tiled", "direct"]))
    desc = "image %dx%d n=%d %s pad=%d clip=%d default=%g real=%d %s p=%s impl=%s" % (H, W, n, interp, padding, clip, default, real, scene, pk, impl)
    x, y = coords(rng, n, H, W, real, scene)
    if clip and n > 16:            # beyond the sensor: clipped events (Q8: they pile up at (0, 0) in the nearest branch)
        k = rng.integers(0, n, max(1, n // 50))
        x[k] += np.float32(W); y[k[: len(k) // 2]] += np.float32(H)
    elif not clip and real:
        x = np.minimum(x, np.float32(W - 1.001)); y = np.minimum(y, np.float32(H - 1.001))
        x = np.maximum(x, 0); y = np.maximum(y, 0)
    p = weights(rng, n, pk)
    xa, ya = (x, y) if real else (x.astype(np.int64), y.astype(np.int64))
    with np.errstate(all="ignore"):
        ref = R.events_to_image_torch(xa, ya, p, sensor_size=(H, W), clip_out_of_range=clip, interpolation=interp,
                                      padding=padding, default=default, accum="f64")
        fin = np.where(np.isfinite(p), np.abs(p), 0).astype(np.float32)
        mag = R.events_to_image_torch(xa, ya, fin, sensor_size=(H, W), clip_out_of_range=clip, interpolation=interp,
                                      padding=padding, default=abs(default), accum="f64")
    os.environ["EVK_IMPL"] = impl
    try:
        got = E.events_to_image_torch(*device_cols(rng, (xa, ya, p)),
                                      sensor_size=(H, W), clip_out_of_range=clip, interpolation=interp, padding=padding,
                                      default=default).cpu().numpy()
    except Exception as e:  # noqa: BLE001
        return desc, "raised %s: %s" % (type(e).__name__, e)
    finally:
        os.environ.pop("EVK_IMPL", None)
    err = same(got, ref, mag, "image")
    if err is None and rng.random() < 0.5:
        # numpy entry point (image.py:5-44): integer coordinates on the (H+1, W+1) canvas, bit-exact
        xi, yi = rng.integers(0, W + 1, n), rng.integers(0, H + 1, n)
        pi = rng.integers(-3, 4, n) if rng.random() < 0.5 else np.ones(n, np.int64)
        mv = bool(rng.integers(0, 2))
        os.environ["EVK_IMPL"] = impl
        try:
            a = E.events_to_image(xi, yi, pi, sensor_size=(H, W), meanval=mv, default=default)
        finally:
            os.environ.pop("EVK_IMPL", None)
        with np.errstate(all="ignore"):
            b = R.events_to_image(xi, yi, pi, sensor_size=(H, W), meanval=mv, default=default)
        if not (a.dtype == b.dtype and np.array_equal(a, b, equal_nan=True)):
            err = "events_to_image (meanval=%d): not bit-exact, %d cells differ" % (mv, int((a != b).sum()))
    return desc, err


def case_native(rng):
    """events in the on-disk dtypes (int16 coordinates, uint8 / bool polarities, float32 time stamps) through the public call"""
    H, W = [(180, 240), (260, 346), (480, 640), (720, 1280)][int(rng.integers(0, 4))]
    B = int(rng.integers(1, 10))
    n = int(rng.choice(N_CHOICES))
    scene = str(rng.choice(["uniform", "blob", "edge"]))
    impl = str(rng.choice(["auto", "tiled", "direct"]))
    desc = "native %dx%dx%d n=%d %s impl=%s" % (B, H, W, n, scene, impl)
    x, y = coords(rng, n, H, W, False, scene)
    t = times(rng, n, "sorted")
    p8 = rng.integers(0, 2, n).astype(np.uint8)
    with np.errstate(all="ignore"):
        ref = R.events_to_voxel_torch(x, y, t, p8.astype(np.float32), B, sensor_size=(H, W), accum="f64")
    os.environ["EVK_IMPL"] = impl
    try:
        got = E.events_to_voxel_torch(torch.from_numpy(x.astype(np.int16)).cuda(), torch.from_numpy(y.astype(np.int16)).cuda(),
                                      torch.from_numpy(t).cuda(), torch.from_numpy(p8).cuda(), B, sensor_size=(H, W)).cpu().numpy()
    except Exception as e:  # noqa: BLE001
        return desc, "raised %s: %s" % (type(e).__name__, e)
    finally:
        os.environ.pop("EVK_IMPL", None)
    return desc, same(got, ref, None, "grid")


def case_windows(rng):
    """voxel_grids_fixed_n_torch / voxel_grids_fixed_t_torch / events_to_voxel_timesync_torch / events_to_neg_pos_voxel_torch
    (voxel_grid.py:37-112,155-182): the reference loops events_to_voxel_torch over slices; so does the check"""
    H, W = int(rng.integers(2, 300)), int(rng.integers(2, 400))
    B = int(rng.integers(1, 8))
    n = int(rng.choice([65, 1000, 8193, 100_000, 400_000, 1_000_003]))
    which = str(rng.choice(["fixed_n", "fixed_t", "timesync", "neg_pos"]))
    scene = str(rng.choice(["uniform", "blob", "edge"]))
    impl = str(rng.choice(["auto", "tiled", "direct"]))
    x, y = coords(rng, n, H, W, bool(rng.integers(0, 2)), scene)
    t = times(rng, n, "sorted")
    p = weights(rng, n, str(rng.choice(["pm1", "pm1z", "float"])))
    desc = "windows %s %dx%dx%d n=%d %s impl=%s" % (which, B, H, W, n, scene, impl)
    tt = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    os.environ["EVK_IMPL"] = impl
    try:
        with np.errstate(all="ignore"):
            if which == "fixed_n":
                k = int(rng.choice([max(2, n // 7), max(2, n // 3), n, 33 if n <= 1000 else max(2, n // 11)]))
                got = E.voxel_grids_fixed_n_torch(*tt, B, k, sensor_size=(H, W))
                ref = [R.events_to_voxel_torch(x[i:i + k], y[i:i + k], t[i:i + k], p[i:i + k], B, sensor_size=(H, W), accum="f64")
                       for i in range(0, n - k, k)]
                desc += " k=%d" % k
            elif which == "fixed_t":
                span = float(t[-1] - t[0])
                dt = span / float(rng.choice([1.5, 3.0, 7.3]))
                got = E.voxel_grids_fixed_t_torch(*tt, B, dt, sensor_size=(H, W))
                ref = []
                for ts0 in np.arange(t[0].item(), t[-1].item() - dt, dt):
                    a, b = np.searchsorted(t, ts0), np.searchsorted(t, ts0 + dt)
                    ref.append(R.events_to_voxel_torch(x[a:b], y[a:b], t[a:b], p[a:b], B, sensor_size=(H, W), accum="f64"))
            elif which == "timesync":
                t0 = float(t[0]) + 0.2 * float(t[-1] - t[0]); t1 = float(t[0]) + 0.7 * float(t[-1] - t[0])
                got = [E.events_to_voxel_timesync_torch(*tt, B, t0, t1, sensor_size=(H, W))]
                a, b = np.searchsorted(t, t0), np.searchsorted(t, t1)
                ref = [R.events_to_voxel_torch(x[a:b], y[a:b], t[a:b], p[a:b], B, sensor_size=(H, W), accum="f64")]
            else:
                got = list(E.events_to_neg_pos_voxel_torch(*tt, B, sensor_size=(H, W)))
                ref = [R.events_to_voxel_torch(x, y, t, np.where(p > 0, 1, 0).astype(np.float32), B, sensor_size=(H, W), accum="f64"),
                       R.events_to_voxel_torch(x, y, t, np.where(p <= 0, 1, 0).astype(np.float32), B, sensor_size=(H, W), accum="f64")]
    except Exception as e:  # noqa: BLE001
        return desc, "raised %s: %s" % (type(e).__name__, e)
    finally:
        os.environ.pop("EVK_IMPL", None)
    if len(got) != len(ref):
        return desc, "%d grids vs %d" % (len(got), len(ref))
    for k, (a, b) in enumerate(zip(got, ref)):
        hot = scene in ("blob", "edge")
        err = same(a.cpu().numpy(), b, np.abs(b) if hot else None, "grid %d" % k, 1e-3)   # windows / small calls: float32 atomics
        if err is not None:
            return desc, err
    return desc, None


def case_errors(rng):
    """coordinates beyond the sensor: index_put_ wraps -size <= index < 0 and raises IndexError outside (image.py:93-99 re-raises
    it); the oracle decides which, the call must do the same -- for every EVK_IMPL, synchronous and deferred reporting"""
    H, W = int(rng.integers(4, 500)), int(rng.integers(4, 700))
    n = int(rng.choice([3, 65, 1000, 8193, 100_000, 400_000, 1_000_003]))
    which = str(rng.choice(["voxel", "nearest", "bilinear"]))
    impl = str(rng.choice(["auto", "tiled", "direct"]))
    how = str(rng.choice(["neg_wrap", "neg_far", "beyond", "edge", "frac_neg"]))
    errs = str(rng.choice(["strict", "deferred"]))
    clip = bool(rng.integers(0, 2)) or which == "bilinear"
    pad = bool(rng.integers(0, 2))
    desc = "errors %s %dx%d n=%d impl=%s %s clip=%d pad=%d EVK_ERRORS=%s" % (which, H, W, n, impl, how, clip, pad, errs)
    x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
    if which == "bilinear":
        x = np.minimum(x + rng.random(n).astype(np.float32), np.float32(W - 1)); y = np.minimum(y + rng.random(n).astype(np.float32), np.float32(H - 1))
    k = rng.integers(0, n, min(n, 3))
    if how == "neg_wrap":
        x[k] = -float(rng.integers(1, W)); y[k[:1]] = -float(rng.integers(1, H))
    elif how == "neg_far":
        x[k[:1]] = -float(W + 1 + rng.integers(0, 50))
    elif how == "beyond":
        if rng.random() < 0.5:
            x[k[:1]] = float(W + rng.integers(0, 3))
        else:
            y[k[:1]] = float(H + rng.integers(0, 3))
    elif how == "edge":
        x[k] = float(W - 1); y[k] = float(H - 1)
    else:
        x[k] = -0.5; y[k[:1]] = -0.25          # truncation toward zero (nearest) vs floor (bilinear)
    t = np.sort(rng.uniform(0, 1, n)).astype(np.float32)
    p = weights(rng, n, str(rng.choice(["pm1", "float"])))
    B = int(rng.integers(1, 6))
    ref, ref_exc = None, None
    try:
        with np.errstate(all="ignore"):
            if which == "voxel":
                ref = R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64")
            else:
                ref = R.events_to_image_torch(x, y, p, sensor_size=(H, W), clip_out_of_range=clip,
                                              interpolation=None if which == "nearest" else "bilinear", padding=pad, accum="f64")
    except IndexError as e:
        ref_exc = e
    os.environ["EVK_IMPL"] = impl
    os.environ["EVK_ERRORS"] = errs
    got, exc = None, None
    try:
        c = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
        if which == "voxel":
            got = E.events_to_voxel_torch(*c, B, sensor_size=(H, W))
        else:
            got = E.events_to_image_torch(c[0], c[1], c[3], sensor_size=(H, W), clip_out_of_range=clip,
                                          interpolation=None if which == "nearest" else "bilinear", padding=pad)
        E.check_errors()           # deferred reports surface here at the latest
        got = got.cpu().numpy()
    except IndexError as e:
        exc = e
    except Exception as e:  # noqa: BLE001
        return desc, "raised %s: %s" % (type(e).__name__, e)
    finally:
        os.environ.pop("EVK_IMPL", None); os.environ.pop("EVK_ERRORS", None)
        try:
            E.check_errors()
        except Exception:  # noqa: BLE001
            pass
    if (ref_exc is None) != (exc is None):
        return desc, "reference %s, here %s" % ("raises IndexError" if ref_exc else "returns", "raises IndexError" if exc else "returns")
    if ref_exc is not None:
        return desc, None
    mag = None
    if which != "voxel":       # clipped events pile up at (0, 0) with their weights (Q8): float32 atomics in the direct kernels
        with np.errstate(all="ignore"):
            mag = R.events_to_image_torch(x, y, np.abs(p), sensor_size=(H, W), clip_out_of_range=clip,
                                          interpolation=None if which == "nearest" else "bilinear", padding=pad, accum="f64")
    return desc, same(got, ref, mag, which, 1e-4)


def case_prims(rng):
    """the building blocks: linvel_warp.warp and events_bounds_mask (bit-exact, float64), events_to_image_drv with arbitrary
    Jacobians, interpolate_to_image / interpolate_to_derivative_img, the dense-flow warp, the Gaussian blur (bit-identical to
    scipy's: oracle gaussian_filter_reflect is pinned to it by fixture f10)"""
    from event_utils_amd.contrast_max.objectives import gaussian_filter_device
    from event_utils_amd.transforms.optic_flow import warp_events_flow_torch
    which = str(rng.choice(["warp", "drv", "splat", "flow", "blur"]))
    H, W = int(rng.integers(4, 300)), int(rng.integers(4, 400))
    n = int(rng.choice([1, 65, 1000, 8193, 100_000, 400_000]))
    desc = "prims %s %dx%d n=%d" % (which, H, W, n)
    try:
        with np.errstate(all="ignore"):
            if which == "warp":
                x = rng.uniform(-5, W + 5, n); y = rng.uniform(-5, H + 5, n); t = np.sort(rng.uniform(0, 10 ** float(rng.integers(-3, 10)), n))
                prm = rng.normal(0, 1, 2) * float(rng.choice([0.0, 1e-3, 50.0, 1e4]))
                t0 = float(t[-1]) if rng.random() < 0.7 else float(rng.uniform(t[0], t[-1]))
                got = E.linvel_warp().warp(x, y, t, np.ones(n), t0, prm, compute_grad=True)
                ref = R.linvel_warp().warp(x, y, t, np.ones(n), t0, prm, compute_grad=True)
                for a, b, nm in zip(got, ref, ("x'", "y'", "jx", "jy")):
                    if not (np.asarray(a).dtype == np.float64 and np.array_equal(a, b)):
                        return desc, "%s not bit-exact" % nm
                lim = (0, W, 0, H) if rng.random() < 0.5 else (float(rng.uniform(-3, 3)), float(rng.uniform(3, W + 3)), float(rng.uniform(-3, 3)), float(rng.uniform(3, H + 3)))
                if not np.array_equal(E.events_bounds_mask(got[0], got[1], *lim), R.events_bounds_mask(ref[0], ref[1], *lim)):
                    return desc, "bounds mask differs"
                return desc, None
            if which == "drv":
                x = rng.uniform(-1, W + 2, n); y = rng.uniform(-1, H + 2, n)
                x, y = np.maximum(x, 0), np.maximum(y, 0)           # (negative pixels wrap or raise: the errors kind)
                p = rng.normal(size=n); jx = rng.normal(size=(2, n)); jy = rng.normal(size=(2, n))
                grad = bool(rng.integers(0, 2))
                gi, gd = E.events_to_image_drv(x, y, p, jx, jy, sensor_size=(H, W), compute_gradient=grad)
                ri, rd = R.events_to_image_drv(x, y, p, jx, jy, sensor_size=(H, W), compute_gradient=grad, accum="f64")
                mi, md = R.events_to_image_drv(x, y, np.abs(p), np.abs(jx), np.abs(jy), sensor_size=(H, W), compute_gradient=grad, accum="f64")
                err = same(gi, ri, mi, "iwe", 1e-6)
                if err is None and grad:
                    err = same(gd, rd, None, "d_iwe") if gd is not None else "d_iwe is None"
                return desc, err
            if which == "splat":
                px = rng.integers(0, W - 1, n); py = rng.integers(0, H - 1, n)
                dx = rng.random(n).astype(np.float32); dy = rng.random(n).astype(np.float32)
                kind = str(rng.choice(["random", "coords", "mixed"]))
                if kind != "random":      # pixels and fractions taken from float32 coordinates (what upstream's callers pass): records
                    xc, yc = coords(rng, n, H, W, True, str(rng.choice(["uniform", "blob", "pixel", "edge"])))
                    take = np.ones(n, bool) if kind == "coords" else rng.random(n) < 0.7
                    px = np.where(take, np.floor(xc).astype(np.int64), px); py = np.where(take, np.floor(yc).astype(np.int64), py)
                    dx = np.where(take, xc - np.floor(xc), dx).astype(np.float32); dy = np.where(take, yc - np.floor(yc), dy).astype(np.float32)
                    px = np.minimum(px, W - 2); py = np.minimum(py, H - 2)
                if n > 64 and rng.random() < 0.5:
                    px[:16] = -1; py[16:32] = -1                  # wrap once, as index_put_ does
                w = rng.normal(size=n).astype(np.float32) if rng.random() < 0.5 else weights(rng, n, str(rng.choice(["pm1", "pm1z", "ints"])))
                os.environ["EVK_IMPL"] = str(rng.choice(["auto", "tiled", "direct"]))
                desc += " %s impl=%s" % (kind, os.environ["EVK_IMPL"])
                w1 = rng.normal(size=(2, n)).astype(np.float32); w2 = rng.normal(size=(2, n)).astype(np.float32)
                ref = R.interpolate_to_image(px, py, dx, dy, w, np.zeros((H, W), np.float32), accum="f64")
                img = torch.zeros(H, W, device="cuda")
                E.interpolate_to_image(*(torch.from_numpy(a).cuda() for a in (px, py, dx, dy, w)), img)
                err = same(img.cpu().numpy(), ref, R.interpolate_to_image(px, py, dx, dy, np.abs(w), np.zeros((H, W), np.float32), accum="f64"), "splat", 1e-6)
                if err is not None:
                    return desc, err
                refd = R.interpolate_to_derivative_img(px, py, dx, dy, np.zeros((2, H, W), np.float32), w1, w2, accum="f64")
                dimg = torch.zeros(2, H, W, device="cuda")
                E.interpolate_to_derivative_img(*(torch.from_numpy(a).cuda() for a in (px, py, dx, dy)), dimg, torch.from_numpy(w1).cuda(), torch.from_numpy(w2).cuda())
                magd = R.interpolate_to_derivative_img(px, py, dx, dy, np.zeros((2, H, W), np.float32), np.abs(w1), np.abs(w2), accum="f64")
                return desc, same(dimg.cpu().numpy(), refd, np.full(refd.shape, np.max(np.abs(magd)) * 4 + 1.0), "derivative splat", 1e-6)
            if which == "flow":
                n = max(n, 2)           # (the oracle squeezes its inputs as upstream does: one event is a 0-d array there)
                x = rng.uniform(-2, W + 1, n).astype(np.float32); y = rng.uniform(-2, H + 1, n).astype(np.float32)
                t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32)
                flow = (rng.normal(size=(2, H, W)) * 30).astype(np.float32)
                t0 = None if rng.random() < 0.5 else float(rng.uniform(0, 0.1))
                gx, gy = warp_events_flow_torch(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(t).cuda(), None,
                                                torch.from_numpy(flow).cuda(), t0=t0)
                rx, ry = R.warp_events_flow_torch(x, y, t, None, flow, t0=t0)
                for a, b, nm in ((gx, rx, "x"), (gy, ry, "y")):
                    err = same(a.cpu().numpy(), b, None, "warped " + nm)
                    if err is not None:
                        return desc, err
                return desc, None
            sigma = float(rng.choice([0.5, 1.0, 2.0, 3.3]))
            a = rng.normal(size=(2, H, W) if rng.random() < 0.5 else (H, W)).astype(np.float32)
            got = gaussian_filter_device(torch.from_numpy(a).cuda(), sigma).cpu().numpy()
            ref = R.gaussian_filter_reflect(a, sigma)
            return desc + " sigma=%g" % sigma, None if np.array_equal(got, ref) else "blur not bit-identical (max %.3e)" % np.max(np.abs(got - ref))
    except Exception as e:  # noqa: BLE001
        return desc, "raised %s: %s" % (type(e).__name__, e)
    finally:
        os.environ.pop("EVK_IMPL", None)


def case_search(rng):
    """grid_search_initial (every sample and the best one) and the objective landscape.  (The optimisers are not fuzzed: BFGS on
    forward differences over a rugged landscape amplifies a 1e-7 difference of the objective into another local optimum -- on the
    oracle and on the HIP path alike; their parity is the golden trace f9 and tests/test_gpu_parity.py::test_f9_*.)"""
    from event_utils_amd.contrast_max import events_cmax as C
    which = str(rng.choice(["grid", "landscape"]))
    H, W = int(rng.integers(60, 200)), int(rng.integers(80, 260))
    n = int(rng.choice([20_000, 60_000, 200_000]))
    v = rng.uniform(-60, 60, 2)
    desc = "search %s %dx%d n=%d true flow (%.1f, %.1f)" % (which, H, W, n, v[0], v[1])
    # edges moving at v: events at x = x_edge + v t (vertical lines) and y = y_edge + v t (horizontal lines), a little noise
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32).astype(np.float64)
    vert = rng.random(n) < 0.5
    ex, ey = rng.choice(np.linspace(0.25, 0.75, 4) * W, n), rng.choice(np.linspace(0.25, 0.75, 3) * H, n)
    x = np.where(vert, ex, rng.uniform(0.15 * W, 0.85 * W, n)) + v[0] * t + rng.normal(0, 0.3, n)
    y = np.where(vert, rng.uniform(0.15 * H, 0.85 * H, n), ey) + v[1] * t + rng.normal(0, 0.3, n)
    x, y = (np.clip(a, 1, lim - 2).astype(np.float32).astype(np.float64) for a, lim in ((x, W), (y, H)))
    p = rng.choice([-1.0, 1.0], n) if rng.random() < 0.5 else np.ones(n)
    eo, ro = E.variance_objective(), R.variance_objective()
    eo.sensor_size = ro.sensor_size = (H, W)
    ro.accum = "f64"
    try:
        with np.errstate(all="ignore"):
            if which == "grid":
                kw = dict(log_scale=bool(rng.integers(0, 2)), num_samples_per_param=int(rng.choice([3, 5])))
                if rng.random() < 0.5:
                    kw["param_ranges"] = [[-80, 80], [-100, 60]]
                r = C.grid_search_initial(x, y, t, p, E.linvel_warp(), eo, (H, W), **kw)
                ref = R.grid_search_initial(x, y, t, p, R.linvel_warp(), ro, (H, W), **kw)
                if not np.array_equal(np.array(r["params"]), np.array(ref["params"])):
                    return desc, "sample positions differ"
                ev, rv = np.asarray(r["eval"], np.float64), np.asarray(ref["eval"], np.float64)
                if np.max(np.abs(ev - rv)) > 2e-5 * np.max(np.abs(rv)):
                    return desc, "sample values: max error %.3e of %.3e" % (np.max(np.abs(ev - rv)), np.max(np.abs(rv)))
                # the best sample: the same one unless two samples tie within the tolerance
                if tuple(r["min_params"]) != tuple(ref["min_params"]):
                    i = ref["params"].index(tuple(r["min_params"]))
                    if abs(rv[i] - ref["min_func_eval"]) > 2e-5 * abs(ref["min_func_eval"]):
                        return desc, "best sample %s vs %s" % (r["min_params"], ref["min_params"])
                return desc, None
            if which == "landscape":
                res = int(rng.choice([40, 50]))
                got = C.objective_landscape(x, y, t, p, eo, E.linvel_warp(), x_range=(-100, 100), y_range=(-100, 100), resolution=res, img_size=(H, W))
                ref = R.objective_landscape(x, y, t, p, ro, R.linvel_warp(), x_range=(-100, 100), y_range=(-100, 100), resolution=res, img_size=(H, W))
                return desc, same(got, ref, None, "landscape", 0) if np.max(np.abs(np.asarray(got) - ref)) > 3e-5 else None
            return desc, None
    except Exception as e:  # noqa: BLE001
        return desc, "raised %s: %s" % (type(e).__name__, e)


def case_misc(rng):
    """events_to_voxel (numpy float64 path), the timestamp images, the event-weights gather, batched objective evaluation"""
    from event_utils_amd.events import DeviceEvents
    which = str(rng.choice(["voxel_np", "ts_image", "ts_image_torch", "ts_hard", "gather", "batch"]))
    H, W = int(rng.integers(4, 300)), int(rng.integers(4, 400))
    n = int(rng.choice([2, 65, 1000, 8193, 100_000, 400_000]))
    impl = str(rng.choice(["auto", "tiled", "direct"]))
    desc = "misc %s %dx%d n=%d impl=%s" % (which, H, W, n, impl)
    os.environ["EVK_IMPL"] = impl
    try:
        with np.errstate(all="ignore"):
            if which == "voxel_np":
                B = int(rng.integers(1, 9))
                x, y = rng.integers(0, W + 1, n), rng.integers(0, H + 1, n)       # the (H+1, W+1) canvas of events_to_image
                t = np.sort(rng.uniform(0, 1, n)); p = rng.choice([-1.0, 1.0], n) if rng.random() < 0.5 else rng.normal(size=n)
                got = E.events_to_voxel(x, y, t, p, B, sensor_size=(H, W))
                ref = R.events_to_voxel(x, y, t, p, B, sensor_size=(H, W))
                if got.dtype != np.float64:
                    return desc, "dtype %s" % got.dtype
                return desc + " B=%d" % B, same(got, ref, None, "grid")
            if which in ("ts_image", "ts_image_torch"):
                x = rng.uniform(0, W + 2, n); y = rng.uniform(0, H + 2, n)
                t = np.sort(rng.uniform(10, 11, n)); p = rng.choice([-1.0, 1.0], n)
                pad = bool(rng.integers(0, 2))
                if which == "ts_image":
                    norm = bool(rng.integers(0, 2))
                    got = E.events_to_timestamp_image(x, y, t, p, sensor_size=(H, W), padding=pad, normalize_timestamps=norm)
                    ref = R.events_to_timestamp_image(x, y, t, p, sensor_size=(H, W), padding=pad, normalize_timestamps=norm, accum="f64")
                else:
                    rev = bool(rng.integers(0, 2))
                    c = [torch.from_numpy(a.astype(np.float32)) for a in (x, y, t - 10.0, p)]
                    got = [g.cpu().numpy() for g in E.events_to_timestamp_image_torch(*c, sensor_size=(H, W), padding=pad, timestamp_reverse=rev)]
                    ref = R.events_to_timestamp_image_torch(*(a.numpy() for a in c), sensor_size=(H, W), padding=pad, timestamp_reverse=rev, accum="f64")
                for k in range(2):
                    # a ratio of two float32 images: where the count is a handful of weights, the quotient carries their rounding
                    err = same(got[k], ref[k], np.ones_like(ref[k]), "image %d" % k, 2e-5)
                    if err is not None:
                        return desc, err
                return desc, None
            if which == "ts_hard":
                # the one-pass timestamp path (round 6) on what the plain case leaves out: scenes that cut hot tiles, polarities of
                # every kind (zeros -> the non-positive class, NaN -> neither), unsorted / constant time stamps, pixels that wrap
                n = int(rng.choice([2, 65, 8193, 60_000, 400_000, 1_000_003, 2_500_000]))
                impl = str(rng.choice(["auto", "tiled"] + (["direct"] if n <= 100_000 else [])))
                os.environ["EVK_IMPL"] = impl
                scene = str(rng.choice(["uniform", "blob", "pixel", "edge"]))
                tk, pk = str(rng.choice(["sorted", "const", "few", "unsorted"])), str(rng.choice(["pm1", "pm1z", "float", "special"]))
                x, y = coords(rng, n, H + 1, W + 1, True, scene)
                if n > 100:
                    k = n // 50
                    x[:k] = rng.uniform(-1, 0, k).astype(np.float32)                  # px = -1: wraps to the last column
                    y[k:2 * k] = rng.uniform(-1, 0, k).astype(np.float32)
                    x[2 * k:3 * k] = rng.uniform(W, W + 3, k).astype(np.float32)      # clipped: pixel (0, 0) with its weights
                t, pw = times(rng, n, tk), weights(rng, n, pk)
                rev = bool(rng.integers(0, 2))
                desc = "misc ts_hard %dx%d n=%d %s t=%s p=%s rev=%d impl=%s" % (H, W, n, scene, tk, pk, rev, impl)
                c = device_cols(rng, (x, y, t, pw))
                got = [g.cpu().numpy() for g in E.events_to_timestamp_image_torch(*c, sensor_size=(H, W), timestamp_reverse=rev)]
                ref = R.events_to_timestamp_image_torch(x, y, t, pw, sensor_size=(H, W), timestamp_reverse=rev, accum="f64")
                for k in range(2):
                    err = same(got[k], ref[k], np.ones_like(ref[k]), "image %d" % k, 2e-5)
                    if err is not None:
                        return desc, err
                return desc, None
            if which == "gather":
                img = rng.normal(size=(H, W))
                x = rng.uniform(0, W + 1, n); y = rng.uniform(0, H + 1, n)
                got, ref = E.image_to_event_weights(x, y, img), R.image_to_event_weights(x, y, img)
                return desc, None if (got.dtype == np.float64 and np.array_equal(got, ref)) else "gather not bit-exact"
            # batch: evaluate_function_batch at K flows against K oracle evaluations
            H, W = max(H, 32), max(W, 32)
            x = rng.uniform(1, W - 1, n).astype(np.float32).astype(np.float64); y = rng.uniform(1, H - 1, n).astype(np.float32).astype(np.float64)
            t = np.sort(rng.uniform(0, 0.1, n).astype(np.float32).astype(np.float64)); p = rng.choice([-1.0, 1.0], n)
            K = int(rng.integers(1, 8))
            flows = [rng.normal(0, 1, 2) * float(rng.choice([5.0, 100.0, 1500.0])) for _ in range(K)]
            eo, ro = E.variance_objective(), R.variance_objective()
            eo.sensor_size = ro.sensor_size = (H, W); ro.accum = "f64"
            ev = DeviceEvents.from_arrays(x, y, t, p)
            got = eo.evaluate_function_batch(flows, ev, None, None, None, E.linvel_warp(), (H, W), 1.0)
            gnum = eo.evaluate_numeric_gradient(flows[0], ev, None, None, None, E.linvel_warp(), (H, W), 1.0, epsilon=1.0)
            ref = [float(ro.evaluate_function(q, x, y, t, p, R.linvel_warp(), (H, W), blur_sigma=1.0)) for q in flows]
            for k in range(K):
                if abs(float(got[k]) - ref[k]) > 3e-5 * abs(ref[k]) + 1e-12:
                    return desc + " K=%d" % K, "flow %d %s: %.9g vs %.9g" % (k, flows[k], float(got[k]), ref[k])
            f1 = [float(ro.evaluate_function(flows[0] + np.eye(2)[i], x, y, t, p, R.linvel_warp(), (H, W), blur_sigma=1.0)) for i in range(2)]
            rnum = np.array([f1[0] - ref[0], f1[1] - ref[0]])
            if np.max(np.abs(np.asarray(gnum, np.float64) - rnum)) > 1e-4 * abs(ref[0]) + 1e-12:     # a difference of two float32 values
                return desc, "numeric gradient %s vs %s" % (gnum, rnum)
            return desc + " K=%d" % K, None
    except Exception as e:  # noqa: BLE001
        return desc, "raised %s: %s" % (type(e).__name__, e)
    finally:
        os.environ.pop("EVK_IMPL", None)


N_IWE = [1, 2, 64, 1000, 8193, 149_999, 150_001, 400_000, 1_200_000]


def iwe_inputs(rng):
    sensor = None if rng.random() < 0.2 else ([(180, 240), (260, 346), (480, 640), (720, 1280)][int(rng.integers(0, 4))]
                                               if rng.random() < 0.4 else (int(rng.integers(8, 720)), int(rng.integers(8, 1280))))
    canvas = (180, 240) if sensor is None else sensor
    img_size = canvas if rng.random() < 0.5 else (max(2, canvas[0] + int(rng.integers(-10, 30))), max(2, canvas[1] + int(rng.integers(-10, 30))))
    n = int(rng.choice(N_IWE))
    H, W = canvas
    scene = str(rng.choice(["uniform", "uniform", "blob", "edge", "pixels"]))
    x = rng.uniform(-2, W + 2, n); y = rng.uniform(-2, H + 2, n)
    if scene == "blob" and n > 8:
        hot = rng.random(n) < 0.7
        x[hot] = W / 2 + rng.uniform(-3, 3, hot.sum()); y[hot] = H / 3 + rng.uniform(-3, 3, hot.sum())
    elif scene == "edge" and n > 8:
        hot = rng.random(n) < 0.8
        x[hot] = W * 0.37 + rng.normal(0, 0.6, hot.sum())
    elif scene == "pixels":        # sensor events: integer pixels
        x, y = np.floor(np.clip(x, 0, W - 1)), np.floor(np.clip(y, 0, H - 1))
    T = float(rng.choice([0.05, 0.2, 1.0]))
    t = np.sort(rng.uniform(0, T, n))
    f32 = rng.random() < 0.75
    if f32:          # float32-representable columns -> float32 device columns; else float64 ones
        x, y, t = (a.astype(np.float32).astype(np.float64) for a in (x, y, t))
        t = np.sort(t)
    pk = str(rng.choice(["pm1", "pm1", "float", "x100"]))
    p = rng.choice([-1.0, 1.0], n) * (100.0 if pk == "x100" else 1.0)
    if pk == "float":
        p = rng.normal(size=n).astype(np.float32).astype(np.float64)
    scale = float(rng.choice([0.0, 10.0, 100.0, 1000.0]))
    prm = rng.normal(0, 1, 2) * scale
    impl = str(rng.choice(["auto", "tiled", "direct"]))
    desc = "sensor=%s img_size=%s n=%d %s f32=%d p=%s T=%g prm=(%.1f, %.1f) impl=%s" % (sensor, img_size, n, scene, f32, pk, T, prm[0], prm[1], impl)
    return desc, sensor, img_size, n, scene, x, y, t, p, prm, impl


def case_iwe(rng):
    desc, sensor, img_size, n, scene, x, y, t, p, prm, impl = iwe_inputs(rng)
    grad, pol = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    desc = "iwe grad=%d pol=%d %s" % (grad, pol, desc)
    with np.errstate(all="ignore"):
        ri, rd = R.get_iwe(prm, x, y, t, p, R.linvel_warp(), img_size, compute_gradient=grad, use_polarity=pol,
                           sensor_size=sensor, accum="f64")
        mi, _ = R.get_iwe(prm, x, y, t, p, R.linvel_warp(), img_size, compute_gradient=False, use_polarity=False,
                          sensor_size=sensor, accum="f64")
    os.environ["EVK_IMPL"] = impl
    try:
        iwe, diwe = E.get_iwe(prm, x, y, t, p, E.linvel_warp(), img_size, compute_gradient=grad, use_polarity=pol, sensor_size=sensor)
    except Exception as e:  # noqa: BLE001
        return desc, "raised %s: %s" % (type(e).__name__, e)
    finally:
        os.environ.pop("EVK_IMPL", None)
    hot = scene in ("blob", "edge")
    # (float64 columns that are not float32-representable stay float64 on the device and take the direct kernels)
    direct_like = impl == "direct" or (impl == "auto" and n < tiled.TILED_MIN_EVENTS_IWE) or "f32=0" in desc
    err = same(iwe, ri, mi, "iwe", 1e-4 if direct_like and hot else 4e-7)
    if err is None and grad:
        if diwe is None:
            return desc, "d_iwe is None"
        # the derivative image's magnitudes: |p| * |t_ref - t| <= T per event and weight -- bounded through the IWE of |p|
        dm = np.broadcast_to(np.asarray(mi, np.float64) * float(t[-1] - t[0]) * 4.0, np.asarray(rd).shape)
        err = same(diwe, rd, dm, "d_iwe", 1e-4 if direct_like and hot else 4e-7)
    elif err is None and diwe is not None:
        err = "d_iwe returned without compute_gradient"
    return desc, err


def case_objective(rng):
    from event_utils_amd.contrast_max import objectives as O
    from event_utils_amd.events import DeviceEvents
    desc, sensor, img_size, n, scene, x, y, t, p, prm, impl = iwe_inputs(rng)
    if n < 64:
        n = 1000; x, y, t, p = (np.resize(a, n) for a in (x, y, t, p)); t = np.sort(t)
    name = str(rng.choice(["variance", "variance", "variance", "sos", "soe", "moa", "sosa", "r1", "rms"]))
    sigma = [None, 0.0, 1.0, 2.0][int(rng.integers(0, 4))]
    resident = bool(rng.integers(0, 2))
    desc = "objective %s sigma=%s resident=%d %s" % (name, sigma, resident, desc)
    mk = {"variance": "variance_objective", "sos": "sos_objective", "soe": "soe_objective", "moa": "moa_objective",
          "sosa": "sosa_objective", "r1": "r1_objective", "rms": "rms_objective"}[name]
    ro, eo = getattr(R, mk)(), getattr(O, mk)()
    ro.sensor_size, ro.accum = sensor, "f64"
    eo.sensor_size = sensor
    with np.errstate(all="ignore"):
        rf = float(ro.evaluate_function(prm, x, y, t, p, R.linvel_warp(), img_size, blur_sigma=sigma))
        rg = np.asarray(ro.evaluate_gradient(prm, x, y, t, p, R.linvel_warp(), img_size, blur_sigma=sigma), np.float64) \
            if getattr(ro, "has_derivative", True) and hasattr(ro, "evaluate_gradient") else None
    os.environ["EVK_IMPL"] = impl
    try:
        args = (DeviceEvents.from_arrays(x, y, t, p), None, None, None) if resident else (x, y, t, p)
        f = float(eo.evaluate_function(prm, *args, E.linvel_warp(), img_size, blur_sigma=sigma))
        g = np.asarray(eo.evaluate_gradient(prm, *args, E.linvel_warp(), img_size, blur_sigma=sigma), np.float64) \
            if rg is not None and eo.has_derivative else None
    except Exception as e:  # noqa: BLE001
        return desc, "raised %s: %s" % (type(e).__name__, e)
    finally:
        os.environ.pop("EVK_IMPL", None)
    if not np.isfinite(rf) or not np.isfinite(f):
        return desc, None if (np.isnan(rf) and np.isnan(f)) or rf == f else "f %r vs %r" % (f, rf)
    # float32 images summed / squared: 2e-5 relative (as tests/test_gpu_parity.py for the other objectives); tiny objectives
    # (a handful of events on a large canvas) get the absolute floor of the float32 image they are computed from
    # (soe = sum of exp(image): a relative error of the image's largest value times that value)
    rel = 3e-5 * (max(1.0, abs(np.log(max(abs(rf), 1e-300)))) if name == "soe" else 1.0)
    if abs(f - rf) > rel * abs(rf) + 1e-12:
        return desc, "f %.9g vs %.9g" % (f, rf)
    if g is not None:
        if g.shape != rg.shape:
            return desc, "gradient shape %s vs %s" % (g.shape, rg.shape)
        direct_like = impl == "direct" or (impl == "auto" and n < tiled.TILED_MIN_EVENTS_IWE) or "f32=0" in desc
        gtol = 1e-3 if direct_like and scene in ("blob", "edge") else 1e-4     # float32 atomics on hot pixels, as the reference's
        gtol *= max(1.0, abs(np.log(max(abs(rf), 1e-300)))) if name == "soe" else 1.0
        if np.isfinite(rg).all() and np.max(np.abs(g - rg)) > gtol * np.max(np.abs(rg)) + 1e-6 * abs(rf) + 1e-12:
            return desc, "gradient %s vs %s" % (g, rg)
    return desc, None


if __name__ == "__main__":
    budget = float(arg("--seconds", "240"))
    seed = int(arg("--seed0", "0"))
    kinds = arg("--kinds", "voxel,image,native,iwe,objective,windows,misc,errors,prims,search").split(",")
    fns = {"voxel": case_voxel, "image": case_image, "native": case_native, "iwe": case_iwe, "objective": case_objective,
           "windows": case_windows, "misc": case_misc, "errors": case_errors, "prims": case_prims, "search": case_search}
    t0, done, failed = time.time(), {k: 0 for k in kinds}, []
    while time.time() - t0 < budget:
        kind = kinds[seed % len(kinds)]
        rng = np.random.default_rng(900_000 + seed)
        desc, err = fns[kind](rng)
        done[kind] += 1
        if err is not None:
            failed.append((seed, desc, err))
            print("FAIL seed %d: %s -> %s" % (seed, desc, err), flush=True)
        seed += 1
    print("cases", done, "failures", len(failed), "next seed", seed)
    sys.exit(1 if failed else 0)
