"""Differential fuzzing of the public voxel / event-image calls against the CPU oracle (oracle/reference_np.py), beyond the
sizes and seeds of tests/: sensor sizes up to 1300 x 800, event counts around every boundary of the one-pass path (one wave,
one sub-chunk, the 'auto' thresholds, several sub-chunks per workgroup), scenes that cut hot tiles, polarities of every kind
(+-1, zeros, small integers, float32, huge, NaN / infinite), time stamps that are constant / few-valued / unsorted, every
EVK_IMPL.  Test infrastructure (imports the oracle): not part of the product.
usage: python tools/fuzz_parity.py [--seconds S] [--seed0 K] [--kinds voxel,image,native]     exit code 1 on any mismatch"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import event_utils_amd as E  # noqa: E402
from event_utils_amd import tiled  # noqa: E402
from oracle import reference_np as R  # noqa: E402

N_CHOICES = [1, 2, 63, 64, 65, 1000, 8191, 8192, 8193, 12_289, 79_999, 80_001, 319_999, 320_001, 350_001, 1_000_003, 2_500_000]


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
        cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
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
        got = E.events_to_image_torch(torch.from_numpy(xa).cuda(), torch.from_numpy(ya).cuda(), torch.from_numpy(p).cuda(),
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


if __name__ == "__main__":
    budget = float(arg("--seconds", "240"))
    seed = int(arg("--seed0", "0"))
    kinds = arg("--kinds", "voxel,image,native").split(",")
    fns = {"voxel": case_voxel, "image": case_image, "native": case_native}
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
