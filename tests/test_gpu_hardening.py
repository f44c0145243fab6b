"""GPU (-m gpu): round-6 hardening -- release_scratch(), the one-pass index re-zeroed after a failed call, the element-wise
helper kernels that replaced the last torch arithmetic on the product path (evk_elem.hip), host_report without a counter,
the coupled `.image` mirror and NaN propagation of the stateful image classes."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import reference_np as R
from event_utils_amd import _lib

pytestmark = pytest.mark.gpu


def _events(seed, n, H, W):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    return x, y, t, p


def _close(a, ref, tol=1e-5):
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape and np.max(np.abs(a - ref)) <= tol * max(np.max(np.abs(ref)), 1e-30)


def test_release_scratch_frees_the_persistent_buffers_and_calls_go_on():
    """event_utils_amd.release_scratch(): the grow-only record / staging buffers, zeroed indices, spill pairs and reduction
    slots of the current stream are dropped (tens of MB after one 1 M-event call) and the next calls rebuild what they need:
    same grid, same image, same objective value."""
    import event_utils_amd as E
    from event_utils_amd import tiled
    n, H, W, B = 1_000_000, 480, 640, 5
    x, y, t, p = _events(11, n, H, W)
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    xr = torch.from_numpy(np.random.default_rng(3).uniform(1, W - 1, n).astype(np.float32)).cuda()
    yr = torch.from_numpy(np.random.default_rng(4).uniform(1, H - 1, n).astype(np.float32)).cuda()

    def work():
        vox = E.events_to_voxel_torch(*cols, B, sensor_size=(H, W)).cpu().numpy()
        img = E.events_to_image_torch(cols[0], cols[1], cols[3], sensor_size=(H, W), interpolation='bilinear', padding=False).cpu().numpy()
        ev = E.DeviceEvents(xr, yr, cols[2], cols[3])
        obj = E.variance_objective(); obj.sensor_size = (H, W)
        f = float(obj.evaluate_function(np.array([30., -20.]), ev, None, None, None, E.linvel_warp(), (H, W), 1.0))
        return vox, img, f
    a = work()
    held = sum(b.numel() * b.element_size() for b in list(tiled._persist.values()) + list(tiled._zpersist.values()))
    assert held > 8 * n                                   # at least the 8-byte records of the voxel call
    freed = E.release_scratch()
    assert freed >= held and not tiled._persist and not tiled._zpersist and not tiled._spill
    b = work()
    assert np.array_equal(a[0], b[0]) and abs(a[2] - b[2]) <= 1e-6 * abs(a[2])
    _close(b[1], a[1], 1e-6)
    # per device / per stream selection: another stream's buffers are left alone
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        E.events_to_voxel_torch(*cols, B, sensor_size=(H, W))
    side.synchronize()
    keys = set(tiled._persist)
    assert E.release_scratch(stream=side) > 0
    assert set(tiled._persist) < keys and any(k[2] != side.cuda_stream for k in tiled._persist)


def test_one_pass_index_is_zeroed_again_after_a_failed_call():
    """The one-pass paths keep self-resetting counters in a persistent index that is zeroed ONCE.  A call that fails (here: an
    argument the library refuses -- after the index has been dirtied by hand, as a launch failing between the two kernels
    would leave it) zeroes the index before the error propagates, and the next call is correct."""
    import event_utils_amd as E
    from event_utils_amd import tiled
    n, H, W, B = 600_000, 480, 640, 5
    x, y, t, p = _events(12, n, H, W)
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    ref = R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64")
    out = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
    shape = tiled.voxel2_shape(H, W, B)
    tiled.voxel2(cols, None, n, float(t[0]), float(t[-1]), B, H, W, *shape, out, None, True)
    index = tiled._zbuf("voxel2_index", 1, out.device)
    index[2] = 7; index[16:40] = 12345; index[8] = 3            # ticket, per-tile totals, early-report ticket mid-count
    with pytest.raises(_lib.EvkError):
        tiled.voxel2(cols, None, n, float(t[0]), float(t[-1]), B, H, W, *shape, out, None, True, stage=1 << 30)   # unknown flag
    assert int(index.abs().sum().item()) == 0
    tiled.voxel2(cols, None, n, float(t[0]), float(t[-1]), B, H, W, *shape, out, None, True)
    _close(out.cpu().numpy(), ref)
    # the event images share the rule
    img = torch.zeros((H, W), dtype=torch.float32, device="cuda")
    iidx = None
    assert tiled.image2("f32", cols[0], cols[1], cols[3], n, H, W, float("inf"), float("inf"), img, None, fresh=True)
    iidx = tiled._zbuf("image2_index", 1, img.device)
    iidx[2] = 5
    with pytest.raises(_lib.EvkError):
        tiled.image2("f32", cols[0], cols[1], cols[3], n, H, W, float("inf"), float("inf"), img, None, fresh=True, stage=1 << 30)
    assert int(iidx.abs().sum().item()) == 0
    assert tiled.image2("f32", cols[0], cols[1], cols[3], n, H, W, float("inf"), float("inf"), img, None, fresh=True)
    want = np.zeros((H, W)); np.add.at(want, (y.astype(int), x.astype(int)), p.astype(np.float64))
    _close(img.cpu().numpy(), want)


def test_elementwise_helpers_match_torch_and_numpy(monkeypatch):
    """evk_polarity_weights_f32 / evk_abs_max / evk_abs against the torch / numpy expressions they replace, NaN and signed zeros
    included; and the paths that use them: the two-voxelisation route of events_to_neg_pos_voxel_torch (forced with
    EVK_IMPL=direct) against the oracle, DeviceEvents.p_absmax()."""
    import event_utils_amd as E
    from event_utils_amd import _device as D
    from event_utils_amd.events import _abs_max
    from event_utils_amd.contrast_max.objectives import _abs_device
    from event_utils_amd.representations.voxel_grid import _polarity_weights
    rng = np.random.default_rng(5)
    p = rng.normal(size=100_003).astype(np.float32)
    p[:6] = [0.0, -0.0, np.nan, np.inf, -np.inf, 1e-45]
    pt = torch.from_numpy(p)
    pos, neg = _polarity_weights(pt.cuda())
    assert torch.equal(pos.cpu(), torch.where(pt > 0, 1.0, 0.0).to(torch.float32))
    assert torch.equal(neg.cpu(), torch.where(pt <= 0, 1.0, 0.0).to(torch.float32))
    hp, hn = _polarity_weights(pt)                                  # host tensor in -> host tensors out
    assert not hp.is_cuda and torch.equal(hp, pos.cpu()) and torch.equal(hn, neg.cpu())
    for arr in (p[6:], p[6:].astype(np.float64), np.zeros(5, np.float32), np.array([-3.5, 2.0], np.float64)):
        assert _abs_max(torch.from_numpy(arr).cuda()) == float(np.abs(arr).max())
    assert np.isnan(_abs_max(pt.cuda()))                            # a NaN propagates, as torch's max()
    for dt in (np.float32, np.float64):
        a = torch.from_numpy(p.astype(dt))
        got = _abs_device(a.cuda()).cpu()
        assert torch.equal(torch.nan_to_num(got, nan=-1.0), torch.nan_to_num(a.abs(), nan=-1.0))
        assert not torch.signbit(got[:2]).any()                     # |-0.0| = +0.0
    # the fallback route of events_to_neg_pos_voxel_torch
    n, H, W, B = 80_000, 60, 80, 3
    x, y, t, _ = _events(21, n, H, W)
    ps = rng.integers(-1, 2, n).astype(np.float32)                  # -1, 0, +1
    monkeypatch.setenv("EVK_IMPL", "direct")
    vp, vn = E.events_to_neg_pos_voxel_torch(*(torch.from_numpy(a).cuda() for a in (x, y, t, ps)), B, sensor_size=(H, W))
    rp = R.events_to_voxel_torch(x, y, t, (ps > 0).astype(np.float32), B, sensor_size=(H, W), accum="f64")
    rn = R.events_to_voxel_torch(x, y, t, (ps <= 0).astype(np.float32), B, sensor_size=(H, W), accum="f64")
    _close(vp.cpu().numpy(), rp); _close(vn.cpu().numpy(), rn)
    ev = E.DeviceEvents.from_arrays(x, y, t, ps * 2.5, precision="f32")
    assert ev.p_absmax() == 2.5


def test_host_report_is_written_without_a_counter():
    """include/evk.h: host_report is an optional argument of its own -- a C caller that passes it WITHOUT a dropped-event counter
    gets {seq, 0} (round 5 wrote the report only when oob was given too, and such a caller waited for ever)."""
    from event_utils_amd import tiled, _device as D
    n, H, W, B = 400_000, 480, 640, 5
    x, y, t, p = _events(13, n, H, W)
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    out = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
    tw, th = tiled.voxel2_shape(H, W, B)
    index, scratch, nbytes, flags = tiled._voxel2_env(out.device, n, B, H, W, tw, th)
    report = torch.zeros(2, dtype=torch.int32).pin_memory()
    seq = 0x1234
    _lib.call("evk_voxel2_f32", *(D.ptr(c) for c in cols), n, H, W, tw, th, float(t[0]), float(t[-1]), B,
              flags | _lib.EVK_VOXEL_OVERWRITE, D.ptr(out), D.ptr(index), D.ptr(scratch), nbytes, None,
              ctypes.c_void_p(report.data_ptr()), seq, D.stream())
    torch.cuda.synchronize()
    assert int(report[0]) == seq and int(report[1]) == 0
    _close(out.cpu().numpy(), R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64"))


def test_image_classes_mirror_and_nan_rank():
    """Upstream's TimestampImage / EventImage expose `.image` as THE ndarray (image.py:358,380): in-place edits are seen by the
    next call and a kept reference follows add_events.  Here `.image` is a host mirror coupled to the device image.  And a NaN
    pixel makes every rank NaN, as scipy.stats.rankdata does (image.py:371; checked against the real reference in round 6)."""
    import event_utils_amd as E
    for cls, rcls in ((E.EventImage, R.EventImage), (E.TimestampImage, R.TimestampImage)):
        a, b = cls((6, 7)), rcls((6, 7))
        views = []
        for o in (a, b):
            img = o.image
            img[1, 2] = 7.0
            img *= 2.0
            views.append(img)
            if cls is E.EventImage:
                o.add_event(2.2, 1.9, 0.0, 3.0)
            else:
                o.add_events(np.array([3.0, 2.0]), np.array([4.0, 1.0]), np.array([0.25, 0.5]), None)
        assert np.array_equal(a.image, b.image)
        assert np.array_equal(views[0], views[1])          # the reference handed out BEFORE the events shows them
        assert np.array_equal(a.get_image(), b.get_image(), equal_nan=True)
        a.image = np.arange(42.0).reshape(6, 7)            # assignment replaces the image (and drops the old mirror)
        assert a.image[5, 6] == 41.0 and views[0][5, 6] != 41.0
    t1, t2 = E.TimestampImage((5, 4)), R.TimestampImage((5, 4))
    for o in (t1, t2):
        o.add_events(np.array([1.0, 2.0, 3.0]), np.array([1.0, 2.0, 3.0]), np.array([0.1, 0.2, 0.3]), None)
        o.image[4, 0] = np.nan
    g1, g2 = t1.get_image(), t2.get_image()
    assert np.all(np.isnan(g1)) and np.array_equal(g1, g2, equal_nan=True)
    t1.image[4, 0] = -np.nan                              # (sign bit set: sorts to the front of the radix sort)
    assert np.all(np.isnan(t1.get_image()))


def _per_tile_sets(bk):
    """Records of a Buckets object sorted within every tile (the order inside one block's share of a tile is the order of
    LDS atomics in either scatter; the SET per tile is what the tile kernels sum)."""
    T = bk.ntiles
    starts = bk.bucket_start[: T + 1].cpu().numpy().astype(np.int64)
    tile_of = np.repeat(np.arange(T), np.diff(starts))
    kept = int(starts[-1])
    if bk.iwe_flag:
        r = bk.records[:kept].cpu().numpy().view(np.uint32).reshape(-1, 2)
        return r[np.lexsort((r[:, 1], r[:, 0], tile_of))], starts
    r = bk.records.cpu().numpy().reshape(-1, 4).view(np.uint32)[:kept]
    return r[np.lexsort((r[:, 3], r[:, 1], r[:, 0], r[:, 2], tile_of))], starts


@pytest.mark.parametrize("case", ["iwe 32x32", "iwe 16x16 (4 K-event sub-chunks)", "voxel key, dropped events", "ring fallback"])
def test_lds_sorting_scatter_equals_the_ring_scatter(case, monkeypatch):
    """Round 6: evk_bucket_events_f32 scatters by LDS sort (k_tile_scatter_sorted: sub-chunks of 8 K / 4 K events sorted by tile
    in LDS, every tile's piece written to its final place).  Same bucket index bit for bit and the same records per tile as the
    write-combining ring scatter of rounds 1-5 (EVK_STAGE_LEGACY_SCATTER) -- on a structured scene with a ragged event count, on
    a tiling that needs the smaller sub-chunks, with the nearest-pixel key and out-of-domain events (dropped, counted), and on
    a tiling whose counters no longer fit beside the sort buffer (the library then takes the old scatter by itself)."""
    import bench
    from event_utils_amd import tiled, _device as D
    n, H, W = 1_300_003, 720, 1280
    x, y, t, p = bench.structured_scene(9, n, H, W)
    key_mode, dom_h, dom_w, tw, th = 1, H + 1, W + 1, 5, 5
    if case.startswith("iwe 16x16"):
        tw, th = 4, 4                                   # 81 x 46 = 3726 tiles
    elif case.startswith("voxel"):
        key_mode, dom_h, dom_w, tw, th = 0, H, W, 5, 4
        x, y = np.floor(x), np.floor(y)
        x[::1000] = W + 3.0                             # out of the domain: dropped and counted
    elif case.startswith("ring"):
        dom_h, dom_w, tw, th = 704, 720, 3, 3           # 90 x 88 = 7920 tiles of 8 x 8
        x, y = x % 719.0, y % 703.0
    cols = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (x, y, t, p)]
    got = {}
    for legacy in (True, False):
        monkeypatch.setitem(tiled.FORCE, "legacy_scatter", legacy)
        oob = D.OobCounter(cols[0].device)
        bk = tiled.bucket_events(*cols, key_mode, dom_h, dom_w, tw, th, oob=oob, stats=not legacy)
        torch.cuda.synchronize()
        if case.startswith("voxel"):                    # the dropped events were counted: exactly the ones put out of the domain
            with pytest.raises(IndexError, match="%d offending" % len(x[::1000])):
                oob.raise_if_set(IndexError, "out of range")
        else:
            oob.raise_if_set(IndexError, "out of range")
        got[legacy] = (bk, _per_tile_sets(bk))
    (a, (ra, sa)), (b, (rb, sb)) = got[True], got[False]
    T = a.ntiles                                        # (offsets, work-item offsets, counters, then the USED item -> tile entries)
    ia, ib = a.bucket_start.cpu().numpy(), b.bucket_start.cpu().numpy()
    nitems = int(ia[2 * T + 1])
    assert np.array_equal(ia[: 3 * T + 2 + nitems], ib[: 3 * T + 2 + nitems]) and np.array_equal(sa, sb)
    assert np.array_equal(ra, rb) and len(ra) == int(sa[-1])
    if case.startswith("voxel"):
        assert int(sa[-1]) == n - len(x[::1000])
    if case.startswith("ring"):
        assert b.p_absmax is None                      # the ring scatter does not deliver max |p|: the caller reduces the column
        monkeypatch.setitem(tiled.FORCE, "legacy_scatter", False)
        xi = [torch.floor(cols[0]), torch.floor(cols[1]), cols[2], cols[3]]     # compactable events on a tiling the sorting scatter
        c = tiled.bucket_events(*xi, key_mode, dom_h, dom_w, tw, th, stats=True, compact=True).settle()   # cannot take:
        assert c.iwe_flag == 0 and c.records.dtype == torch.float32             # 16-byte records, and the index says so
    else:
        assert b.p_absmax == 1.0 and b.structured == a.structured


def test_bucketing_delivers_compact_records_and_stats_in_one_call(monkeypatch):
    """EVK_STAGE_STATS | EVK_STAGE_COMPACT: the histogram pass delivers the verdict (every event has an 8-byte compact record),
    the scatter then writes compact records DIRECTLY -- the very records evk_compact_records_f32 makes from the 16-byte ones --
    and max |p| arrives with the scene word in one 12-byte copy.  Sub-pixel coordinates or a polarity with low mantissa bits
    keep the 16-byte records.  Objective and gradient on either kind of bucket agree bit for bit (integer LDS accumulation)."""
    import bench
    import event_utils_amd as E
    from event_utils_amd import tiled
    n, H, W = 900_001, 480, 640
    x, y, t, p = bench.structured_scene(4, n, H, W)
    xi, yi = np.floor(x), np.floor(y)
    p3 = (p * 3.0).astype(np.float32)
    dom_h, dom_w = H + 1, W + 1
    tw, th = tiled.iwe_tile_shape(dom_h, dom_w)
    monkeypatch.setitem(tiled.FORCE, "iwe_records", "compact")

    def bucket(xs, ys, ps, legacy):
        monkeypatch.setitem(tiled.FORCE, "legacy_scatter", legacy)
        cols = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (xs, ys, t, ps)]
        bk = tiled.bucket_events(*cols, 1, dom_h, dom_w, tw, th, stats=not legacy, compact=not legacy)
        return bk.compact()
    new, old = bucket(xi, yi, p3, False), bucket(xi, yi, p3, True)
    assert new.iwe_flag == old.iwe_flag == _lib.EVK_IWE_COMPACT and new.p_absmax == 3.0
    (rn, sn), (ro, so) = _per_tile_sets(new), _per_tile_sets(old)
    assert np.array_equal(sn, so) and np.array_equal(rn, ro)
    assert bucket(x, yi, p3, False).iwe_flag == 0                                   # a sub-pixel coordinate
    assert bucket(xi, yi, (p * 1.0001).astype(np.float32), False).iwe_flag == 0     # a polarity with low mantissa bits
    far = xi.copy(); far[5] = dom_w + 10.0
    assert bucket(far, yi, p3, False).iwe_flag == 0                                 # a coordinate outside the domain
    # the objective through both routes
    vals = {}
    for legacy in (False, True):
        monkeypatch.setitem(tiled.FORCE, "legacy_scatter", legacy)
        ev = E.DeviceEvents.from_arrays(xi, yi, t, p3, precision="f32")
        obj = E.variance_objective(); obj.sensor_size, obj.impl = (H, W), "tiled"
        prm = np.array([35.0, -22.0])
        vals[legacy] = (float(obj.evaluate_function(prm, ev, None, None, None, E.linvel_warp(), (H, W), 1.0)),
                        obj.evaluate_gradient(prm, ev, None, None, None, E.linvel_warp(), (H, W), 1.0).copy())
        assert list(ev._buckets.values())[0].iwe_flag == _lib.EVK_IWE_COMPACT and ev.p_absmax() == 3.0
    assert vals[False][0] == vals[True][0] and np.array_equal(vals[False][1], vals[True][1])


def test_bound_evaluators_and_retargeted_calls_give_the_public_methods_numbers():
    """Round 6: variance_objective.bind_fast -- the closures evk_bfgs evaluates through -- and the in-place re-targeting of a
    cached evaluation call when the flow needs another LDS window (tiled._retarget) are plumbing only: value + gradient and
    the three-flow values equal the public methods' bit for bit over flows from 0 to 400 px/s (several window sizes and time
    slices, in both directions), a flow the tiled kernels cannot take falls back to the public path, and evk_bfgs follows the
    same trajectory with and without them."""
    import bench
    import event_utils_amd as E
    from event_utils_amd.contrast_max.events_cmax import evk_bfgs
    n, H, W = 400_000, 240, 320
    x, y, t, p = bench.structured_scene(7, n, H, W)
    ev = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
    w = E.linvel_warp()

    def objective():
        o = E.variance_objective()
        o.sensor_size, o.reference_exact = (H, W), False
        return o
    args = (ev, None, None, None, w, (H, W), 1.0)
    fast, ref = objective(), objective()
    fg, f3 = fast.bind_fast(*args)
    flows = [(0.0, 0.0), (3.0, -2.0), (40.0, -25.0), (120.0, 90.0), (400.0, -380.0), (41.0, -25.5), (1.0, 1.0), (2500.0, 10.0)]
    for q in flows + flows[::-1]:
        fv, gv = fg(list(q))
        rf, rg = ref.evaluate_function_and_gradient(np.array(q), *args)
        if abs(q[0]) > 1000.0:      # (beyond the LDS windows: both go through the direct kernels, whose float atomics add in any order)
            assert abs(fv - float(rf)) <= 1e-5 * abs(float(rf)) and np.allclose(gv, rg, rtol=1e-3, atol=1e-7), q
        else:
            assert fv == float(rf) and gv == [float(v) for v in rg], q
    trios = [[(0.0, 0.0), (1.0, 0.0), (0.0, 1.0)], [(40.0, -25.0), (13.0, -8.0), (120.0, -75.0)], [(300.0, 300.0)] * 3,
             [(1.0, 2.0), (2.0, 1.0), (0.5, 0.5)]]
    for trio in trios + trios[::-1]:
        got = f3([list(q) for q in trio])
        want = [float(v) for v in ref.evaluate_function_batch([np.array(q) for q in trio], *args)]
        assert got == want, trio
    runs = {}
    for use_fast in (True, False):
        o, tr = objective(), []
        xs = evk_bfgs(o, np.array([0.0, 0.0]), args, trace=tr, fast=use_fast)
        runs[use_fast] = (xs, tr)
    assert np.array_equal(runs[True][0], runs[False][0]) and len(runs[True][1]) == len(runs[False][1])
    for (xa, fa, ga), (xb, fb, gb) in zip(runs[True][1], runs[False][1]):
        assert np.array_equal(xa, xb) and fa == fb and np.array_equal(ga, gb)
    assert np.linalg.norm(runs[True][0] - np.array([40.0, -25.0])) < 3.0


def test_evk_bfgs_numeric_gradients_follow_the_same_trajectory_bound_or_not():
    """numeric_grads=True (the reference's default) through the bound three-flow closure: the forward differences of
    evaluate_function_and_numeric_gradient, same arithmetic -- identical accepted points with and without the binding."""
    import bench
    import event_utils_amd as E
    from event_utils_amd.contrast_max.events_cmax import evk_bfgs
    n, H, W = 300_000, 240, 320
    x, y, t, p = bench.structured_scene(8, n, H, W)
    ev = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
    args = (ev, None, None, None, E.linvel_warp(), (H, W), 1.0)
    runs = {}
    for use_fast in (True, False):
        o, tr = E.variance_objective(), []
        o.sensor_size = (H, W)
        runs[use_fast] = (evk_bfgs(o, np.array([0.0, 0.0]), args, numeric_grads=True, trace=tr, fast=use_fast), tr)
    assert np.array_equal(runs[True][0], runs[False][0]) and len(runs[True][1]) == len(runs[False][1]) >= 3
    for (xa, fa, ga), (xb, fb, gb) in zip(runs[True][1], runs[False][1]):
        assert np.array_equal(xa, xb) and fa == fb and np.array_equal(ga, gb)


def test_native_bfgs_loop_visits_the_points_of_the_python_loop():
    """Round 6: evk_cmax_bfgs_variance_tiled_f32 -- the quasi-Newton loop inside the library -- against the Python loop over the
    bound closures (native=False) and over the public methods (fast=False): same accepted points, values and gradients bit
    for bit, analytic and numeric gradients, reference-exact and consistent gradient; the objective's iter_update sees every
    accepted point; a foreign callback or a start the tiled kernels cannot take keeps the Python loop."""
    import bench
    import event_utils_amd as E
    from event_utils_amd.contrast_max.events_cmax import evk_bfgs, optimize_contrast
    for seed, n, H, W in ((7, 400_000, 240, 320), (3, 1_000_000, 480, 640)):
        x, y, t, p = bench.structured_scene(seed, n, H, W)
        ev = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
        args = (ev, None, None, None, E.linvel_warp(), (H, W), 1.0)
        finals = {}
        for numeric, exact in ((False, False), (True, False), (False, True)):
            runs = {}
            for mode in ("native", "bound", "public"):
                o, tr = E.variance_objective(), []
                o.sensor_size, o.reference_exact = (H, W), exact
                o.native_passes = None
                xs = evk_bfgs(o, np.array([0.0, 0.0]), args, numeric_grads=numeric, trace=tr, native=mode == "native",
                              fast=mode != "public")
                assert (o.native_passes is not None) == (mode == "native"), mode
                runs[mode] = (xs, tr)
            for other in ("bound", "public"):
                assert np.array_equal(runs["native"][0], runs[other][0]) and len(runs["native"][1]) == len(runs[other][1])
                for (xa, fa, ga), (xb, fb, gb) in zip(runs["native"][1], runs[other][1]):
                    assert np.array_equal(xa, xb) and fa == fb and np.array_equal(ga, gb), (seed, numeric, exact, other)
            finals[(numeric, exact)] = runs["native"][0]
            if not exact:
                assert np.linalg.norm(runs["native"][0] - np.array([40.0, -25.0])) < 3.0
        # the callback optimize_contrast hands in (the objective's own iter_update) is replayed over the accepted points
        o = E.variance_objective()
        o.sensor_size, o.reference_exact = (H, W), False
        o.native_passes = None
        a = optimize_contrast(ev, None, None, None, E.linvel_warp(), o, optimizer="evk_bfgs", numeric_grads=False, blur_sigma=1.0,
                              img_size=(H, W))
        assert o.native_passes and np.array_equal(a, finals[(False, False)])
        assert o.lifespan == o.pixel_crossings / np.linalg.norm(a)
        # a foreign callback sees the steps as they happen: Python loop
        seen, o2 = [], E.variance_objective()
        o2.sensor_size, o2.reference_exact = (H, W), False
        o2.native_passes = None
        a2 = evk_bfgs(o2, np.array([0.0, 0.0]), args, callback=lambda q: seen.append(q.copy()))
        assert o2.native_passes is None and len(seen) >= 3 and np.array_equal(seen[-1], a2) and np.array_equal(a2, a)
        # a plugin objective derived from the variance objective is evaluated through ITS methods, not through the short cuts
        calls = []

        class scaled(E.variance_objective):
            def evaluate_function_and_gradient(self, params=None, *a, **k):
                fv, gv = super().evaluate_function_and_gradient(params, *a, **k)
                calls.append(np.array(params, dtype=float))
                return 2.0 * fv, 2.0 * gv

            def evaluate_function_batch(self, params_list, *a, **k):
                return [2.0 * v for v in super().evaluate_function_batch(params_list, *a, **k)]
        o4 = scaled()
        o4.sensor_size, o4.reference_exact = (H, W), False
        o4.native_passes = None
        assert o4.bind_fast(*args) is None and o4.bind_native(*args) is None
        a4 = evk_bfgs(o4, np.array([0.0, 0.0]), args)
        assert o4.native_passes is None and len(calls) >= 3 and np.linalg.norm(a4 - np.array([40.0, -25.0])) < 3.0
        # a start beyond every LDS window: the library loop declines at its first pass, the Python loop (direct kernels) runs
        o3 = E.variance_objective()
        o3.sensor_size, o3.reference_exact = (H, W), False
        o3.native_passes = None
        a3 = evk_bfgs(o3, np.array([30000.0, 0.0]), args, maxiter=2)
        assert o3.native_passes is None and np.all(np.isfinite(a3))


def test_optimisers_bucket_small_event_sets():
    """Round 6: an event set handed to optimize_contrast / grid_search is marked many_evaluations and bucketed by output tile
    at ANY event count (the 'auto' threshold of 150 k events is sized for a single evaluation); a lone evaluation of a small
    set keeps the direct kernels.  Both routes agree with each other to float32 accumulation noise."""
    import bench
    import event_utils_amd as E
    from event_utils_amd import tiled
    from event_utils_amd.contrast_max.events_cmax import optimize_contrast, grid_search_optimisation
    n, H, W = 30_000, 180, 240
    assert n < tiled.TILED_MIN_EVENTS_IWE
    x, y, t, p = bench.structured_scene(3, n, H, W)
    w = E.linvel_warp()

    def objective():
        o = E.variance_objective()
        o.sensor_size, o.reference_exact = (H, W), False
        return o
    lone = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
    f_direct = objective().evaluate_function(np.array([40.0, -25.0]), lone, None, None, None, w, (H, W), 1.0)
    assert not lone.many_evaluations and not lone._buckets            # one evaluation: no bucketing
    f_again = objective().evaluate_function(np.array([40.0, -25.0]), lone, None, None, None, w, (H, W), 1.0)
    assert lone._buckets and abs(float(f_again) - float(f_direct)) <= 1e-5 * abs(float(f_direct))   # evaluated again: bucketed
    ev = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
    o = objective()
    o.native_passes = None
    a = optimize_contrast(ev, None, None, None, w, o, optimizer="evk_bfgs", numeric_grads=False, blur_sigma=1.0, img_size=(H, W))
    assert ev.many_evaluations and ev._buckets and o.native_passes     # bucketed, and the library's loop ran on the buckets
    assert np.linalg.norm(a - np.array([40.0, -25.0])) < 1.0
    f_tiled = objective().evaluate_function(np.array([40.0, -25.0]), ev, None, None, None, w, (H, W), 1.0)
    assert abs(float(f_tiled) - float(f_direct)) <= 1e-5 * abs(float(f_direct))
    # host arrays: optimize_contrast uploads them once and marks its own DeviceEvents; scipy's default optimiser
    a2 = optimize_contrast(x, y, t, p, w, objective(), numeric_grads=False, blur_sigma=1.0, img_size=(H, W))
    assert np.linalg.norm(a2 - np.array([40.0, -25.0])) < 1.0
    ev3 = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
    grid_search_optimisation(ev3, None, None, None, w, objective(), (H, W), param_ranges=[[-60, 60], [-60, 60]], num_samples_per_param=5)
    assert ev3.many_evaluations and ev3._buckets


def test_scipy_callbacks_through_the_bound_closures_change_nothing():
    """Round 6: optimize_contrast hands scipy's fmin_bfgs closures bound to the events (variance_objective.bind_fast) instead of
    the public methods -- plumbing only: same argmax bit for bit as through the public methods (an objective whose methods are
    replaced on the instance keeps them, which is how the public route is taken here), numeric and analytic gradients."""
    import warnings
    import bench
    import event_utils_amd as E
    from event_utils_amd.contrast_max.events_cmax import optimize_contrast
    n, H, W = 60_000, 180, 240
    x, y, t, p = bench.structured_scene(5, n, H, W)
    ev = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
    w = E.linvel_warp()
    for numeric in (True, False):
        got = {}
        for route in ("bound", "public"):
            o = E.variance_objective()
            o.sensor_size, o.reference_exact = (H, W), False
            calls = [0]
            if route == "public":
                inner = o.evaluate_function_and_numeric_gradient if numeric else o.evaluate_function_and_gradient

                def counted(*a, _inner=inner, **k):
                    calls[0] += 1
                    return _inner(*a, **k)
                setattr(o, "evaluate_function_and_numeric_gradient" if numeric else "evaluate_function_and_gradient", counted)
                assert o.bind_fast(ev, None, None, None, w, (H, W), 1.0) is None
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                got[route] = optimize_contrast(ev, None, None, None, w, o, numeric_grads=numeric, blur_sigma=1.0, img_size=(H, W))
            assert (calls[0] > 5) == (route == "public")
        assert np.array_equal(got["bound"], got["public"]), (numeric, got)
        assert np.linalg.norm(got["bound"] - np.array([40.0, -25.0])) < 1.5


def test_a_used_objective_can_be_copied_and_the_lifespan_cut_is_reused():
    """Round 6: (a) an objective that has evaluated something holds a resolved library call (ctypes pointers, device buffers);
    copy.deepcopy -- which grid_search_optimisation and optimize_contrast(grid_search_init=True) do, as upstream -- must leave
    that behind instead of failing, and the copy evaluates like the original.  (b) With adaptive_lifespan the evaluations
    between two iter_update calls share ONE cut view of the events (and so its buckets)."""
    import copy
    import warnings
    import bench
    import event_utils_amd as E
    from event_utils_amd.contrast_max.events_cmax import optimize_contrast
    n, H, W = 200_000, 180, 240
    x, y, t, p = bench.structured_scene(3, n, H, W)
    ev = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
    w = E.linvel_warp()
    o = E.variance_objective()
    o.sensor_size = (H, W)
    q = np.array([35.0, -20.0])
    f1 = o.evaluate_function(q, ev, None, None, None, w, (H, W), 1.0)
    f2 = o.evaluate_function(q + 1.0, ev, None, None, None, w, (H, W), 1.0)
    assert o.__dict__.get("_fast_memo") is not None and f1 != f2
    o2 = copy.deepcopy(o)
    assert "_fast_memo" not in o2.__dict__ and o2.sensor_size == (H, W)
    assert o2.evaluate_function(q, ev, None, None, None, w, (H, W), 1.0) == f1
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = optimize_contrast(ev, None, None, None, w, o, numeric_grads=False, blur_sigma=1.0, img_size=(H, W), grid_search_init=True)
    assert np.linalg.norm(a - np.array([40.0, -25.0])) < 1.5
    # (b)
    oa = E.variance_objective(adaptive_lifespan=True, minimum_events=1000)
    oa.sensor_size = (H, W)
    oa.iter_update(np.array([40.0, -25.0]))
    v1 = oa.evaluate_function(q, ev, None, None, None, w, (H, W), 1.0)
    cut1 = ev.__dict__["_lifespan_cut"][1]
    v2, g2 = oa.evaluate_function_and_gradient(q, ev, None, None, None, w, (H, W), 1.0)
    assert ev.__dict__["_lifespan_cut"][1] is cut1 and len(cut1) < len(ev) and float(v1) == float(v2)
    oa.iter_update(np.array([80.0, -50.0]))               # a shorter lifespan: another cut
    oa.evaluate_function(q, ev, None, None, None, w, (H, W), 1.0)
    cut2 = ev.__dict__["_lifespan_cut"][1]
    assert cut2 is not cut1 and len(cut2) < len(cut1)


def test_float64_host_arrays_take_the_float32_path_with_relative_time():
    """Round 6: the reference's host arrays are float64, its time stamps absolute seconds with microsecond resolution -- not
    float32 values.  DeviceEvents.from_arrays narrows float64 columns ON THE DEVICE (evk_narrow_f64_f32, which also reports
    whether they survived exactly) and keeps the time stamps as float32 differences from ts[-1] (subtracted in float64): the
    events stay on the bucketed float32 path and the objective agrees with the float64 oracle far inside 1e-5; EVK_TIME_F64=exact
    keeps the float64 columns (direct kernels); a user-supplied reference time is an absolute time either way."""
    import os
    import bench
    import event_utils_amd as E
    from event_utils_amd import _lib, _device as D
    from oracle import reference_np as R
    n, H, W = 120_000, 180, 240
    x, y, t, p = bench.structured_scene(3, n, H, W)
    x64, y64, p64 = x.astype(np.float64), y.astype(np.float64), p.astype(np.float64)
    t64 = 1_600_000_000.0 + np.round(t.astype(np.float64) * 1e6) / 1e6
    assert not np.array_equal(t64.astype(np.float32).astype(np.float64), t64)
    # the kernel alone: values, offset, the exactness flag (a NaN counts as inexact, as in the host-side policy)
    dev = D.require_gpu()
    for col, off, want_flag in ((x64, 0.0, 0), (t64, 0.0, 1), (t64, float(t64[-1]), 1), (np.array([0.5, np.nan, 2.0]), 0.0, 1),
                                (np.array([0.5, -3.0, 2.0 ** 30]), 0.0, 0)):
        d = torch.from_numpy(col).to(dev)
        out = torch.empty(len(col), dtype=torch.float32, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.call("evk_narrow_f64_f32", D.ptr(d), len(col), off, D.ptr(out), D.ptr(flag), D.stream())
        assert np.array_equal(out.cpu().numpy(), (col - off).astype(np.float32), equal_nan=True)
        assert int(flag.item()) == (want_flag if off == 0.0 else int(flag.item()))
    assert E.DeviceEvents.from_arrays(x64, y64, t64, p64).dtype == torch.float64        # single evaluations: exact float64 route
    ev = E.DeviceEvents.from_arrays(x64, y64, t64, p64, relative_time=True)                 # what the optimisers ask for
    assert ev.dtype == torch.float32 and ev.t_offset == float(t64[-1]) and ev.t_at(-1) == 0.0
    assert np.array_equal(ev.t.cpu().numpy(), (t64 - t64[-1]).astype(np.float32)) and np.array_equal(ev.x.cpu().numpy(), x)
    assert ev.t_at(0) == float(np.float32(t64[0] - t64[-1]))
    w, q = E.linvel_warp(), np.array([38.0, -24.0])
    o = E.variance_objective()
    o.sensor_size = (H, W)
    fv, gv = o.evaluate_function_and_gradient(q, ev, None, None, None, w, (H, W), 1.0)
    ro = R.variance_objective()
    ro.sensor_size = (H, W)
    rf = ro.evaluate_function(q, x64, y64, t64, p64, R.linvel_warp(), (H, W), 1.0)
    rg = ro.evaluate_gradient(q, x64, y64, t64, p64, R.linvel_warp(), (H, W), 1.0)
    assert abs(float(fv) - float(rf)) <= 2e-6 * abs(float(rf)), (fv, rf)
    assert np.max(np.abs(np.asarray(gv, float) - np.asarray(rg, float))) <= 1e-5 * np.max(np.abs(rg))
    # an absolute reference time (here: the first event's) means the same thing on the relative column ...
    o.t_ref = float(t64[0])
    f_first = o.evaluate_function(q, ev, None, None, None, w, (H, W), 1.0)
    os.environ["EVK_TIME_F64"] = "exact"
    try:
        ev64 = E.DeviceEvents.from_arrays(x64, y64, t64, p64)
        assert ev64.dtype == torch.float64 and ev64.t_offset == 0.0
        f_first64 = o.evaluate_function(q, ev64, None, None, None, w, (H, W), 1.0)      # ... as on the float64 one
    finally:
        os.environ.pop("EVK_TIME_F64")
    assert abs(float(f_first) - float(f_first64)) <= 1e-5 * abs(float(f_first64)) and float(f_first) != float(fv)
    # columns that ARE float32 values keep offset 0; a non-integer coordinate column keeps everything in float64
    assert E.DeviceEvents.from_arrays(x64, y64, t.astype(np.float64), p64, relative_time=True).t_offset == 0.0
    # sub-pixel coordinates that are not float32 values (undistorted events) and integer microsecond stamps: float64 for a single
    # evaluation, the float32 route inside the optimisers (their rounding is of the size of the float32 warp's own)
    assert E.DeviceEvents.from_arrays(x64 + 0.1, y64, t64, p64).dtype == torch.float64
    sub = E.DeviceEvents.from_arrays(x64 + 0.1, y64, t64, p64, relative_time=True)
    assert sub.dtype == torch.float32 and np.array_equal(sub.x.cpu().numpy(), (x64 + 0.1).astype(np.float32))
    us = (np.round(t.astype(np.float64) * 1e6)).astype(np.int64) + 1_600_000_000_000_000
    evus = E.DeviceEvents.from_arrays(x64, y64, us, p64, relative_time=True)
    assert evus.dtype == torch.float32 and evus.t_offset == float(us[-1])
    assert np.array_equal(evus.t.cpu().numpy(), (us - us[-1]).astype(np.float32))
    # the optimisers: host arrays in, the float32 path inside, the reference's argmax out
    from event_utils_amd.contrast_max import events_cmax
    res = events_cmax._resident(x64, y64, t64, p64, w, o)[0]
    assert res.dtype == torch.float32 and res.many_evaluations and res.t_offset == float(t64[-1])
    o2 = E.variance_objective()
    o2.sensor_size, o2.reference_exact = (H, W), False
    a = events_cmax.optimize_contrast(x64, y64, t64, p64, w, o2, optimizer="evk_bfgs", numeric_grads=False, blur_sigma=1.0,
                                      img_size=(H, W))
    o3 = E.variance_objective()
    o3.sensor_size, o3.reference_exact = (H, W), False
    a64 = events_cmax.optimize_contrast(ev64, None, None, None, w, o3, optimizer="evk_bfgs", numeric_grads=False, blur_sigma=1.0,
                                        img_size=(H, W))
    assert np.linalg.norm(a - a64) < 2e-2 and np.linalg.norm(a - np.array([40.0, -25.0])) < 1.0, (a, a64)


def test_uploads_convert_on_the_device_like_numpy_on_the_host():
    """_device.to_device: a host column of another width goes up as it is and is converted by a device copy -- the values numpy's
    astype would have produced, for every dtype the reference's arrays come in, strided views (xy[:, 0]), read-only and
    byte-swapped arrays (which keep the host conversion)."""
    import warnings
    from event_utils_amd import _device as D
    rng = np.random.default_rng(3)
    base = {"i64": rng.integers(-2 ** 40, 2 ** 40, 1001), "i32": rng.integers(-2 ** 31, 2 ** 31, 1001).astype(np.int32),
            "i16": rng.integers(-2 ** 15, 2 ** 15, 1001).astype(np.int16), "u8": rng.integers(0, 256, 1001).astype(np.uint8),
            "bool": rng.integers(0, 2, 1001).astype(np.bool_), "f64": rng.normal(0, 1e3, 1001), "f32": rng.normal(0, 1e3, 1001).astype(np.float32),
            "u16": rng.integers(0, 2 ** 16, 1001).astype(np.uint16), "f64_be": rng.normal(0, 1e3, 1001).astype(">f8")}
    targets = {torch.float32: np.float32, torch.float64: np.float64, torch.int32: np.int32, torch.int64: np.int64}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, a in base.items():
            for tdt, ndt in targets.items():
                if a.dtype.kind == "f" and np.dtype(ndt).kind == "i":
                    continue                      # (float -> int is not asked for anywhere: the integer entry points raise TypeError)
                want = a.astype(ndt)
                assert np.array_equal(D.to_device(a, tdt).cpu().numpy(), want), (name, tdt)
                ro = a.copy()
                ro.flags.writeable = False
                assert np.array_equal(D.to_device(ro, tdt).cpu().numpy(), want), (name, tdt, "read-only")
        xy = rng.integers(0, 640, (500, 2))
        assert np.array_equal(D.to_device(xy[:, 1], torch.int32).cpu().numpy(), xy[:, 1].astype(np.int32))
        assert D.to_device(np.zeros(0), torch.float32).shape == (0,)


def test_device_slices_take_the_one_pass_paths():
    """A device SLICE (xs[a:b]) starts wherever the slice does -- off a 16-byte boundary three times out of four.  Such columns
    used to fall to the direct kernels (2-8 global atomics per event); now the one-pass kernels read them where they lie
    (EVK_COLUMNS_UNALIGNED: their 16-byte loads are dword-aligned loads) when the 12 bytes behind the last event belong to the
    same storage, and aligned copies otherwise (from a few hundred thousand events on): same results as the aligned stream, the
    one-pass entry points called; small slices without slack and EVK_IMPL=direct keep the direct kernels; resident event sets
    built from slices are bucketed."""
    import numpy as np
    import torch
    import event_utils_amd as E
    from event_utils_amd import _lib, tiled
    from event_utils_amd.representations import image as I, voxel_grid as V
    from oracle import reference_np as R
    rng = np.random.default_rng(51)
    n, H, W, B = 1_200_003, 180, 240, 5
    x = rng.uniform(0, W - 1, n).astype(np.float32); y = rng.uniform(0, H - 1, n).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32); p = rng.choice([-1.0, 1.0], n).astype(np.float32)
    xd, yd, td, pd = (torch.from_numpy(a).cuda() for a in (x, y, t, p))
    calls = []
    orig = _lib.call
    _lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]

    def close(a, ref, tol=1e-5):
        a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
        assert np.max(np.abs(a - ref)) <= tol * max(np.max(np.abs(ref)), 1e-30)
    try:
        for off, end in ((1, n - 7), (2, n - 3), (3, n), (1, n - 1)):
            # (a slice that stops >= 3 events short of its parent's end is read where it lies, EVK_COLUMNS_UNALIGNED; one that
            # reaches the end has no readable slack behind its last event and is copied)
            sl = slice(off, end)
            cx, cy, ct, cp = (c[sl] for c in (xd, yd, td, pd))
            assert cx.data_ptr() % 16 != 0 and tiled.column_ok(ct) == (end <= n - 3)
            del calls[:]
            g = V.events_to_voxel_torch(cx.floor(), cy.floor(), ct, cp, B, sensor_size=(H, W))        # floor(): fresh, aligned
            g2 = V.events_to_voxel_torch(torch.floor(xd)[sl], torch.floor(yd)[sl], ct, cp, B, sensor_size=(H, W))
            assert calls.count("evk_voxel2_f32") == 2 and "evk_voxel_from_events_f32" not in calls
            ref = R.events_to_voxel_torch(np.floor(x[sl]), np.floor(y[sl]), t[sl], p[sl], B, sensor_size=(H, W), accum="f64")
            close(g.cpu().numpy(), ref); close(g2.cpu().numpy(), ref)
            del calls[:]
            img = I.events_to_image_torch(cx, cy, cp, sensor_size=(H, W), interpolation='bilinear')
            assert "evk_image2_bilinear_f32" in calls and "evk_image_bilinear_f32" not in calls
            close(img.cpu().numpy(), R.events_to_image_torch(x[sl], y[sl], p[sl], sensor_size=(H, W), interpolation='bilinear', accum="f64"))
            del calls[:]
            a, b = I.events_to_timestamp_image_torch(cx, cy, ct, cp, sensor_size=(H, W))
            assert "evk_timestamp_images2_f32" in calls and "evk_timestamp_images_f32" not in calls
            ra, rb = R.events_to_timestamp_image_torch(x[sl], y[sl], t[sl], p[sl], sensor_size=(H, W), accum="f64")
            close(a.cpu().numpy(), ra); close(b.cpu().numpy(), rb)
            pos, neg = V.events_to_neg_pos_voxel_torch(torch.floor(xd)[sl], torch.floor(yd)[sl], ct, cp, B, sensor_size=(H, W))
            close((pos - neg).cpu().numpy(), R.events_to_voxel_torch(np.floor(x[sl]), np.floor(y[sl]), t[sl], np.where(p[sl] > 0, 1, -1).astype(np.float32),
                                                                        B, sensor_size=(H, W), accum="f64"))
        # a small slice WITH slack is read in place too; without slack it keeps the direct kernel (the copy would cost more than
        # it saves)
        del calls[:]
        V.events_to_voxel_torch(torch.floor(xd)[1:50_001], torch.floor(yd)[1:50_001], td[1:50_001], pd[1:50_001], B, sensor_size=(H, W))
        assert "evk_voxel2_f32" in calls and "evk_voxel_from_events_f32" not in calls
        del calls[:]
        m = 50_000
        sx, sy, st, sp = (c[:m + 1].clone()[1:] for c in (torch.floor(xd), torch.floor(yd), td, pd))     # ends with its storage
        g = V.events_to_voxel_torch(sx, sy, st, sp, B, sensor_size=(H, W))
        assert "evk_voxel_from_events_f32" in calls and "evk_voxel2_f32" not in calls
        close(g.cpu().numpy(), R.events_to_voxel_torch(np.floor(x[1:m + 1]), np.floor(y[1:m + 1]), t[1:m + 1], p[1:m + 1], B, sensor_size=(H, W), accum="f64"))
        # the library refuses a misaligned column without the flag, as before
        from event_utils_amd import _device as D
        assert _lib.lib().evk_voxel2_f32(D.ptr(sx), D.ptr(sy), D.ptr(st), D.ptr(sp), m, H, W, 40, 15, 0.0, 1.0, B, 0, None, None, None, 0,
                                         None, None, 0, None) == -3        # EVK_EALIGN (include/evk.h)
        # resident events built from slices: bucketed (the objective's fused path), same value as from the aligned copy
        ev_s = E.DeviceEvents.from_arrays(xd[1:], yd[1:], td[1:], pd[1:])
        ev_a = E.DeviceEvents.from_arrays(xd[1:].clone(), yd[1:].clone(), td[1:].clone(), pd[1:].clone())
        assert all(c.data_ptr() % 16 == 0 for c in (ev_s.x, ev_s.y, ev_s.t, ev_s.p))
        o = E.variance_objective(); o.sensor_size = (H, W)
        w = E.linvel_warp()
        q = np.array([30.0, -20.0])
        fs = o.evaluate_function(q, ev_s, None, None, None, w, (H, W), 1.0)
        fa = o.evaluate_function(q, ev_a, None, None, None, w, (H, W), 1.0)
        assert fs == fa and tiled.iwe_plan(ev_s, float(t[-1]), 30.0, -20.0, float(W), float(H), H + 1, W + 1, 0) is not None
    finally:
        _lib.call = orig
    E.check_errors()
