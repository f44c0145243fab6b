"""GPU (-m gpu): the tile-bucketed LDS path (evk_tiled.hip) forced on, against the oracle and against the direct
(global-atomic) path, including the cases that stress its special handling: ragged sizes, empty tiles, events on tile
borders, negative-wrap coordinates, flows larger than the LDS halo (time slices + global-atomic spill), the Q1 canvas
quirk, adaptive lifespan slices."""
import numpy as np
import pytest
import torch

from oracle import reference_np as R
from event_utils_amd import _lib

pytestmark = pytest.mark.gpu
TOL = 1e-5


def f64(a):
    return np.asarray(a, dtype=np.float64)


def close(a, ref, tol=TOL):
    a, ref = f64(a), f64(ref)
    assert a.shape == ref.shape
    scale = max(np.max(np.abs(ref)), 1e-30)
    err = np.max(np.abs(a - ref))
    assert err <= tol * scale, "max err %.3e vs tol %.3e" % (err, tol * scale)


@pytest.fixture()
def E(monkeypatch):
    import event_utils_amd as E
    monkeypatch.setenv("EVK_IMPL", "tiled")
    return E


def _tiled():
    from event_utils_amd import tiled
    return tiled


def _events(seed, n, H, W, real=False, t_hi=0.1):
    rng = np.random.default_rng(seed)
    if real:
        x = rng.uniform(1, W - 1, n).astype(np.float32); y = rng.uniform(1, H - 1, n).astype(np.float32)
    else:
        x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
    t = np.sort(rng.uniform(0, t_hi, n)).astype(np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    return x, y, t, p


def test_bucketing_is_a_permutation(E):
    """Every event appears exactly once in its own tile's segment, segments tile the record array."""
    from event_utils_amd import tiled
    n, H, W = 300_001, 100, 150
    x, y, t, p = _events(3, n, H, W)
    xd, yd, td, pd = (torch.from_numpy(a).cuda() for a in (x, y, t, p))
    bk = tiled.bucket_events(xd, yd, td, pd, 0, H, W, 4, 3)
    rec = bk.records.cpu().numpy(); bs = bk.bucket_start.cpu().numpy().astype(np.int64)[:bk.ntiles + 1]
    assert bs[0] == 0 and bs[-1] == n and np.all(np.diff(bs) >= 0)
    tiles_x = -(-W // 16)
    key = (rec[:, 1].astype(np.int64) >> 3) * tiles_x + (rec[:, 0].astype(np.int64) >> 4)
    seg = np.searchsorted(bs, np.arange(n), side="right") - 1
    assert np.array_equal(key, seg)
    a = np.stack([x, y, t, p], 1)
    assert np.array_equal(rec[np.lexsort(rec.T[::-1])], a[np.lexsort(a.T[::-1])])
    # time order inside a tile is preserved up to intra-block interleaving: check it is at least mostly sorted
    inv = sum(int(np.sum(np.diff(rec[bs[k]:bs[k + 1], 2]) < 0)) for k in range(0, len(bs) - 1, 7))
    assert inv < 0.5 * n / 7


@pytest.mark.parametrize("n", [5, 64, 1001, 50_000, 400_003])
@pytest.mark.parametrize("shape", [(48, 64, 5), (50, 70, 3), (480, 640, 5), (7, 9, 2)])
def test_voxel_tiled_vs_oracle(E, n, shape):
    H, W, B = shape
    x, y, t, p = _events(n + H, n, H, W)
    x = x + np.float32(0.3)          # fractional coordinates: truncation
    x[x >= W] = W - 1
    ref = R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64")
    v = E.events_to_voxel_torch(*(torch.from_numpy(a).cuda() for a in (x, y, t, p)), B, sensor_size=(H, W))
    close(v.cpu().numpy(), ref)


@pytest.mark.parametrize("knobs", [{"share_cu": True}, {"xcd_order": False}, {"tile": (32, 16)}, {"tile": (31, 33)},
                                   {"EVK_VOXEL_DETERMINISTIC": "1"}, {"rec": 4},
                                   {"rec": 4, "EVK_VOXEL_DETERMINISTIC": "1", "share_cu": True}, {"rec": 8},
                                   {"count": False}, {"count": False, "rec": 4}, {"tiles_wg": 512},
                                   {"tiles_wg": 512, "count": False}])
def test_voxel_path_variants_agree_with_the_oracle(E, monkeypatch, knobs):
    """Every kernel shape of the voxel fast path at a small size (tiled.FORCE: the library picks them by size): the partition
    geometry a multi-rank job gets (8 K-event sub-chunks, room for a collective's workgroups), plain work-item order,
    power-of-two and odd tile shapes instead of the balanced choice, fixed-point (order-free) accumulation, 4-byte compact
    records (the default above 16 M events) and 8-byte records, with and without the unit-polarity counting mode, 512- and
    768-thread tile workgroups."""
    from event_utils_amd import tiled
    for k, v in knobs.items():
        if k.startswith("EVK_"):
            monkeypatch.setenv(k, v)
        else:
            monkeypatch.setitem(tiled.FORCE, k, v)
    for (n, H, W, B, seed) in ((700_001, 480, 640, 5, 3), (90_000, 100, 130, 3, 4)):
        x, y, t, p = _events(seed, n, H, W)
        x[: n // 3] = 7; y[: n // 3] = 9                          # a hot pixel: split tiles
        ref = R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64")
        v = E.events_to_voxel_torch(*(torch.from_numpy(a).cuda() for a in (x, y, t, p)), B, sensor_size=(H, W))
        close(v.cpu().numpy(), ref)
    E.check_errors()


def test_voxel_deterministic_mode_is_bit_reproducible_at_full_size(E, monkeypatch):
    """EVK_VOXEL_DETERMINISTIC=1 (a deterministic mode): int64 fixed-point cells, integer adds commute
    -- configs[1] (10 M events, 640x480, 5 bins) gives the SAME bits on every run and for a permuted event order of equal
    time stamps; a non-finite weight is refused, not silently dropped."""
    n, H, W, B = 10_000_000, 480, 640, 5
    x, y, t, p = _events(1, n, H, W)
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    plain = E.events_to_voxel_torch(*cols, B, sensor_size=(H, W)).cpu().numpy()
    monkeypatch.setenv("EVK_VOXEL_DETERMINISTIC", "1")
    runs = [E.events_to_voxel_torch(*cols, B, sensor_size=(H, W)).cpu().numpy() for _ in range(3)]
    assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2])
    close(runs[0], plain, 1e-6)
    assert abs(float(runs[0].astype(np.float64).sum()) - float(p.astype(np.float64).sum())) <= 1e-3 * n ** 0.5
    # the same multiset of events in another order (time stamps of each swapped pair made equal): same bits
    k = np.arange(0, n - 1, 2)
    t2 = t.copy(); t2[k + 1] = t2[k]
    base = E.events_to_voxel_torch(*(torch.from_numpy(a).cuda() for a in (x, y, t2, p)), B, sensor_size=(H, W)).cpu().numpy()
    x3, y3, p3 = x.copy(), y.copy(), p.copy()
    for a in (x3, y3, p3):
        a[k], a[k + 1] = a[k + 1].copy(), a[k].copy()
    perm = E.events_to_voxel_torch(*(torch.from_numpy(a).cuda() for a in (x3, y3, t2, p3)), B, sensor_size=(H, W)).cpu().numpy()
    assert np.array_equal(base, perm)
    p4 = p[:400_000].copy(); p4[7] = np.inf
    with pytest.raises(ValueError):
        E.events_to_voxel_torch(*(torch.from_numpy(a).cuda() for a in (x[:400_000], y[:400_000], t[:400_000], p4)), B,
                                sensor_size=(H, W))


def test_deterministic_mode_at_small_event_counts_views_and_constant_time_stamps(E, monkeypatch):
    """EVK_VOXEL_DETERMINISTIC=1 is a property of the CALL, not of its size: 5 000 events (far below the 'auto' threshold,
    where the float-atomic kernel would run) handed over as strided, unaligned views take the one-pass path and give the same
    bits for a permuted order; EVK_IMPL=direct contradicts the mode and raises; ts[-1] == ts[0] (Q9) gives the reference's
    NaN cells, not a refusal."""
    from oracle import reference_np as R
    n, H, W, B = 5000, 180, 240, 5
    rng = np.random.default_rng(77)
    x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
    hot = rng.random(n) < 0.6
    x[hot] = 17.0; y[hot] = 23.0                       # thousands of float32 weights on one pixel: the order would matter
    t = np.repeat(np.sort(rng.uniform(0, 1, n // 50)), 50).astype(np.float32)
    p = rng.normal(size=n).astype(np.float32)
    monkeypatch.setenv("EVK_VOXEL_DETERMINISTIC", "1")

    def views(cols):      # every second element of a buffer that starts 4 bytes off a 16-byte boundary
        out = []
        for a in cols:
            buf = torch.zeros(2 * n + 1, dtype=torch.float32, device="cuda")
            buf[1::2] = torch.from_numpy(a).cuda()
            out.append(buf[1::2])
        return out
    base = E.events_to_voxel_torch(*views((x, y, t, p)), B, sensor_size=(H, W)).cpu().numpy()
    close(base, R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64"))
    perm = np.concatenate([rng.permutation(50) + 50 * i for i in range(n // 50)])
    again = E.events_to_voxel_torch(*views(tuple(a[perm] for a in (x, y, t, p))), B, sensor_size=(H, W)).cpu().numpy()
    assert np.array_equal(base, again)
    monkeypatch.setenv("EVK_IMPL", "direct")
    with pytest.raises(ValueError):
        E.events_to_voxel_torch(*views((x, y, t, p)), B, sensor_size=(H, W))
    monkeypatch.delenv("EVK_IMPL")
    tc = np.full(n, 2.5, np.float32)
    got = E.events_to_voxel_torch(*(torch.from_numpy(a).cuda() for a in (x, y, tc, p)), B, sensor_size=(H, W)).cpu().numpy()
    with np.errstate(all="ignore"):
        ref = R.events_to_voxel_torch(x, y, tc, p, B, sensor_size=(H, W), accum="f64")
    assert np.isnan(ref).any() and np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.array_equal(got[~np.isnan(ref)], ref[~np.isnan(ref)])
    E.check_errors()


@pytest.mark.parametrize("mode", ["counting", "fixed"])
def test_integer_modes_are_order_free_also_where_hot_tiles_are_cut(E, monkeypatch, mode):
    """Half of the events in a 100x100 px blob: its tiles are cut into pieces whose partial tiles the last piece sums.  The
    integer modes -- the unit-polarity counting mode (default) and EVK_VOXEL_DETERMINISTIC with arbitrary weights -- hand
    the pieces' EXACT int64 cells over and round once, so a permuted stream (which moves events between the pieces) gives the
    same bits (round 3 staged float32 partial tiles: only run-to-run reproducible)."""
    n, H, W, B = 4_000_000, 480, 640, 5
    x, y, t, p = _events(31, n, H, W)
    rng = np.random.default_rng(8)
    hot = rng.random(n) < 0.5
    x[hot] = (W // 3 + rng.integers(0, 100, int(hot.sum()))).astype(np.float32)
    y[hot] = (H // 3 + rng.integers(0, 100, int(hot.sum()))).astype(np.float32)
    if mode == "fixed":
        p = (p * rng.uniform(0.1, 3.0, n)).astype(np.float32)
        monkeypatch.setenv("EVK_VOXEL_DETERMINISTIC", "1")
    # blocks of 4096 consecutive events share ONE time stamp, so that permuting inside a block keeps the stream sorted
    blk = 4096
    t = np.repeat(t[::blk], blk)[:n].astype(np.float32)
    base = E.events_to_voxel_torch(*(torch.from_numpy(a).cuda() for a in (x, y, t, p)), B, sensor_size=(H, W)).cpu().numpy()
    from oracle import reference_np as R
    close(base, R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64"))
    perm = np.concatenate([rng.permutation(blk) + i * blk for i in range(n // blk)] + [np.arange(n // blk * blk, n)])
    assert perm.shape[0] == n
    again = E.events_to_voxel_torch(*(torch.from_numpy(np.ascontiguousarray(a[perm])).cuda() for a in (x, y, t, p)), B,
                                    sensor_size=(H, W)).cpu().numpy()
    assert np.array_equal(base, again)
    # a different event COUNT in front moves every cut: the blob's cells must not change either
    pad = 12_345
    xs, ys, ts, ps = (np.concatenate([a[:1].repeat(pad), a]) for a in (x, y, t, p))
    xs[:pad], ys[:pad] = 5.0, 5.0                     # the extra events sit on pixel (5, 5)
    more = E.events_to_voxel_torch(*(torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (xs, ys, ts, ps)), B,
                                   sensor_size=(H, W)).cpu().numpy()
    more[:, 5, 5] = base[:, 5, 5]
    assert np.array_equal(base, more)


@pytest.mark.parametrize("kind", ["wide", "zero_one", "unsorted", "mixed", "early", "nan"])
def test_compact_records_are_exact_for_any_input(E, monkeypatch, kind):
    """4-byte records (t_norm delta | polarity code | cell): whatever does not fit -- polarities other than +-1 / 0, time
    stamps that are unsorted, sparse or so close to ts[0] that float32 steps are tiny, NaN -- escapes to the exact side array;
    the grid equals the oracle's, and the 8-byte-record grid of the same events bit for bit wherever no NaN is involved."""
    n, H, W, B = 500_003, 260, 346, 5
    x, y, t, p = _events(21, n, H, W)
    rng = np.random.default_rng(5)
    if kind == "wide":
        p = (p * rng.uniform(0.1, 3.0, n)).astype(np.float32)
    elif kind == "zero_one":
        p = rng.integers(0, 2, n).astype(np.float32)
    elif kind == "unsorted":
        t = rng.permutation(t); t[0], t[-1] = 0.0, 0.1
    elif kind == "mixed":
        p[::7] = 0.25
        t[n // 2: n // 2 + 4000] = t[n // 2: n // 2 + 4000][::-1]
    elif kind == "early":
        t = np.sort(np.concatenate([rng.uniform(0, 1e-9, n // 3), rng.uniform(0, 0.1, n - n // 3)])).astype(np.float32)
        t[0] = 0.0
    elif kind == "nan":       # the reference multiplies EVERY bin's weight by p: NaN * 0 = inf * 0 = NaN reach all B bins
        p[5::1001] = np.nan
        p[6::1013] = np.inf
    ref = R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64")
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    monkeypatch.setitem(_tiled().FORCE, "rec", 4)
    v4 = E.events_to_voxel_torch(*cols, B, sensor_size=(H, W)).cpu().numpy()
    monkeypatch.setitem(_tiled().FORCE, "rec", 8)
    v8 = E.events_to_voxel_torch(*cols, B, sensor_size=(H, W)).cpu().numpy()
    with np.errstate(invalid="ignore"):
        for v in (v4, v8):
            assert np.array_equal(np.isnan(v), np.isnan(ref)) and np.array_equal(np.isposinf(v), np.isposinf(ref))
    ok = np.isfinite(ref)
    scale = max(np.abs(ref[ok]).max(), 1e-30)
    assert np.abs(v4[ok] - ref[ok]).max() <= TOL * scale and np.abs(v8[ok] - ref[ok]).max() <= TOL * scale
    assert np.abs(v4[ok].astype(np.float64) - v8[ok]).max() <= 1e-6 * scale    # same per-event values, float64 sums


@pytest.mark.parametrize("scene", ["band", "blob"])
def test_compact_records_with_three_entries_per_lane_on_structured_scenes(E, scene, monkeypatch):
    """4-byte records take three table entries per lane and batch (k_voxel_tiles2, E = 3).  13 M events make a tile's column
    longer than one wave's first 64 entries, and a structured scene then exercises what uniform events never do: a band at
    three times the mean density gives every entry 4-7 chunks -- a wave's three entries per lane overflow its chunk list and it
    falls back to one entry at a time -- and a blob gives long segments (streamed by the whole wave) and cut tiles."""
    n, H, W, B = 13_000_000, 480, 640, 5
    x, y, t, p = _events(41, n, H, W)
    rng = np.random.default_rng(8)
    hot = rng.random(n) < (0.35 if scene == "band" else 0.6)
    if scene == "band":
        # 35 % of the events in a fifth of the rows: 2.4 x the mean per tile -- below the cut at 2.5 x, so a tile keeps its whole
        # column of ~1590 entries, ~38 records = 5 chunks each: ~990 chunks per wave against a list of 448
        y[hot] = rng.integers(200, 296, hot.sum()).astype(np.float32)
    else:
        x[hot] = (W // 3 + rng.integers(0, 100, hot.sum())).astype(np.float32)
        y[hot] = (H // 3 + rng.integers(0, 100, hot.sum())).astype(np.float32)
    ref = R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64")
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    monkeypatch.setitem(_tiled().FORCE, "rec", 4)
    for _ in range(2):
        close(E.events_to_voxel_torch(*cols, B, sensor_size=(H, W)).cpu().numpy(), ref)
    monkeypatch.setenv("EVK_VOXEL_DETERMINISTIC", "1")
    a = E.events_to_voxel_torch(*cols, B, sensor_size=(H, W))
    b = E.events_to_voxel_torch(*cols, B, sensor_size=(H, W))
    assert torch.equal(a, b)
    close(a.cpu().numpy(), ref)


@pytest.mark.parametrize("count", [True, False])
@pytest.mark.parametrize("rec", [8, 4])
def test_unit_polarity_counting_mode_with_events_outside_the_time_range(E, monkeypatch, count, rec):
    """The tile kernel's counting mode (unit polarities: an integer count per bin + ONE float64 sum, grid[b] = S0[b] - G[b] +
    G[b - 1]) against the oracle, with everything that leaves its straight-line path: a time range narrower than the stream
    (events up to a bin width outside [ts[0], ts[-1]] still reach the edge bins, those further out nothing), events exactly on
    the range's ends and on bin boundaries, zero polarities, a hot pixel (cut tiles); accumulate and overwrite mode.  Then a
    polarity of 0.5 anywhere in the stream must switch the call to the float64 path (same oracle, same bar)."""
    from event_utils_amd.representations.voxel_grid import _voxel_f32_device
    monkeypatch.setitem(_tiled().FORCE, "count", count)
    monkeypatch.setitem(_tiled().FORCE, "rec", rec)
    n, H, W, B = 900_000, 480, 640, 5
    x, y, t, p = _events(12, n, H, W)
    p[::11] = 0.0
    x[: n // 4] = 300; y[: n // 4] = 200
    t_lo, t_hi = np.float32(0.03), np.float32(0.07)          # the stream spans [0, 0.1]: a bin is 0.01 wide
    t[1000:1100] = t_lo; t[2000:2100] = t_hi; t[3000:3100] = t_lo + (t_hi - t_lo) * np.float32(0.5)
    t = np.sort(t)
    for pp in (p, np.where(np.arange(n) == n // 2, np.float32(0.5), p).astype(np.float32)):
        ref = R.events_to_voxel_torch(x, y, t, pp, B, sensor_size=(H, W), accum="f64", t_range=(t_lo, t_hi))
        cols = [torch.from_numpy(a).cuda() for a in (x, y, t, pp)]
        for _ in range(2):
            v = _voxel_f32_device(*cols, B, (H, W), float(t_lo), float(t_hi), impl="tiled", check=False)
            close(v.cpu().numpy(), ref)
        base = torch.full((B, H, W), 1.5, device="cuda")
        _voxel_f32_device(*cols, B, (H, W), float(t_lo), float(t_hi), out=base, impl="tiled", check=False)
        close(base.cpu().numpy() - 1.5, ref, 1e-4)
        if count and pp is p:      # integer sums: the same bits on every run
            a = _voxel_f32_device(*cols, B, (H, W), float(t_lo), float(t_hi), impl="tiled", check=False)
            assert torch.equal(a, v)
    # dt == 0 (Q9): every t_norm is NaN (0 / 0) -- the reference's NaN weights reach every bin of every pixel that holds an event
    tz = np.full(n, 0.05, dtype=np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        ref = R.events_to_voxel_torch(x, y, tz, p, B, sensor_size=(H, W), accum="f64")
    v = _voxel_f32_device(*[torch.from_numpy(a).cuda() for a in (x, y, tz, p)], B, (H, W), 0.05, 0.05, impl="tiled", check=False)
    v = v.cpu().numpy()
    assert np.array_equal(np.isnan(v), np.isnan(ref)) and np.isnan(ref).any()
    assert np.all(v[~np.isnan(ref)] == 0.0)


def test_cut_tile_hand_over_is_stable_over_many_launches(E):
    """The pieces of a cut tile hand their partial tiles to the last-arriving piece inside one launch (agent-scope stores and
    loads, a relaxed ticket, no fence), across XCDs, with the L2s warm from the previous launches.  A blob that cuts ~25 tiles
    into ~200 pieces, 30 launches back to back on the same buffers: unit polarities accumulate integers, so every launch must
    give the SAME bits (a stale word anywhere would show), and the oracle's grid."""
    n, H, W, B = 6_000_000, 480, 640, 5
    x, y, t, p = _events(23, n, H, W)
    rng = np.random.default_rng(4)
    hot = rng.random(n) < 0.5
    x[hot] = (W // 3 + rng.integers(0, 100, hot.sum())).astype(np.float32)
    y[hot] = (H // 3 + rng.integers(0, 100, hot.sum())).astype(np.float32)
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    first = E.events_to_voxel_torch(*cols, B, sensor_size=(H, W))
    filler = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    for i in range(30):
        if i % 3 == 0:
            filler.random_(0, 255)      # other traffic through the caches between the launches
        assert torch.equal(E.events_to_voxel_torch(*cols, B, sensor_size=(H, W)), first), i
    close(first.cpu().numpy(), R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64"))


def test_neg_pos_grids_in_deterministic_mode(E, monkeypatch):
    """Split-polarity tile kernel with fixed-point cells: both grids bit-reproducible and equal to the float64 accumulation
    (unit weights: every partial sum is exactly representable either way)."""
    n, H, W, B = 600_000, 260, 346, 5
    x, y, t, p = _events(9, n, H, W)
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    a = E.events_to_neg_pos_voxel_torch(*cols, B, sensor_size=(H, W))
    monkeypatch.setenv("EVK_VOXEL_DETERMINISTIC", "1")
    b = E.events_to_neg_pos_voxel_torch(*cols, B, sensor_size=(H, W))
    c = E.events_to_neg_pos_voxel_torch(*cols, B, sensor_size=(H, W))
    for u, v, w in zip(a, b, c):
        assert torch.equal(v, w)
        close(v.cpu().numpy(), u.cpu().numpy(), 1e-6)


def test_voxel_tiling_is_balanced_over_the_cus(E):
    """The one-pass path tiles the sensor so that every CU gets the same number of tiles (the tile kernel runs one
    workgroup per tile, all resident): 512 tiles at 640x480, a multiple of 256 within 1 % at 1280x720."""
    from event_utils_amd import tiled
    for (H, W, planes) in ((480, 640, 5), (720, 1280, 5), (480, 640, 10), (260, 346, 5)):
        tw, th = tiled.voxel2_shape(H, W, planes)
        T = -(-W // tw) * -(-H // th)
        assert _lib.lib().evk_voxel2_num_tiles(H, W, tw, th) == T and (tw | 1) * th <= 1024
        per_cu = -(-T // 256)
        assert per_cu * 256 - T <= 0.05 * T, (H, W, tw, th, T)
        assert per_cu * tw * th <= 1.06 * H * W / 256 + 64, (H, W, tw, th, T)


def test_voxel_tiled_errors_and_wrap(E, monkeypatch):
    n, H, W = 2000, 40, 60
    x, y, t, p = _events(1, n, H, W)
    x[5] = -1.0; y[7] = -3.0                      # wrap like torch index_put_
    ref = R.events_to_voxel_torch(x, y, t, p, 4, sensor_size=(H, W), accum="f64")
    v = E.events_to_voxel_torch(*(torch.from_numpy(a).cuda() for a in (x, y, t, p)), 4, sensor_size=(H, W))
    close(v.cpu().numpy(), ref)
    x[11] = W + 2.0
    bad = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    monkeypatch.setenv("EVK_ERRORS", "strict")        # one synchronisation per call, the reference's CPU behaviour
    with pytest.raises(IndexError):
        E.events_to_voxel_torch(*bad, 4, sensor_size=(H, W))
    with pytest.raises(IndexError):                   # host tensors: always strict
        E.events_to_voxel_torch(*(c.cpu() for c in bad), 4, sensor_size=(H, W))
    monkeypatch.setenv("EVK_ERRORS", "deferred")      # device in, device out: reported like a CUDA device-side assert
    v = E.events_to_voxel_torch(*bad, 4, sensor_size=(H, W))     # enqueues, returns
    assert v.is_cuda
    with pytest.raises(IndexError):
        E.check_errors()                               # ... and surfaces here (or at the next call on the stream)
    E.check_errors()                                   # reported once
    E.events_to_voxel_torch(*bad, 4, sensor_size=(H, W))
    torch.cuda.synchronize()
    with pytest.raises(IndexError):                   # the next call on the stream reports the previous one
        E.events_to_voxel_torch(*(torch.from_numpy(a).cuda() for a in (np.abs(x) % W, y, t, p)), 4, sensor_size=(H, W))
    E.check_errors()
    # all events in one pixel (one hot tile, every other tile empty)
    x[:] = 17; y[:] = 23
    ref = R.events_to_voxel_torch(x, y, t, p, 4, sensor_size=(H, W), accum="f64")
    v = E.events_to_voxel_torch(*(torch.from_numpy(a).cuda() for a in (x, y, t, p)), 4, sensor_size=(H, W))
    close(v.cpu().numpy(), ref)


@pytest.mark.parametrize("shape", [(2200, 3900, 2), (3000, 4100, 9)])
def test_voxel_sensors_beyond_the_tile_limit(E, shape):
    """More than 8192 tiles at the preferred shape: the tiles are enlarged (first case) or, when the enlarged tile's
    accumulators no longer fit the LDS, the direct kernel takes over (second case) -- never an error."""
    H, W, B = shape
    n = 200_000
    x, y, t, p = _events(3, n, H, W)
    ref = R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64")
    v = E.events_to_voxel_torch(*(torch.from_numpy(a).cuda() for a in (x, y, t, p)), B, sensor_size=(H, W))
    close(v.cpu().numpy(), ref)


def test_iwe_sensor_beyond_the_tile_limit(E):
    H, W, n = 3000, 4100, 100_000
    x, y, t, p = _events(4, n, H, W, real=True)
    prm = np.array([40., -30.])
    ref = R.get_iwe(prm, f64(x), f64(y), f64(t), f64(p), R.linvel_warp(), (H, W), sensor_size=(H, W), accum="f64")[0]
    iwe = E.get_iwe(prm, x, y, t, p, E.linvel_warp(), (H, W), sensor_size=(H, W))[0]
    close(iwe, ref)


def test_voxel_tiled_equals_direct_at_full_size(E, monkeypatch):
    """configs[1] at full size: 10M events, 640x480x5; tiled vs direct (both HIP) and mass conservation."""
    H, W, B, n = 480, 640, 5, 10_000_000
    x, y, t, p = _events(1, n, H, W)
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    vt = E.events_to_voxel_torch(*cols, B, sensor_size=(H, W))
    monkeypatch.setenv("EVK_IMPL", "direct")
    vd = E.events_to_voxel_torch(*cols, B, sensor_size=(H, W))
    close(vt.cpu().numpy(), vd.cpu().numpy())
    assert abs(vt.double().sum().item() - float(p.astype(np.float64).sum())) <= 1e-3 * np.sqrt(n)


@pytest.mark.parametrize("prm", [(0., 0.), (30., -20.), (-280., 310.), (2500., -1800.), (-9000., 12000.)])
@pytest.mark.parametrize("shape", [(180, 240, 60_000), (480, 640, 300_000)])
def test_iwe_tiled_vs_oracle(E, prm, shape):
    H, W, n = shape
    x, y, t, p = _events(H + n, n, H, W, real=True)
    prm = np.array(prm)
    ri, rd = R.get_iwe(prm, f64(x), f64(y), f64(t), f64(p), R.linvel_warp(), (H, W), compute_gradient=True,
                       sensor_size=(H, W), accum="f64")
    iwe, diwe = E.get_iwe(prm, x, y, t, p, E.linvel_warp(), (H, W), compute_gradient=True, sensor_size=(H, W))
    close(iwe, ri); close(diwe, rd)
    iwe2, none = E.get_iwe(prm, x, y, t, p, E.linvel_warp(), (H, W), compute_gradient=False, use_polarity=False,
                           sensor_size=(H, W))
    ri2, _ = R.get_iwe(prm, f64(x), f64(y), f64(t), f64(p), R.linvel_warp(), (H, W), use_polarity=False,
                       sensor_size=(H, W), accum="f64")
    assert none is None
    close(iwe2, ri2)


def test_iwe_tiled_q1_canvas_and_unsorted_times(E):
    """img_size larger than the hard-wired (181, 241) canvas (Q1), and a time column that is NOT sorted (the window
    placement is only a hint; every event must still land)."""
    H, W, n = 200, 300, 50_000
    x, y, t, p = _events(9, n, 170, 230, real=True)
    rng = np.random.default_rng(1)
    t = rng.permutation(t)
    t[-1] = 0.1
    prm = np.array([120., -75.])
    ri, rd = R.get_iwe(prm, f64(x), f64(y), f64(t), f64(p), R.linvel_warp(), (H, W), compute_gradient=True, accum="f64")
    iwe, diwe = E.get_iwe(prm, x, y, t, p, E.linvel_warp(), (H, W), compute_gradient=True)
    assert iwe.shape == (181, 241)
    close(iwe, ri); close(diwe, rd)


def test_objective_tiled_matches_golden_and_lifespan(E, golden):
    g = golden("f8_objective")
    x, y, t, p = f64(g["xs"]), f64(g["ys"]), f64(g["ts"]), f64(g["ps"])
    from event_utils_amd.events import DeviceEvents
    ev = DeviceEvents.from_arrays(x, y, t, p)
    w, obj = E.linvel_warp(), E.variance_objective()
    obj.impl = "tiled"
    for i, prm in enumerate(g["params"]):
        for j, s in enumerate(g["sigmas"]):
            f = obj.evaluate_function(prm, ev, None, None, None, w, (180, 240), blur_sigma=s)
            gr = obj.evaluate_gradient(prm, ev, None, None, None, w, (180, 240), blur_sigma=s)
            assert abs(f - g["f"][i, j]) <= TOL * abs(g["f"][i, j])
            assert np.max(np.abs(f64(gr) - g["grad"][i, j])) <= TOL * np.max(np.abs(g["grad"][i, j])) + 1e-9
    assert len(ev._buckets) == 1            # bucketed once, reused by all 30 evaluations
    al = E.variance_objective(adaptive_lifespan=True, minimum_events=5000)
    al.impl = "tiled"
    al.iter_update(np.array([400., -250.]))
    f = al.evaluate_function(np.array([40., -25.]), ev, None, None, None, w, (180, 240), blur_sigma=1.0)
    assert abs(f - g["al_f"]) <= TOL * abs(g["al_f"])


@pytest.mark.parametrize("shape", [(48, 64, 5, 400_000), (480, 640, 5, 3_000_000)])
def test_voxel_hot_tiles_are_split_and_combined(E, shape):
    """Clustered events (as real event data are): most events in a few pixels -> their tiles exceed the per-item cap,
    are split over several workgroups and recombined by the last-arriving part.  Repeated calls check that the arrival
    counters reset themselves; accumulate mode (out pre-filled) and overwrite mode are both covered."""
    from event_utils_amd import tiled
    from event_utils_amd.representations.voxel_grid import _voxel_f32_device
    H, W, B, n = shape
    x, y, t, p = _events(77, n, H, W)
    rng = np.random.default_rng(5)
    hot = rng.random(n) < 0.7
    x[hot] = (W // 2 + rng.integers(0, 3, hot.sum())).astype(np.float32)
    y[hot] = (H // 3 + rng.integers(0, 2, hot.sum())).astype(np.float32)
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    bk = tiled.bucket_events(*cols, 0, H, W, 4, 4)       # (the three-pass bucketing's own plan: 16x16 tiles)
    idx = bk.bucket_start.cpu().numpy()
    part_start = idx[bk.ntiles + 1: 2 * bk.ntiles + 2]
    assert part_start[-1] > bk.ntiles, "the hot tile should have been split"
    ref = R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64")
    for _ in range(3):
        v = E.events_to_voxel_torch(*cols, B, sensor_size=(H, W))
        close(v.cpu().numpy(), ref)
    base = torch.full((B, H, W), 2.5, device="cuda")
    _voxel_f32_device(*cols, B, (H, W), float(t[0]), float(t[-1]), out=base)
    close(base.cpu().numpy() - 2.5, ref, 1e-4)
    assert np.all(bk.bucket_start.cpu().numpy()[2 * bk.ntiles + 2: 3 * bk.ntiles + 2] == 0)


def test_iwe_hot_tiles(E):
    H, W, n = 180, 240, 500_000
    x, y, t, p = _events(31, n, H, W, real=True)
    rng = np.random.default_rng(6)
    hot = rng.random(n) < 0.8
    x[hot] = (100 + 4 * rng.random(hot.sum())).astype(np.float32)
    y[hot] = (60 + 4 * rng.random(hot.sum())).astype(np.float32)
    for prm in (np.array([30., -20.]), np.array([-400., 250.])):
        ri, rd = R.get_iwe(prm, f64(x), f64(y), f64(t), f64(p), R.linvel_warp(), (H, W), compute_gradient=True,
                           sensor_size=(H, W), accum="f64")
        iwe, diwe = E.get_iwe(prm, x, y, t, p, E.linvel_warp(), (H, W), compute_gradient=True, sensor_size=(H, W))
        close(iwe, ri); close(diwe, rd)


@pytest.mark.parametrize("fixed", [True, False])
def test_iwe_accumulator_modes_on_hot_pixels(E, monkeypatch, fixed):
    """The two LDS accumulator modes of the tiled IWE kernel (64-bit fixed point, float64) against the oracle -- on uniform
    events and on a scene whose events all carry the same sign and pile up on a few pixels (sums of ~1e5 per pixel)."""
    monkeypatch.setitem(_tiled().FORCE, "iwe_fixed", fixed)
    H, W, n = 180, 240, 400_000
    x, y, t, p = _events(77, n, H, W, real=True)
    rng = np.random.default_rng(8)
    hot = rng.random(n) < 0.9
    x[hot] = (100.25 + 1.5 * rng.random(hot.sum())).astype(np.float32)
    y[hot] = (60.5 + 1.5 * rng.random(hot.sum())).astype(np.float32)
    p_same = np.abs(p) * np.float32(3.0)                  # every event adds with the same sign: sums of ~1e5 per pixel
    for pol, prm in ((p_same, np.array([0., 0.])), (p_same, np.array([12., -7.])), (p, np.array([30., -20.]))):
        ri, rd = R.get_iwe(prm, f64(x), f64(y), f64(t), f64(pol), R.linvel_warp(), (H, W), compute_gradient=True,
                           sensor_size=(H, W), accum="f64")
        iwe, diwe = E.get_iwe(prm, x, y, t, pol, E.linvel_warp(), (H, W), compute_gradient=True, sensor_size=(H, W))
        close(iwe, ri); close(diwe, rd)
    # three flows in one pass (the numeric-gradient path) on the same hot scene
    obj, robj = E.variance_objective(), R.variance_objective()
    obj.sensor_size = robj.sensor_size = (H, W)
    robj.accum = "f64"
    prm = np.array([12., -7.])
    g = obj.evaluate_numeric_gradient(prm, x, y, t, p_same, E.linvel_warp(), (H, W), 1.0)
    d = [f64(a) for a in (x, y, t, p_same)]
    f0 = float(robj.evaluate_function(prm, *d, R.linvel_warp(), (H, W), 1.0))
    fr = [float(robj.evaluate_function(prm + e, *d, R.linvel_warp(), (H, W), 1.0)) for e in (np.array([1., 0.]), np.array([0., 1.]))]
    gr = np.array([fr[0] - f0, fr[1] - f0])
    assert np.abs(g - gr).max() <= 2e-5 * abs(f0), (g, gr, f0)   # differences of two values each good to 1e-5


def test_batch3_numeric_gradient_matches_single_evaluations(E, golden):
    """f(v), f(v+e1), f(v+e2) from ONE pass over the events == three separate evaluations; the batched
    forward-difference gradient == the one assembled from separate evaluations."""
    from event_utils_amd.events import DeviceEvents
    g = golden("f8_objective")
    ev = DeviceEvents.from_arrays(*(f64(g[k]) for k in ("xs", "ys", "ts", "ps")))
    w, obj = E.linvel_warp(), E.variance_objective()
    obj.impl = "tiled"
    for prm in (np.array([0., 0.]), np.array([38.5, -24.0]), np.array([-120., 75.])):
        gb = obj.evaluate_numeric_gradient(prm, ev, None, None, None, w, (180, 240), 1.0)
        f0 = f64(obj.evaluate_function(prm, ev, None, None, None, w, (180, 240), 1.0))
        f1 = f64(obj.evaluate_function(prm + [1, 0], ev, None, None, None, w, (180, 240), 1.0))
        f2 = f64(obj.evaluate_function(prm + [0, 1], ev, None, None, None, w, (180, 240), 1.0))
        gs = np.array([f1 - f0, f2 - f0])
        assert np.max(np.abs(gb - gs)) <= 2e-6 * max(abs(f0), 1e-3), (gb, gs)
    # big canvas + many events: the batched path and the separate path agree at VGA too
    rng = np.random.default_rng(3)
    n, H, W = 400_000, 480, 640
    x, y, t, p = _events(5, n, H, W, real=True)
    ev = DeviceEvents.from_arrays(x, y, t, p)
    obj.sensor_size = (H, W)
    prm = np.array([55., -35.])
    gb = obj.evaluate_numeric_gradient(prm, ev, None, None, None, w, (H, W), 1.0)
    fs = [f64(obj.evaluate_function(q, ev, None, None, None, w, (H, W), 1.0)) for q in (prm, prm + [1, 0], prm + [0, 1])]
    assert np.max(np.abs(gb - np.array([fs[1] - fs[0], fs[2] - fs[0]]))) <= 2e-6 * abs(fs[0])


def test_optimize_numeric_uses_batched_gradient(E, golden):
    g8, g = golden("f8_objective"), golden("f9_optimize_trace")
    x, y, t, p = (f64(g8[k]) for k in ("xs", "ys", "ts", "ps"))
    obj = E.variance_objective()
    calls = {"g": 0}
    orig = obj.evaluate_numeric_gradient
    obj.evaluate_numeric_gradient = lambda *a, **k: (calls.__setitem__("g", calls["g"] + 1), orig(*a, **k))[1]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        argmax = E.optimize(x, y, t, p, E.linvel_warp(), obj, numeric_grads=True, img_size=(180, 240))
    assert calls["g"] > 0
    fa = f64(obj.evaluate_function(np.asarray(argmax, float), x, y, t, p, E.linvel_warp(), (180, 240), 1.0))
    fr = f64(obj.evaluate_function(g["numeric_argmax"], x, y, t, p, E.linvel_warp(), (180, 240), 1.0))
    assert fa <= fr + 0.02 * abs(fr)


@pytest.mark.parametrize("n,shape", [(400_003, (480, 640, 5)), (70_000, (50, 70, 3)), (2_000_000, (720, 1280, 5))])
def test_neg_pos_voxel_grids_in_one_pass(E, n, shape):
    """events_to_neg_pos_voxel_torch (voxel_grid.py:155-182): both grids from one bucketing + one tile-kernel pass
    (EVK_VOXEL_SPLIT_POLARITY) == the oracle's two voxelisations with where(ps > 0) / where(ps <= 0) weights.
    Polarities are general reals with zeros (-> negative grid) and a NaN (-> neither grid)."""
    from event_utils_amd.representations import voxel_grid as V
    H, W, B = shape
    x, y, t, _ = _events(n, n, H, W)
    rng = np.random.default_rng(n)
    p = rng.normal(size=n).astype(np.float32)
    p[::7] = 0.0
    p[5] = np.nan
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    vp, vn = V.events_to_neg_pos_voxel_torch(*cols, B, sensor_size=(H, W))
    assert vp.shape == vn.shape == (B, H, W) and vp.is_cuda
    close(vp.cpu().numpy(), R.events_to_voxel_torch(x, y, t, (p > 0).astype(np.float32), B, sensor_size=(H, W), accum="f64"))
    close(vn.cpu().numpy(), R.events_to_voxel_torch(x, y, t, (p <= 0).astype(np.float32), B, sensor_size=(H, W), accum="f64"))
    assert abs(vp.double().sum().item() + vn.double().sum().item() - (n - 1)) <= 1e-3 * n ** 0.5   # every event but the NaN, once


def test_neg_pos_voxel_edge_cases(E):
    from event_utils_amd.representations import voxel_grid as V
    H, W, B, n = 40, 60, 4, 5000
    x, y, t, p = _events(2, n, H, W)
    cols = lambda: [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    t_same = np.full(n, 0.25, np.float32)                         # dt == 0: NaN at every event pixel of BOTH grids (Q9)
    vp, vn = V.events_to_neg_pos_voxel_torch(*(torch.from_numpy(a).cuda() for a in (x, y, t_same, p)), B, sensor_size=(H, W))
    rp = R.events_to_voxel_torch(x, y, t_same, (p > 0).astype(np.float32), B, sensor_size=(H, W), accum="f64")
    assert np.array_equal(np.isnan(vp.cpu().numpy()), np.isnan(rp)) and np.array_equal(np.isnan(vn.cpu().numpy()), np.isnan(rp))
    x[11] = W + 2.0
    with pytest.raises(IndexError):      # device tensors in, device grids out: reported as events_to_voxel_torch does
        V.events_to_neg_pos_voxel_torch(*cols(), B, sensor_size=(H, W))
        E.check_errors()
    with pytest.raises(IndexError):      # host tensors: before the call returns
        V.events_to_neg_pos_voxel_torch(*(torch.from_numpy(a) for a in (x, y, t, p)), B, sensor_size=(H, W))


def test_objective_at_1080p_and_three_planes(E):
    """1080p canvases have more 32x32 blur tiles x planes than the reduction scratch has slots (34 * 61 * 3 = 6222 > 4096):
    the fused post-pass strides over the tiles instead of refusing the launch.  The three-flow pass of the numeric gradient
    must equal three separate evaluations, and the single evaluation the oracle."""
    H, W, n = 1080, 1920, 600_000
    x, y, t, p = _events(5, n, H, W, real=True)
    obj = E.variance_objective()
    obj.sensor_size = (H, W)
    w, prm = E.linvel_warp(), np.array([25.0, -40.0])
    f0, g = obj.evaluate_function_and_numeric_gradient(prm, x, y, t, p, w, (H, W), 1.0)
    fs = [float(obj.evaluate_function(prm + d, x, y, t, p, w, (H, W), 1.0)) for d in (np.zeros(2), np.array([1., 0.]), np.array([0., 1.]))]
    assert abs(float(f0) - fs[0]) <= 2e-6 * abs(fs[0])
    assert np.abs(np.asarray(g) - np.array([fs[1] - fs[0], fs[2] - fs[0]])).max() <= 2e-5 * abs(fs[0])
    robj = R.variance_objective(); robj.sensor_size = (H, W); robj.accum = "f64"
    fr = float(robj.evaluate_function(prm, *(f64(a) for a in (x, y, t, p)), R.linvel_warp(), (H, W), 1.0))
    assert abs(fs[0] - fr) <= 1e-5 * abs(fr)


def test_a_subclass_that_overrides_warp_is_called(E):
    """The fused linear-flow kernels replace linvel_warp.warp only for that very method: a plugin that subclasses
    linvel_warp and overrides warp() goes through the generic path with ITS warp."""
    H, W, n = 120, 160, 200_000
    x, y, t, p = _events(9, n, H, W, real=True)
    calls = []

    class doubled(E.linvel_warp):
        def warp(self, xs, ys, ts, ps, t0, params, compute_grad=False):
            calls.append(1)
            return super().warp(xs, ys, ts, ps, t0, 2.0 * np.asarray(params), compute_grad)

    class renamed(E.linvel_warp):      # no override: still the fused kernels
        pass
    prm = np.array([20.0, -10.0])
    d = [f64(a) for a in (x, y, t, p)]
    iwe, _ = E.get_iwe(prm, *d, doubled(), (H, W), sensor_size=(H, W))
    ref, _ = R.get_iwe(2.0 * prm, *d, R.linvel_warp(), (H, W), sensor_size=(H, W), accum="f64")
    assert calls and np.abs(f64(iwe) - ref).max() <= 1e-5 * np.abs(ref).max()
    iwe2, _ = E.get_iwe(prm, *d, renamed(), (H, W), sensor_size=(H, W))
    ref2, _ = R.get_iwe(prm, *d, R.linvel_warp(), (H, W), sensor_size=(H, W), accum="f64")
    assert np.abs(f64(iwe2) - ref2).max() <= 1e-5 * np.abs(ref2).max()


@pytest.mark.parametrize("hot", [False, True])
@pytest.mark.parametrize("sensor", [(180, 240), (720, 1280)])
def test_compact_records_are_bit_identical(E, sensor, hot, monkeypatch):
    """EVK_IWE_COMPACT: sensor events (integer pixels, +-1 polarity) bucketed into 8-byte records give the same IWE, dIWE
    and three-flow images as the 16-byte records, bit for bit (x, y are rebuilt from the tile origin).  With a hot spot the
    plan splits tiles into parts (odd record ranges) whose tight windows a few events leave -- those take float atomics in
    either mode, so two runs agree to rounding only."""
    H, W = sensor
    n = 700_001
    rng = np.random.default_rng(31)
    x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
    if hot:
        x[: n // 3] = 7 + x[: n // 3] % 5
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32)
    p = rng.choice(np.array([-1.0, 1.0, 0.5, 0.0, -3.0], dtype=np.float32), n)
    out = {}
    for mode in ("full", "compact"):
        monkeypatch.setitem(_tiled().FORCE, "iwe_records", mode)
        ev = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
        obj = E.variance_objective(); obj.sensor_size = (H, W); obj.impl = "tiled"
        prm = np.array([55.0, -35.0])
        iwe, d = E.get_iwe(prm, ev, None, None, None, E.linvel_warp(), (H, W), sensor_size=(H, W), compute_gradient=True)
        fg = obj.evaluate_function_and_numeric_gradient(prm, ev, None, None, None, E.linvel_warp(), (H, W), 1.0)
        flags = [b.iwe_flag for b in ev._buckets.values()]
        assert flags and all(f == (_lib.EVK_IWE_COMPACT if mode == "compact" else 0) for f in flags)
        out[mode] = (np.asarray(iwe), np.asarray(d), float(fg[0]), np.asarray(fg[1]))
    a, b = out["full"], out["compact"]
    if hot:
        for k in (0, 1):
            assert np.abs(f64(a[k]) - f64(b[k])).max() <= 2e-6 * np.abs(f64(a[k])).max()
        assert abs(a[2] - b[2]) <= 2e-6 * abs(a[2])
        assert np.abs(a[3] - b[3]).max() <= 4e-6 * abs(a[2])   # forward differences of f
    else:
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert a[2] == b[2] and np.array_equal(a[3], b[3])
    ref, _ = R.get_iwe(np.array([55.0, -35.0]), *(f64(v) for v in (x, y, t, p)), R.linvel_warp(), (H, W), sensor_size=(H, W),
                       accum="f64")
    assert np.abs(f64(b[0]) - ref).max() <= 1e-5 * np.abs(ref).max()


@pytest.mark.parametrize("spoil", ["subpixel", "outside", "polarity"])
def test_events_that_do_not_compact_keep_their_records(E, spoil, monkeypatch):
    """One event with a sub-pixel coordinate, a coordinate outside the domain or a polarity with low mantissa bits: the
    bucketing keeps the 16-byte records (the verdict of evk_compact_records_f32), results as before."""
    monkeypatch.setitem(_tiled().FORCE, "iwe_records", "compact")
    H, W, n = 180, 240, 300_000
    rng = np.random.default_rng(33)
    x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    if spoil == "subpixel":
        y[12345] += 0.25
    elif spoil == "outside":
        x[777] = W + 3.0
    else:
        p[n - 1] = np.float32(0.1)
    ev = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
    iwe, _ = E.get_iwe(np.array([40.0, 10.0]), ev, None, None, None, E.linvel_warp(), (H, W), sensor_size=(H, W))
    assert [b.iwe_flag for b in ev._buckets.values()] == [0]
    ref, _ = R.get_iwe(np.array([40.0, 10.0]), *(f64(v) for v in (x, y, t, p)), R.linvel_warp(), (H, W), sensor_size=(H, W),
                       accum="f64")
    assert np.abs(f64(np.asarray(iwe)) - ref).max() <= 1e-5 * np.abs(ref).max()


def test_structured_scenes_are_flagged_and_balanced(E, monkeypatch):
    """The bucketing plan: a moving-edge scene (tiles hold 0.5 .. 2 x the mean) is flagged `structured`, its full tiles
    are split so that no work item exceeds ~the mean, and the item count stays within 2 per tile; uniform events are
    neither flagged nor split."""
    import os
    import bench
    from event_utils_amd import tiled
    H, W, n = 480, 640, 3_000_000
    for scene in ("edges", "uniform"):
        if scene == "edges":
            x, y, t, p = bench.structured_scene(5, n, H, W)
        else:
            x, y, t, p = _events(6, n, H, W, real=True)
        cols = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (x, y, t, p)]
        bk = tiled.bucket_events(*cols, 1, H + 1, W + 1, 4, 4)
        T = bk.ntiles
        idx = bk.bucket_start.cpu().numpy().astype(np.int64)
        counts = np.diff(idx[: T + 1])
        parts = np.diff(idx[T + 1: 2 * T + 2])
        assert counts.sum() == n and parts.min() >= 1
        if scene == "edges":
            assert bk.structured and parts.sum() <= 2 * T and parts.max() >= 2
            assert (counts / parts).max() <= 1.05 * max(counts.mean(), 4096)
        else:
            assert not bk.structured and parts.max() == 1


def test_voxel_on_bucketed_records_entry_point(E):
    """evk_voxel_tiled_f32 (include/evk.h): the voxel grid from tile-bucketed 16-byte records (evk_bucket_events_f32) -- the C
    entry point for callers that bucket a stream once and voxelise it repeatedly.  The Python dispatch takes the one-pass
    path (evk_voxel2_f32); this entry is checked here through ctypes, uniform and hot-pixel scene."""
    from event_utils_amd import tiled, _lib, _device as D
    L = _lib.lib()
    for (n, H, W, B, tw, th) in ((600_001, 480, 640, 5, 5, 4), (90_000, 100, 130, 3, 4, 4)):
        x, y, t, p = _events(51, n, H, W)
        x[: n // 3] = 7; y[: n // 3] = 9
        cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
        bk = tiled.bucket_events(*cols, 0, H, W, tw, th)
        nbytes = int(L.evk_voxel_tiled_staging_bytes(bk.ntiles, n, B, tw, th))
        staging = torch.empty(max(nbytes, 1), dtype=torch.uint8, device="cuda")
        out = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
        _lib.call("evk_voxel_tiled_f32", D.ptr(bk.records), D.ptr(bk.bucket_start), n, H, W, tw, th, float(t[0]), float(t[-1]), B,
                  _lib.EVK_VOXEL_OVERWRITE, D.ptr(out), D.ptr(staging), nbytes, D.stream())
        close(out.cpu().numpy(), R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64"))


@pytest.mark.timeout(300)
def test_live_voxel_call_is_bit_identical_to_the_two_launches(E, monkeypatch):
    """EVK_VOXEL2_LIVE (evk_voxel_live.h; opt-in: measured slower than the two launches it overlaps): the
    consumer kernel on the library's second stream accumulates the tiles while the partition sorts, the tile kernel proper
    only what it LEAVES.  Same integer sums as the counting mode: the grid equals the two-launch grid bit for bit -- on
    uniform events (every tile done by the consumers), on a blob (hot tiles left to the plan's pieces), with arbitrary
    polarities (everything left), with time stamps out of order (edge bins), over repeated calls on the same buffers (a
    stale record, table row or status word of the previous call would show), and against the oracle."""
    tiled = _tiled()
    H, W, B, n = 480, 640, 5, 5_000_000
    rng = np.random.default_rng(77)

    def grids(cols):
        out = []
        for live in (True, False, True):
            monkeypatch.setitem(tiled.FORCE, "live", live)
            out.append(E.events_to_voxel_torch(*cols, B, sensor_size=(H, W)))
        torch.cuda.synchronize()
        return out
    for k, kind in enumerate(("uniform", "uniform", "blob", "float polarity", "unsorted", "uniform")):
        x, y, t, p = _events(100 + k, n, H, W)
        if kind == "blob":
            hot = rng.random(n) < 0.5
            x[hot] = (200 + rng.integers(0, 60, hot.sum())).astype(np.float32)
            y[hot] = (100 + rng.integers(0, 60, hot.sum())).astype(np.float32)
        elif kind == "float polarity":
            p = (p * rng.uniform(0.1, 3.0, n)).astype(np.float32)
        elif kind == "unsorted":
            t = rng.permutation(t); t[0], t[-1] = 0.02, 0.08
        cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
        a, b, c = grids(cols)
        assert torch.equal(a, c), kind
        if kind == "float polarity":       # (the two-launch call stages exact polarities differently: same values, float64 sums)
            assert (a - b).abs().max().item() <= 1e-6 * b.abs().max().item(), kind
        else:
            assert torch.equal(a, b), kind
        if kind in ("uniform", "blob") and k != 1:
            close(a.cpu().numpy(), R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64"))
    # who did the work: on uniform events every tile was finished by a consumer (status word = epoch << 2 | DONE)
    idx = [v for key, v in tiled._zpersist.items() if key[0] == "voxel2_index"][0].cpu().numpy().astype(np.uint32)
    status = idx[16 + 2048 + 256: 16 + 2048 + 256 + 512]        # (V2_HDR + V2_MAX_TILES + 256 progress words)
    assert np.all((status & 3) == 2) and len(np.unique(status >> 2)) == 1, np.unique(status & 3, return_counts=True)


@pytest.mark.parametrize("shape", [(480, 640), (720, 1280)])
def test_both_counting_modes_give_the_exact_sums(E, monkeypatch, shape):
    """Unit polarities are accumulated as integers: an int32 count + an int64 fixed-point sum per bin where those planes fit
    two workgroups per CU (640x480), two int64 atomics per event in the float64 mode's planes where they do not (1280x720:
    EVK_VOXEL2_COUNT2, round 5).  With dyadic time stamps every contribution p (1 - f), p f is a multiple of 2^-8, both modes
    hold the EXACT sums, and one rounding to float32 gives the float64 oracle's grid bit for bit -- for a permuted event order
    too, and on a scene whose hot tiles are cut (the pieces hand over exact int64 cells)."""
    tiled = _tiled()
    H, W = shape
    B, n = 5, 1_500_000
    rng = np.random.default_rng(5)
    x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
    x[: n // 3] = 300 + (x[: n // 3] % 9); y[: n // 3] = 200 + (y[: n // 3] % 9)        # a hot spot: cut tiles
    t = (np.sort(rng.integers(0, 1025, n)) / 1024.0).astype(np.float32)
    t[0], t[-1] = 0.0, 1.0
    p = (rng.integers(0, 3, n) - 1).astype(np.float32)                                   # -1, 0, +1
    ref = R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64").astype(np.float32)
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    got = E.events_to_voxel_torch(*cols, B, sensor_size=(H, W)).cpu().numpy()
    assert np.array_equal(got, ref)
    perm = rng.permutation(n)
    pc = [c[torch.from_numpy(perm).cuda()] for c in cols]
    pc[2][0], pc[2][-1] = 0.0, 1.0                                                       # ts[0] / ts[-1] define the normalisation
    x2, y2, t2, p2 = (c.cpu().numpy() for c in pc)
    ref2 = R.events_to_voxel_torch(x2, y2, t2, p2, B, sensor_size=(H, W), accum="f64").astype(np.float32)
    assert np.array_equal(E.events_to_voxel_torch(*pc, B, sensor_size=(H, W)).cpu().numpy(), ref2)
    # the float64 atomics of the same call (both counting modes off) agree to the parity bar, not to the bit
    monkeypatch.setitem(tiled.FORCE, "count", False)
    close(E.events_to_voxel_torch(*cols, B, sensor_size=(H, W)).cpu().numpy(), ref)
