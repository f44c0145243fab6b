"""Kernel-variant dispatch for the two streaming hot paths (voxel grid, fused linear-flow IWE).
impl: 'direct' = one global atomic per contribution (evk_scatter.hip);
      'tiled'  = tile-bucketed events + per-tile LDS accumulate + one flush (evk_tiled.hip).
Default comes from EVK_IMPL (env) or 'auto'."""
import os

from . import _device as D
from . import _lib


def default_impl():
    return _lib.getenv("EVK_IMPL", "auto")


# Hooks for tests and measurements (NOT configuration: the defaults are what the library measured best).  Several kernel shapes
# are chosen by size -- 4-byte voxel records above 16 M events, compact IWE records beyond the Infinity Cache, 768- or
# 512-thread tile workgroups, the unit-polarity counting mode -- and the tests must be able to run each of them at small sizes:
#   rec          None | 4 | 8       record size of the one-pass voxel path (EVK_VOXEL2_REC4 / _REC8)
#   count        True | False       unit-polarity counting mode of its tile kernel (EVK_VOXEL2_NO_COUNT)
#   count2       True | False       ... in the float64 mode's planes where the counting mode's do not fit (EVK_VOXEL2_NO_COUNT2)
#   tiles_wg     0 | 512            512-thread tile workgroups everywhere (EVK_VOXEL2_WG512)
#   xcd_order    True | False       XCD-aware work-item order of the tile kernels (EVK_VOXEL2_NO_XCD_ORDER)
#   share_cu     None | True | False   leave LDS for a collective's workgroups (None: yes in a multi-rank job)
#   iwe_records  "auto" | "compact" | "full"   8-byte compact records of the bucketed IWE path
#   iwe_fixed    True | False       64-bit fixed-point LDS windows of the IWE kernels (False: float64)
#   image_fixed  True | False       fixed-point windows of the bilinear image tile kernel for unit weights
#   live         None | True | False   the voxel tiles accumulated while the partition sorts (EVK_VOXEL2_LIVE; None: only
#                                   with EVK_VOXEL_LIVE=1 -- measured SLOWER than the two launches)
#   legacy_scatter  False | True    the write-combining ring scatter of rounds 1-5 (and the separate compaction pass) instead of
#                                   the LDS-sorting scatter (EVK_STAGE_LEGACY_SCATTER)
FORCE = {"rec": None, "count": True, "tiles_wg": 0, "xcd_order": True, "share_cu": None, "iwe_records": "auto",
         "iwe_fixed": True, "image_fixed": True, "live": None, "count2": True, "legacy_scatter": False, "native_bfgs": True}


# 'auto' thresholds, measured (profiles/r04_direct_tiled_crossover.txt, profiles/r04_small_calls.txt; tools/crossover.py,
# tools/small_calls.py).  DEVICE time: the one-pass path's two launches cost 17-22 us whatever the event count, the direct voxel
# kernel (a memset + 2 global atomics per event at ~21 G/s; ts[0] / ts[-1] read by the kernel itself since round 4) 11 us up to
# 50 k events, 24 us at 200 k: crossover ~150 k events (event images: 320 k nearest, 80 k bilinear).  But a public call on
# device tensors is HOST-bound at these sizes, and the one-pass call is the cheaper one to issue -- its kernels report dropped
# events themselves, the direct route copies the counter and records an event: 23-31 us per call against 36-41 us (135 us for the
# two grids of events_to_neg_pos_voxel_torch) at 30 k, 100 k and 300 k events alike.  So 'auto' takes the one-pass path whenever
# the columns allow it (can_tile); the direct kernels keep unaligned views, float64 columns and EVK_IMPL=direct.  The IWE path
# re-uses its buckets over many evaluations (41 against 50-60 us per evaluation already at 2 k events) but its FIRST evaluation
# of a new event set pays ~110 us of bucketing: 150 k events is where a handful of evaluations break even
TILED_MIN_EVENTS = 1
TILED_MIN_EVENTS_NATIVE = 1
TILED_MIN_EVENTS_NEG_POS = 1
TILED_MIN_EVENTS_IWE = 150_000
# ... and an event set an OPTIMISER works on (DeviceEvents.many_evaluations, set by optimize_contrast / grid_search /
# recursive_search), or any DeviceEvents that is evaluated a second time, is evaluated tens of times: the buckets pay at any event count (round 6: optimize_contrast with evk_bfgs on
# 100 k events of the moving-edge scene at 240x180, 21 passes: 4.1 ms through the direct kernels -- 137-170 us per value + gradient
# pass, their global atomics collide on the edges -- against TILED_IWE_REUSED_MS below)
TILED_MIN_EVENTS_IWE_REUSED = 1
# EVK_VOXEL2_LIVE (round 5) is NOT a default: at 10 M events the live call takes 0.085 ms against 0.072 ms for the two launches
# (profiles/r05_live_ab.txt).  EVK_VOXEL_LIVE=1 requests it from this many events (fewer than ~3 sub-chunks per partition
# workgroup leave the consumer kernel nothing to overlap; the library itself refuses calls it cannot run live)
LIVE_MIN_EVENTS = 6_000_000
_WIN_MAX = {1: 64, 3: 48}       # LDS window edge cap (f64 cells): 64x64x8 B = 32 KB; 3 planes x 48x48x8 B = 54 KB
_persist = {}


class _BoundedCache(dict):
    """Size answers of the library keyed by (geometry, event count).  A stream of windows with ever-changing event counts
    (voxel_grids_fixed_t, a live sensor) would add one entry per distinct count for ever: past 4096 entries the cache starts
    over (an entry costs two cheap library calls to rebuild)."""

    def __setitem__(self, k, v):
        if len(self) >= 4096:
            self.clear()
        dict.__setitem__(self, k, v)


_staging_bytes = _BoundedCache()


def _buf(key, nbytes, device):
    """Grow-only persistent device scratch (avoids re-allocating tens of MB per objective evaluation)."""
    import torch
    k = (key, device.index, D.stream_id(device))   # per stream: calls are stream-ordered
    b = _persist.get(k)
    if b is None or b.numel() < nbytes:
        b = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)
        _persist[k] = b
    return b


_zpersist = {}


def _zbuf(key, nwords, device):
    """Grow-only persistent uint32 scratch that is ZERO when handed out for the first time (the one-pass voxel path keeps
    self-resetting counters in it: evk.h, evk_voxel2_f32)."""
    import torch
    k = (key, device.index, D.stream_id(device))
    b = _zpersist.get(k)
    if b is None or b.numel() < nwords:
        b = torch.zeros(max(int(nwords), 1), dtype=torch.int32, device=device)
        _zpersist[k] = b
    return b


def release_scratch(device=None, stream=None):
    """Drop the library's persistent per-stream scratch -- the grow-only record / table / staging buffers of the one-pass and
    bucketed paths (80 MB after one 10 M-event call, tens of GB after a 2 G-event one), the zeroed indices, the spill pairs of
    the fused objective evaluation and the small reduction slots -- for `device` (default: every device) and `stream` (a
    torch.cuda.Stream or its id; default: every stream).  The memory returns to torch's caching allocator
    (torch.cuda.empty_cache() hands it back to the driver); the next call on the stream allocates what it needs again, with a
    freshly zeroed index.  Safe at any time on the owning stream: the buffers are torch tensors, freed in stream order.
    Returns the number of bytes released."""
    dev_index = None if device is None else __import__("torch").device(device).index
    if stream is not None and not isinstance(stream, int):
        stream = stream.cuda_stream
    freed = 0

    def match(di, sid):
        return (dev_index is None or di == dev_index) and (stream is None or sid == stream)

    # (the marshalled calls cached on live DeviceEvents point into these buffers: dropped with them -- the bucketed records of a
    # DeviceEvents are its own and stay)
    from . import events as _events
    for ev in list(_events._LIVE):
        for k in ("_cmax_calls", "_cmax_last_single", "_cmax_last_b3"):
            ev.__dict__.pop(k, None)
    for store, pos in ((_persist, (1, 2)), (_zpersist, (1, 2)), (_spill, (0, 1)), (D._scratch, (0, 1))):
        for k in [k for k in store if match(k[pos[0]], k[pos[1]])]:
            v = store.pop(k)
            for t in (v if isinstance(v, (list, tuple)) else (v,)):
                if hasattr(t, "numel"):
                    freed += t.numel() * t.element_size()
    return freed


def _rezero_on_failure(index, call):
    """The one-pass paths keep self-resetting counters (tickets, per-tile totals, hand-over words) in `index`: a call that
    fails between its launches could leave them mid-count.  Any failure therefore zeroes the index before it propagates -- the
    next call starts from the state a freshly allocated index has (the IWE path's spill pair has the same rule, _spill_call)."""
    try:
        call()
    except BaseException:
        try:
            index.zero_()
        except Exception:   # noqa: BLE001  (a dead context: the original error is the one to report)
            pass
        raise


class Buckets:
    """Events partitioned by tile: `records` (n_kept, 4) float32 (x, y, t, p), `bucket_start` (ntiles+1) offsets.
    With compact records -- written by the bucketing itself when every event allows it (EVK_STAGE_COMPACT), or by compact() --
    `records` is the 8-byte compact record buffer (include/evk.h, EVK_IWE_COMPACT) and `iwe_flag` carries that flag for the IWE
    entry points.  `stats`: the bucketing also delivered the compact verdict and max |p| (EVK_STAGE_STATS)."""

    def __init__(self, records, bucket_start, key_mode, dom_h, dom_w, tw_log2, th_log2, ntiles, n, stats=False, compact=False):
        self.records, self.bucket_start, self.n = records, bucket_start, n   # bucket_start = the whole bucket index
        self.key_mode, self.dom_h, self.dom_w = key_mode, dom_h, dom_w
        self.tw_log2, self.th_log2, self.ntiles = tw_log2, th_log2, ntiles
        self.iwe_flag = 0
        self._structured = None
        self._stats, self._try_compact = stats, compact
        self._p_absmax = None

    def _tail(self):
        """The last three words of the index in ONE 12-byte copy: scene | compact verdict | bit pattern of max |p|."""
        import numpy as np
        w = self.bucket_start[-3:].cpu().numpy().view(np.uint32)
        self._structured = bool(w[0])
        if self._stats:
            # (0xFFFFFFFF: the scatter that ran does not deliver it -- the caller reduces the column itself)
            self._p_absmax = float(w[2:3].view(np.float32)[0]) if w[2] != 0xFFFFFFFF else None
            self._stats_pmax_known = w[2] != 0xFFFFFFFF
            if self._try_compact and w[1] == 0 and self.n:
                # the scatter wrote 8-byte compact records into the first half of the buffer
                self.records = self.records.view(-1).view(__import__("torch").int64)[: self.n + (self.n & 1)]
                self.iwe_flag = _lib.EVK_IWE_COMPACT
            self._try_compact = False

    @property
    def structured(self):
        """The bucketing's verdict on the scene: True when the fullest tile holds more than 1.25 x the mean tile population.
        Read once (with the other tail words, one 12-byte copy) and kept."""
        if self._structured is None:
            self._tail()
        return self._structured

    @property
    def p_absmax(self):
        """max |p| of the bucketed events when the bucketing computed it (EVK_STAGE_STATS), else None."""
        if self._stats and self._p_absmax is None and getattr(self, "_stats_pmax_known", True):
            self._tail()
        return self._p_absmax

    def settle(self):
        """Read what the bucketing decided on the device (record format, scene, max |p|): one small copy, one synchronisation."""
        if self._structured is None:
            self._tail()
        return self

    def compact(self):
        """Rewrite 16-byte records as 8-byte compact records when that is exact (integer pixel coordinates inside the
        domain, polarities without low mantissa bits: sensor events) and drop the 16-byte ones: every later evaluation
        streams half the bytes.  Since round 6 the bucketing does this itself (bucket_events(..., compact=True): the histogram
        pass delivers the verdict, the scatter writes compact records directly); this separate pass over the records remains for
        buckets built without it.  FORCE["iwe_records"]: "auto" (default) compacts when the 16-byte records do not fit the
        256 MB Infinity Cache (> 16 M events) -- measured on MI355X (tools/iwe_kernel_time.py): 50 M events / 720p, function
        evaluation kernel 0.225 -> 0.211 ms; 10 M events / VGA (cache-resident, bound by arithmetic and LDS atomics, where the
        decode costs instructions) 0.039 -> 0.041 ms; "compact" always tries, "full" never does."""
        import torch
        if self._stats:
            return self.settle()
        if self.iwe_flag or not want_compact(self.key_mode, self.n, self.tw_log2, self.th_log2):
            return self
        dev = self.records.device
        out = torch.empty(int(_lib.lib().evk_compact_records_bytes(self.n)) // 8, dtype=torch.int64, device=dev)
        verdict = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.call("evk_compact_records_f32", D.ptr(self.records), self.n, self.dom_h, self.dom_w, self.tw_log2,
                  self.th_log2, D.ptr(out), D.ptr(verdict), D.stream())
        if int(verdict.item()) == 0:
            self.records, self.iwe_flag = out, _lib.EVK_IWE_COMPACT
        return self


def want_compact(key_mode, n, tw_log2, th_log2):
    """Whether a bucketing of n events should try compact records (the FORCE["iwe_records"] policy, see Buckets.compact)."""
    mode = FORCE["iwe_records"]
    if key_mode != 1 or n == 0 or (1 << (tw_log2 + th_log2)) > 1024 or mode == "full":
        return False
    return mode == "compact" or n * 16 > (256 << 20)


# A device SLICE (xs[a:b]: a window of a resident stream, voxel_grid.py:109-111, the data loaders' index ranges) starts wherever
# the slice does -- off a 16-byte boundary three times out of four.  Such columns used to fall to the direct kernels: 0.98 ms
# instead of 0.072 ms for a 10 M-event voxel grid.  The one-pass kernels' 16-byte loads are dword-aligned loads (gfx950 takes them
# at any dword boundary; evk_part.h, load_col16), so a misaligned column is read WHERE IT LIES (EVK_COLUMNS_UNALIGNED) -- provided
# the 12 bytes behind its last event belong to the same storage (a group of four events that is only partly inside the stream is
# loaded whole): true for every slice that stops short of its parent's end.  A column without that slack, or a strided view, is
# copied to an aligned buffer from REALIGN_ATOMICS / (global atomics per event of the direct kernel) events on; below, the direct
# kernel is the cheaper of the two.
REALIGN_ATOMICS = 800_000


def column_ok(c):
    """Can the one-pass kernels read this device column in place?  Contiguous, and 16-byte aligned or followed by three more
    elements' worth of readable storage."""
    if not c.is_contiguous():
        return False
    if c.data_ptr() % 16 == 0:
        return True
    es = c.element_size()
    return c.data_ptr() % es == 0 and c.untyped_storage().nbytes() - (c.storage_offset() + c.shape[0]) * es >= 3 * es


def unaligned_flag(cols):
    """EVK_COLUMNS_UNALIGNED when one of the columns (column_ok) does not start on a 16-byte boundary."""
    return _lib.EVK_COLUMNS_UNALIGNED if any(c is not None and c.data_ptr() % 16 for c in cols) else 0


def realign(cols, impl, atomics_per_event):
    """`cols` (device columns of one length, or None entries) as the one-pass paths can take them: unchanged when they can be read
    in place (column_ok), the call is small, or EVK_IMPL=direct; else aligned copies of the ones that cannot."""
    import torch
    live = [c for c in cols if c is not None]
    if impl == "direct" or not live or (impl != "tiled" and live[0].shape[0] * atomics_per_event < REALIGN_ATOMICS):
        return cols          # (EVK_IMPL=tiled asks for the one-pass path at any count)
    if not all(c.is_cuda and c.dim() == 1 for c in live) or all(column_ok(c) for c in live):
        return cols
    return tuple(c if (c is None or column_ok(c)) else c.clone(memory_format=torch.contiguous_format) for c in cols)


def can_tile(cols, impl, min_events=None):
    """Tiled path preconditions: float32, contiguous, 16-byte aligned columns, n < 2^32; under 'auto' also enough
    events to amortise the bucketing pre-pass."""
    import torch
    n = cols[0].shape[0]
    if impl not in ("tiled", "auto") or n == 0 or n > 4_000_000_000:
        return False
    if not all(c.dtype == torch.float32 and column_ok(c) for c in cols):
        return False
    return impl == "tiled" or n >= (TILED_MIN_EVENTS if min_events is None else min_events)


def share_cu():
    """Whether the partition kernel should leave LDS for another kernel's workgroups (EVK_STAGE_SHARE_CU): yes when a
    collective may overlap it, i.e. in a torch.distributed job of more than one rank; FORCE["share_cu"] overrides."""
    if FORCE["share_cu"] is not None:
        return bool(FORCE["share_cu"])
    import torch.distributed as dist
    return bool(dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)


def bucket_events(xd, yd, td, pd, key_mode, dom_h, dom_w, tw_log2, th_log2, oob=None, stages=7, into=None, native=None,
                  stats=False, compact=False):
    """evk_bucket_events_f32: counting sort of the SoA columns by output tile (one histogram + one scatter pass).
    native = events.NativeColumns: the same sort reading the on-disk dtypes (evk_bucket_events_native_f32; the four
    column arguments are then ignored).  stats: the histogram pass also delivers the compact verdict and max |p|
    (EVK_STAGE_STATS); compact (implies stats): the scatter writes 8-byte compact records when every event allows it
    (EVK_STAGE_COMPACT) -- Buckets.settle() / .structured / .p_absmax read the outcome."""
    import torch
    L = _lib.lib()
    ntiles = L.evk_bucket_num_tiles(dom_h, dom_w, tw_log2, th_log2)
    if ntiles <= 0:
        raise _lib.EvkError("unsupported tiling %s of domain %s" % ((tw_log2, th_log2), (dom_h, dom_w)))
    n = xd.shape[0] if native is None else native.n
    dev = xd.device if native is None else native.t.device
    if into is not None:
        records, bucket_start = into.records, into.bucket_start
    else:
        records = torch.empty((n, 4), dtype=torch.float32, device=dev)
        bucket_start = torch.empty(int(L.evk_bucket_index_len(ntiles, n)), dtype=torch.int32, device=dev)
    nbytes = int(L.evk_bucket_scratch_bytes(ntiles))
    scratch = _buf("bucket", nbytes, dev)
    stats = bool(stats or compact)
    if FORCE.get("legacy_scatter"):
        stages |= _lib.EVK_STAGE_LEGACY_SCATTER
        compact = False
    stages |= (_lib.EVK_STAGE_STATS if stats else 0) | (_lib.EVK_STAGE_COMPACT if compact else 0)
    tail = (key_mode, dom_h, dom_w, tw_log2, th_log2, D.ptr(records), D.ptr(bucket_start), D.ptr(scratch), nbytes,
            oob.ptr if oob is not None else None, stages | (8 if share_cu() else 0), D.stream())
    if native is None:
        _lib.call("evk_bucket_events_f32", D.ptr(xd), D.ptr(yd), D.ptr(td), D.ptr(pd), n, *tail)
    else:
        _lib.call("evk_bucket_events_native_f32", *native.head(), *tail)
    return Buckets(records, bucket_start, key_mode, dom_h, dom_w, tw_log2, th_log2, ntiles, n, stats=stats, compact=compact)


def voxel_deterministic():
    """EVK_VOXEL_DETERMINISTIC=1: the tile kernel of the one-pass path accumulates 64-bit fixed point (order-free integer
    adds) instead of float64 -- bit-identical grids from run to run and for any order of the events.  Such a call takes the
    one-pass path at any event count (voxel_f32) and costs one synchronisation (range check)."""
    return _lib.getenv("EVK_VOXEL_DETERMINISTIC", "0") == "1"


_shape_cache = {}


def voxel2_shape(H, W, planes):
    """Tile size (width, height in PIXELS) for the one-pass voxel / event-image path, or None when it does not apply.
    The tile kernel runs one workgroup per tile and a launch lasts as long as its busiest CU, so the tiling is chosen to
    minimise (tiles per CU, rounded up) x (pixels per tile): 640x480 -> 512 tiles of 40x15 (2 per CU, against 600 tiles
    of 32x16 = 3 on 88 CUs and 2 on the rest: -17 % measured), 1280x720 -> 1020 tiles of 38x24.  Which tilings the kernels
    take -- the 10-bit cell index, the LDS of the partition and of the tile kernel, the tile count -- is the LIBRARY's
    answer (evk_voxel2_fits); only the preferences are here: fewer tiles (longer record segments), all tiles resident at
    once (<= 3 workgroups per CU) and wide tiles (rows are contiguous in the grid)."""
    key = (H, W, planes, FORCE.get("tile"))
    if key in _shape_cache:
        return _shape_cache[key]
    L = _lib.lib()
    ncu = L.evk_num_cu()
    best = None
    if FORCE.get("tile"):            # (tests: an explicit tile size)
        a, b = FORCE["tile"]
        if L.evk_voxel2_fits(H, W, a, b, planes):
            best = (0.0, a, b)
    else:
        for tw in range(8, 129):
            for th in range(4, 129):
                if (tw | 1) * th > 1024:
                    break
                if not L.evk_voxel2_fits(H, W, tw, th, planes):
                    continue
                T = L.evk_voxel2_num_tiles(H, W, tw, th)
                lds = planes * 8 * (tw | 1) * th + 43 * 1024          # (accumulators + chunk lists: residency estimate only)
                resident = min(3, (160 * 1024) // lds)
                per_cu = -(-T // ncu)
                cost = per_cu * tw * th * (1.0 + T / 20000.0) * (1.0 + 0.08 * max(0, -(-per_cu // resident) - 1)) \
                    * (1.0 + 0.02 * (th > tw)) * (1.0 + 0.01 * max(0.0, tw / th - 4.0))
                if best is None or cost < best[0]:
                    best = (cost, tw, th)
    shape = (best[1], best[2]) if best is not None else None
    _shape_cache[key] = shape
    return shape


def _voxel2_env(dev, n, B, H, W, tw, th, split_polarity=False):
    """(index, scratch, scratch bytes, flags) of a call of the one-pass voxel path: the persistent per-stream buffers and the
    flags every launch of one call must share (geometry, record size, workgroup shapes)."""
    L = _lib.lib()
    ver = "voxel2"
    planes = 2 * B if split_polarity else B
    ntiles = L.evk_voxel2_num_tiles(H, W, tw, th)
    key = (ver, ntiles, n, planes, tw, th)
    sizes = _staging_bytes.get(key)
    if sizes is None:
        sizes = (int(getattr(L, "evk_%s_index_len" % ver)(ntiles, n)),
                 int(getattr(L, "evk_%s_scratch_bytes" % ver)(ntiles, n, planes, tw, th)))
        if sizes[0] <= 0:
            raise _lib.EvkError("evk_%s: unsupported geometry (%d tiles, %d events)" % (ver, ntiles, n))
        _staging_bytes[key] = sizes
    index = _zbuf(ver + "_index", sizes[0], dev)
    scratch = _buf(ver + "_scratch", sizes[1], dev)
    flags = _lib.EVK_VOXEL_SPLIT_POLARITY if split_polarity else 0
    if share_cu():
        flags |= 128         # EVK_VOXEL2_SHARE_CU
    if not FORCE["xcd_order"]:
        flags |= 64          # EVK_VOXEL2_NO_XCD_ORDER
    flags |= {None: 0, 4: _lib.EVK_VOXEL2_REC4, 8: _lib.EVK_VOXEL2_REC8}[FORCE["rec"]]
    if not FORCE["count"]:
        flags |= _lib.EVK_VOXEL2_NO_COUNT
    if not FORCE.get("count2", True):
        flags |= _lib.EVK_VOXEL2_NO_COUNT2
    if FORCE["tiles_wg"] == 512:
        flags |= _lib.EVK_VOXEL2_WG512
    live = FORCE["live"]
    if live is None:
        live = n >= LIVE_MIN_EVENTS and _lib.getenv("EVK_VOXEL_LIVE", "0") == "1"
    if live and not split_polarity and not (flags & 128):
        flags |= _lib.EVK_VOXEL2_LIVE
    if voxel_deterministic():
        flags |= _lib.EVK_VOXEL_DETERMINISTIC
    return index, scratch, sizes[1], flags


def voxel2_band_rows(H, W, B, nbands):
    """Pixel-row ranges [(y_lo, y_hi), ...] of the row bands voxel2_bands cuts a (B, H, W) grid into (whole tile rows), or
    None when the one-pass path has no tiling for this grid.  A function of the grid and the library's tiling ONLY: every
    rank of an event-sharded run gets the same bands, whatever its own events are."""
    shape = voxel2_shape(H, W, B)
    if shape is None:
        return None
    th = shape[1]
    tiles_y = -(-H // th)
    nbands = max(1, min(int(nbands), tiles_y))
    edges = [k * tiles_y // nbands for k in range(nbands + 1)]
    return [(edges[k] * th, min(edges[k + 1] * th, H)) for k in range(nbands) if edges[k + 1] > edges[k]]


def voxel2_bands(cols, n, t_first, t_last, B, H, W, nbands, oob=None):
    """The one-pass voxel path in ROW BANDS (evk_voxel2_band_f32), for event-sharded runs: ONE partition, then the tile kernel
    of every band of tile rows launched on the current stream, each writing a contiguous (B, rows, W) buffer of its own.
    A generator: yields (y_lo, y_hi, band) right after band k's launch, so that the caller can start band k's all-reduce
    (another stream) while band k + 1 accumulates.  None when the one-pass path has no tiling for this grid."""
    import torch
    shape = voxel2_shape(H, W, B)
    if shape is None:
        return None
    tw, th = shape
    dev = cols[0].device
    index, scratch, nbytes, flags = _voxel2_env(dev, n, B, H, W, tw, th)
    rows = voxel2_band_rows(H, W, B, nbands)
    report, seq = oob.report_args() if oob is not None else (None, 0)
    dummy = torch.empty(1, dtype=torch.float32, device=dev)      # (the partition does not touch the grid)
    _rezero_on_failure(index, lambda: _lib.call(
        "evk_voxel2_f32", *(D.ptr(c) for c in cols), n, H, W, tw, th, t_first, t_last, B,
        flags | _lib.EVK_VOXEL2_PARTITION_ONLY, D.ptr(dummy), D.ptr(index), D.ptr(scratch), nbytes,
        oob.ptr if oob is not None else None, report, seq, D.stream()))

    def gen():
        for y0, y1 in rows:
            r0, r1 = y0 // th, -(-y1 // th)
            band = torch.empty((B, y1 - y0, W), dtype=torch.float32, device=dev)
            _rezero_on_failure(index, lambda: _lib.call("evk_voxel2_band_f32", n, H, W, tw, th, B, flags, r0, r1, D.ptr(band),
                                                        D.ptr(index), D.ptr(scratch), nbytes, D.stream()))
            yield y0, y1, band
    return gen()


def voxel2(cols, native, n, t_first, t_last, B, H, W, tw, th, out, oob, fresh, split_polarity=False, stage=0, aligned=False):
    """evk_voxel2_f32 / evk_voxel2_native_f32: partition + tile kernel from ONE library call.  t_first None = ts[0] and
    ts[-1] are read on the device (no transfer before the launch)."""
    index, scratch, nbytes, flags = _voxel2_env(out.device, n, B, H, W, tw, th, split_polarity)
    sizes = (index.numel(), nbytes)
    ver = "voxel2"
    flags |= (_lib.EVK_VOXEL_OVERWRITE if fresh else 0) | stage
    det = voxel_deterministic()
    if t_first is None:
        flags |= _lib.EVK_VOXEL_T_FROM_EVENTS
        t_first = t_last = 0.0
    report, seq = oob.report_args() if (oob is not None and not (stage & _lib.EVK_VOXEL2_TILES_ONLY)) else (None, 0)
    tail = (H, W, tw, th, t_first, t_last, B, flags, D.ptr(out), D.ptr(index), D.ptr(scratch), sizes[1],
            oob.ptr if oob is not None else None, report, seq, D.stream())
    if native is None:
        # a device slice is read where it lies (can_tile checked the slack behind it); aligned=True: the caller has looked already
        if not aligned and any(c is not None and c.data_ptr() % 16 for c in cols):
            tail = tail[:7] + (flags | _lib.EVK_COLUMNS_UNALIGNED,) + tail[8:]
        _rezero_on_failure(index, lambda: _lib.call("evk_%s_f32" % ver, *(D.ptr(c) for c in cols), n, *tail))
    else:
        _rezero_on_failure(index, lambda: _lib.call("evk_%s_native_f32" % ver, *native.head(), *tail))
    if det and not (stage & _lib.EVK_VOXEL2_PARTITION_ONLY):
        bad = int(index[4].item())          # synchronises: the deterministic mode is a debugging / verification mode
        if bad:
            index[4] = 0
            raise ValueError("EVK_VOXEL_DETERMINISTIC: %d contributions were not finite or beyond 2^30 and cannot be "
                             "accumulated in fixed point" % bad)


# event images (evk_image2.hip): in device time the two launches of the one-pass path (~15 / ~18 us whatever the event count) beat
# one global atomic (nearest) or four (bilinear) per event at ~21 G/s from 320 k / 80 k events (profiles/r04_image_crossover.txt);
# the public call is host-bound below that and cheaper to issue on the one-pass path (see above): any count
TILED_MIN_EVENTS_IMAGE = 1
TILED_MIN_EVENTS_IMAGE_BILINEAR = 1
# the average-timestamp images (round 6): as the event images -- the one-pass call of device tensors has no read-back and no
# stream synchronisation (ts[0] / ts[-1] read by the kernel, the IndexError check on the partition kernel's report), the direct one
# has both: 0.054 against 0.090 ms per public call at 50 k events (profiles/r06_timestamp_images.txt): any count
TILED_MIN_EVENTS_TIMESTAMP = 1
# interpolate_to_image on caller-computed pixels / fractions (round 6): four global atomics per event at ~21 G/s against the two
# launches' ~18 us; the call synchronises either way (it raises before it returns)
TILED_MIN_EVENTS_SPLAT_INDEXED = 100_000
# the derivative splats (interpolate_to_derivative_img: 8 atomics per event; events_to_image_drv: 4 or 12): the one-pass tile kernel
# fetches every event's weights by index (dependent loads), so its crossover lies higher than the plain splat's
TILED_MIN_EVENTS_SPLAT_DRV = 100_000


def image2(kind, xd, yd, wd_, n, H, W, clipx, clipy, out, oob, fresh=False, stage=0):
    """evk_image2_nearest_i32 / _nearest_f32 / _bilinear_f32 (kind = 'i32' | 'f32' | 'bilinear'): the event image of the
    device columns accumulated into `out` (H, W) -- or, nearest kinds with fresh=True, written over it (no memset needed).
    Returns False when the one-pass path has no tiling for this image (the caller then uses the direct kernel)."""
    L = _lib.lib()
    shape = voxel2_shape(H, W, 1)
    if shape is None or (kind == "bilinear" and (H < 2 or W < 2)):
        return False
    tw, th = shape
    if (tw + 2) * (th + 1) > 2048:
        return False
    dev = out.device
    ntiles = L.evk_voxel2_num_tiles(H, W, tw, th)
    key = ("image2", ntiles, n, tw, th)
    sizes = _staging_bytes.get(key)
    if sizes is None:
        sizes = (int(L.evk_voxel2_index_len(ntiles, n)), int(L.evk_image2_scratch_bytes(ntiles, n, tw, th)))
        if sizes[0] <= 0:                 # (checked BEFORE it is cached: a later call with the same key must not skip this)
            return False
        _staging_bytes[key] = sizes
    index = _zbuf("image2_index", sizes[0], dev)          # (its own: the header words [0], [1] mean something else here)
    scratch = _buf("voxel2_scratch", sizes[1], dev)
    flags = stage | (_lib.EVK_VOXEL_OVERWRITE if (fresh and kind != "bilinear") else 0)
    if kind == "i32":
        if any(c is not None and c.data_ptr() % 16 for c in (xd, yd, wd_)):
            return False                  # (the integer entry point reads aligned columns only; its callers upload them)
    else:
        flags |= unaligned_flag((xd, yd, wd_))
    if not FORCE["image_fixed"]:
        flags |= _lib.EVK_IMAGE2_NO_FIXED
    if not FORCE["xcd_order"]:
        flags |= 64
    report, seq = oob.report_args() if (oob is not None and not (stage & _lib.EVK_VOXEL2_TILES_ONLY)) else (None, 0)
    tail = (tw, th, flags, D.ptr(out), D.ptr(index), D.ptr(scratch), scratch.numel(), oob.ptr if oob is not None else None,
            report, seq, D.stream())
    if kind == "i32":
        _rezero_on_failure(index, lambda: _lib.call("evk_image2_nearest_i32", D.ptr(xd), D.ptr(yd), D.ptr(wd_), n, H, W, *tail))
    else:
        _rezero_on_failure(index, lambda: _lib.call("evk_image2_%s_f32" % ("bilinear" if kind == "bilinear" else "nearest"),
                                                    D.ptr(xd), D.ptr(yd), D.ptr(wd_), n, H, W, clipx, clipy, *tail))
    return True


def _indexed_env(dev, n, H, W, cols, oob, stage):
    """Tiling, index, scratch, flags and report slot shared by the one-pass calls on indexed / float64 columns, or None."""
    L = _lib.lib()
    shape = voxel2_shape(H, W, 1)
    if shape is None or H < 2 or W < 2:
        return None
    tw, th = shape
    if (tw + 2) * (th + 1) > 2048:
        return None
    ntiles = L.evk_voxel2_num_tiles(H, W, tw, th)
    key = ("indexed2", ntiles, n, tw, th)
    sizes = _staging_bytes.get(key)
    if sizes is None:
        sizes = (int(L.evk_voxel2_index_len(ntiles, n)), int(L.evk_image2_indexed_scratch_bytes(ntiles, n, tw, th)))
        if sizes[0] <= 0:
            return None
        _staging_bytes[key] = sizes
    index = _zbuf("image2_index", sizes[0], dev)
    scratch = _buf("voxel2_scratch", sizes[1], dev)
    flags = stage | unaligned_flag(cols)
    if not FORCE["image_fixed"]:
        flags |= _lib.EVK_IMAGE2_NO_FIXED
    if not FORCE["xcd_order"]:
        flags |= 64
    report, seq = oob.report_args() if (oob is not None and not (stage & _lib.EVK_VOXEL2_TILES_ONLY)) else (None, 0)
    return tw, th, flags, index, scratch, report, seq


def splat_indexed2(pxs, pys, dxs, dys, ws, n, H, W, img, oob, stage=0):
    """evk_image2_splat_indexed_f32: interpolate_to_image (image.py:102-115) of caller-computed pixels / fractions ADDED to `img`
    (H, W) on the one-pass path.  Returns False when it has no tiling for this image (the caller then uses the direct kernel)."""
    env = _indexed_env(img.device, n, H, W, (pxs, pys, dxs, dys, ws), oob, stage)
    if env is None:
        return False
    tw, th, flags, index, scratch, report, seq = env
    _rezero_on_failure(index, lambda: _lib.call(
        "evk_image2_splat_indexed_f32", D.ptr(pxs), D.ptr(pys), D.ptr(dxs), D.ptr(dys), D.ptr(ws), n, H, W, tw, th, flags, D.ptr(img),
        D.ptr(index), D.ptr(scratch), scratch.numel(), oob.ptr if oob is not None else None, report, seq, D.stream()))
    return True


def splat_drv_indexed2(pxs, pys, dxs, dys, w1, w2, n, H, W, d_img, oob, stage=0):
    """evk_image2_splat_drv_indexed_f32: interpolate_to_derivative_img (image.py:117-136; two channels) ADDED to `d_img` (2, H, W).
    w1, w2: contiguous (2, n) float32.  Returns False when the one-pass path has no tiling for this image."""
    env = _indexed_env(d_img.device, n, H, W, (pxs, pys, dxs, dys), oob, stage)
    if env is None:
        return False
    tw, th, flags, index, scratch, report, seq = env
    _rezero_on_failure(index, lambda: _lib.call(
        "evk_image2_splat_drv_indexed_f32", D.ptr(pxs), D.ptr(pys), D.ptr(dxs), D.ptr(dys), D.ptr(w1), D.ptr(w2), n, H, W, tw, th,
        flags, D.ptr(d_img), D.ptr(index), D.ptr(scratch), scratch.numel(), oob.ptr if oob is not None else None, report, seq,
        D.stream()))
    return True


def image_drv2(xd, yd, pd, jx, jy, n, H, W, clipx, clipy, img, d_img, oob, stage=0):
    """evk_image2_drv_f64: events_to_image_drv (image.py:162-217) on float64 device columns, image (and, with jx / jy (2, n), its
    two derivative planes) ADDED to img / d_img.  Returns False when the one-pass path has no tiling for this image."""
    env = _indexed_env(img.device, n, H, W, (), oob, stage)
    if env is None:
        return False
    tw, th, flags, index, scratch, report, seq = env
    _rezero_on_failure(index, lambda: _lib.call(
        "evk_image2_drv_f64", D.ptr(xd), D.ptr(yd), D.ptr(pd), D.ptr(jx), D.ptr(jy), n, H, W, clipx, clipy, tw, th, flags, D.ptr(img),
        D.ptr(d_img), D.ptr(index), D.ptr(scratch), scratch.numel(), oob.ptr if oob is not None else None, report, seq, D.stream()))
    return True


def timestamp_images2(xd, yd, td, pd, n, H, W, clipx, clipy, mode, ta, tdiv, out4, oob, stage=0, from_events=False):
    """evk_timestamp_images2_f32: the four average-timestamp planes of the device columns ADDED to `out4` (4, H, W) on the one-pass
    partition + LDS windows (image.py:219-353).  Returns False when the one-pass path has no tiling for this image (the caller
    then uses the direct kernel, evk_timestamp_images_f32).  from_events (modes 0 / 1): ta / tdiv are formed on the device from
    td[0] / td[-1] (image.py:326-329) -- no read-back before the launches."""
    L = _lib.lib()
    shape = voxel2_shape(H, W, 1)
    if shape is None or H < 2 or W < 2:
        return False
    tw, th = shape
    if (tw + 2) * (th + 1) > 2048:
        return False
    dev = out4.device
    ntiles = L.evk_voxel2_num_tiles(H, W, tw, th)
    key = ("timestamp2", ntiles, n, tw, th)
    sizes = _staging_bytes.get(key)
    if sizes is None:
        sizes = (int(L.evk_voxel2_index_len(ntiles, n)), int(L.evk_timestamp_images2_scratch_bytes(ntiles, n, tw, th)))
        if sizes[0] <= 0:
            return False
        _staging_bytes[key] = sizes
    index = _zbuf("image2_index", sizes[0], dev)
    scratch = _buf("voxel2_scratch", sizes[1], dev)
    flags = stage | (_lib.EVK_VOXEL_T_FROM_EVENTS if from_events else 0) | unaligned_flag((xd, yd, td, pd))
    if not FORCE["image_fixed"]:
        flags |= _lib.EVK_IMAGE2_NO_FIXED
    if not FORCE["xcd_order"]:
        flags |= 64
    report, seq = oob.report_args() if (oob is not None and not (stage & _lib.EVK_VOXEL2_TILES_ONLY)) else (None, 0)
    _rezero_on_failure(index, lambda: _lib.call(
        "evk_timestamp_images2_f32", D.ptr(xd), D.ptr(yd), D.ptr(td), D.ptr(pd), n, H, W, clipx, clipy, mode, float(ta), float(tdiv),
        tw, th, flags, D.ptr(out4), D.ptr(index), D.ptr(scratch), scratch.numel(), oob.ptr if oob is not None else None,
        report, seq, D.stream()))
    return True


def can_tile_image(cols, impl, bilinear=False):
    """Preconditions of the one-pass image path: contiguous, 16-byte aligned 4-byte columns and, under 'auto', enough
    events to amortise its two launches."""
    n = cols[0].shape[0]
    if impl not in ("tiled", "auto") or n == 0 or n > 4_000_000_000:
        return False
    if not all(c is None or (c.element_size() == 4 and column_ok(c)) for c in cols):
        return False
    return impl == "tiled" or n >= (TILED_MIN_EVENTS_IMAGE_BILINEAR if bilinear else TILED_MIN_EVENTS_IMAGE)


def voxel_neg_pos_f32(xd, yd, td, pd, t_first, t_last, B, H, W, oob=None, impl=None, native=None):
    """events_to_neg_pos_voxel_torch core: (2, B, H, W) float32 = [positive events, non-positive events] from ONE
    partition and ONE tile-kernel pass, or None when the one-pass path does not apply (the caller then voxelises the
    two weight columns one after the other, as upstream).  native = events.NativeColumns: the on-disk dtypes, read as stored."""
    import torch
    impl = impl or default_impl()
    if native is not None:
        if not (impl != "direct" and native.aligned() and native.n and (impl == "tiled" or native.n >= TILED_MIN_EVENTS_NATIVE)):
            return None
    else:
        xd, yd, td, pd = realign((xd, yd, td, pd), impl, 4)
        if not can_tile((xd, yd, td, pd), impl, TILED_MIN_EVENTS_NEG_POS):
            return None
    shape2 = voxel2_shape(H, W, 2 * B)
    if shape2 is None:
        return None
    dev = xd.device if native is None else native.t.device
    out = torch.empty((2, B, H, W), dtype=torch.float32, device=dev)
    voxel2((xd, yd, td, pd), native, xd.shape[0] if native is None else native.n, t_first, t_last, B, H, W, *shape2, out, oob,
           True, split_polarity=True)
    return out


def voxel_f32(xd, yd, td, pd, t_first, t_last, B, H, W, out, oob=None, impl=None, fresh=False, native=None):
    """events_to_voxel_torch core on device columns; accumulates into `out` (B, H, W).  fresh=True: `out` is
    uninitialised memory and is fully (over)written -- the tiled path then needs no memset at all.
    native = events.NativeColumns: the tiled path buckets the on-disk dtypes directly (xd..pd may then be callables
    producing the widened float32 columns, only called when the direct kernel has to take over)."""
    impl = impl or default_impl()
    det = voxel_deterministic()
    if det:
        # the integer tile kernel belongs to the one-pass path: the call takes it at ANY event count (the direct kernel adds
        # float32 atomics in whatever order they arrive), after copying columns that path cannot read in place
        if impl == "direct":
            raise ValueError("EVK_VOXEL_DETERMINISTIC=1 needs the one-pass path; EVK_IMPL=direct selects the float-atomic kernels")
        impl = "tiled"
        if native is None:
            xd, yd, td, pd = (c if c.data_ptr() % 16 == 0 else c.clone() for c in (c.contiguous() for c in (xd, yd, td, pd)))
    if native is not None:
        tileable = impl != "direct" and native.aligned() and (impl == "tiled" or native.n >= TILED_MIN_EVENTS_NATIVE)
    else:
        # (the common case -- freshly allocated columns -- costs four pointer reads here and skips the slice handling)
        aligned16 = not ((xd.data_ptr() | yd.data_ptr() | td.data_ptr() | pd.data_ptr()) & 15)
        if not aligned16:
            xd, yd, td, pd = realign((xd, yd, td, pd), impl, 2)
        tileable = can_tile((xd, yd, td, pd), impl)
    if tileable:
        shape2 = voxel2_shape(H, W, B)
        if shape2 is not None:
            voxel2((xd, yd, td, pd), native, xd.shape[0] if native is None else native.n, t_first, t_last, B, H, W, *shape2,
                   out, oob, fresh, aligned=native is None and aligned16)
            return out
    if det and (xd.shape[0] if native is None else native.n):
        raise ValueError("EVK_VOXEL_DETERMINISTIC=1: the one-pass path cannot take this call (%d bins of %dx%d: no tiling "
                         "fits the LDS, or unaligned on-disk columns)" % (B, H, W))
    if t_first is None:
        if native is None:       # ts[0] / ts[-1] are read by the kernel itself: no transfer, no synchronisation before the launch
            if fresh:
                out.zero_()
            _lib.call("evk_voxel_from_events_f32", D.ptr(xd), D.ptr(yd), D.ptr(td), D.ptr(pd), xd.shape[0], B, H, W,
                      D.ptr(out), oob.ptr if oob is not None else None, D.stream())
            return out
        else:                    # the loaders' (ts - ts_0).float(): subtraction in float64, then float32
            import numpy as np
            a, b = D.ends(native.t)
            t_first, t_last = float(np.float32(a - native.t_offset)), float(np.float32(b - native.t_offset))
    if native is not None:
        xd, yd, td, pd = native.widen()
    if fresh:
        out.zero_()
    _lib.call("evk_voxel_f32", D.ptr(xd), D.ptr(yd), D.ptr(td), D.ptr(pd), xd.shape[0], t_first, t_last, B, H, W,
              D.ptr(out), oob.ptr if oob is not None else None, D.stream())
    return out


def iwe_tile_shape(dom_h, dom_w):
    """Tile (log2 w, log2 h) for the IWE kernel: the largest tile that still gives >= ~4 workgroups per CU."""
    for tw, th in ((5, 5), (5, 4), (4, 4)):
        if -(-dom_w // (1 << tw)) * -(-dom_h // (1 << th)) >= 1000:
            return tw, th
    return 4, 4


def _iwe_window(t_first, t_ref, vx, vy, tw, th, planes=3):
    """Time slices and LDS window for the tiled IWE kernel from the flow displacement over the stream."""
    import math
    Dx, Dy = abs((t_first - t_ref) * vx), abs((t_first - t_ref) * vy)
    wmax = _WIN_MAX[planes]
    S = max(1, int(math.ceil(max(Dx / (wmax - tw - 4), Dy / (wmax - th - 4)))))
    rnd = lambda v: min(wmax, (int(v) + 3) // 4 * 4)
    return S, rnd(tw + math.ceil(Dx / S) + 4), rnd(th + math.ceil(Dy / S) + 4)


def iwe_plan(ev, t_ref, vx, vy, bounds_w, bounds_h, ch, cw, flags, impl=None, batch=None):
    """Launch plan of the tiled IWE kernel for DeviceEvents `ev` and this flow, or None when the direct kernel must be
    used (float64 columns, unaligned views, too few events, a flow so large that the gather would test too many
    windows).  Buckets the events on first use (cached on `ev`).  batch = (vxs, vys): three nearby flows evaluated in
    one pass (the window is sized for their envelope; vx, vy are then ignored)."""
    import math
    impl = impl or default_impl()
    vxs, vys = ((vx,), (vy,)) if batch is None else batch
    native = ev.native if ev._cols is None else None     # on-disk dtypes not widened yet: bucket them as they are
    # (the SECOND evaluation of the same DeviceEvents is evidence enough: a loop of the caller's own, scipy driven by hand.
    # One count per evaluation: a fused call that declines takes its count back, its direct route plans again)
    ev._iwe_plans += 1
    min_events = TILED_MIN_EVENTS_IWE_REUSED if (ev.many_evaluations or ev._iwe_plans > 1) else TILED_MIN_EVENTS_IWE
    if native is not None:
        tileable = impl != "direct" and native.aligned() and (impl == "tiled" or native.n >= min_events)
    else:
        tileable = can_tile((ev.x, ev.y, ev.t, ev.p), impl, min_events)
    if not (tileable and all(math.isfinite(v) for v in tuple(vxs) + tuple(vys))):
        return None
    dom_h = max(int(bounds_h) + 1, ch)
    dom_w = max(int(bounds_w) + 1, cw)
    tw, th = iwe_tile_shape(dom_h, dom_w)
    if _lib.lib().evk_bucket_num_tiles(dom_h, dom_w, tw, th) <= 0:   # more than 8192 tiles: direct kernel
        return None
    t_first = ev.t_at(0)
    planes = 3 if (flags & _lib.EVK_IWE_GRADIENT or batch is not None) else 1
    span = abs(t_first - t_ref)
    # the largest displacement over the flows sizes the time slices and the windows (every flow of a batch has its own
    # window origin: evk_tiled.hip, MODE 2)
    Dx = max(abs(v) for v in vxs) * span
    Dy = max(abs(v) for v in vys) * span
    S, win_w, win_h = _iwe_window(0.0, 1.0, Dx, Dy, 1 << tw, 1 << th, planes)
    # windows the gather kernel must test per pixel; huge flows (line-search overshoots) use the direct kernel
    cand = (math.ceil((Dx + win_w) / (1 << tw)) + 1) * (math.ceil((Dy + win_h) / (1 << th)) + 1) * S
    if S > 64 or cand > 128:
        return None
    key = (1, dom_h, dom_w, tw, th, FORCE["iwe_records"])
    bk = ev._buckets.get(key)
    if bk is None:
        # (round 6) ONE bucketing call also decides the record format and delivers max |p|: the histogram pass reads the
        # polarities too, the scatter writes compact records directly when every event allows it
        legacy = bool(FORCE.get("legacy_scatter"))      # (A/B: rounds 1-5's three kernels + the separate compaction pass)
        wc = want_compact(1, len(ev), tw, th) and not legacy
        if native is not None:
            bk = bucket_events(None, None, None, None, 1, dom_h, dom_w, tw, th, native=native, stats=not legacy, compact=wc)
        else:
            bk = bucket_events(ev.x, ev.y, ev.t, ev.p, 1, dom_h, dom_w, tw, th, stats=not legacy, compact=wc)
        ev._buckets[key] = bk.compact()
        if ev._p_absmax is None and bk.p_absmax is not None:
            ev._p_absmax = bk.p_absmax
    skey = (bk.ntiles, bk.n, S, planes, win_w, win_h)
    nbytes = _staging_bytes.get(skey)
    if nbytes is None:
        nbytes = _staging_bytes[skey] = int(_lib.lib().evk_iwe_tiled_staging_bytes(*skey))
    staging = _buf("iwe_staging", nbytes, ev.device)
    # argument prefix shared by evk_iwe_linvel_tiled_f32 and evk_cmax_variance_tiled_f32 (the batch entry points take
    # two host arrays instead of the scalars vx, vy)
    if batch is None:
        flow = (vx, vy)
        keep = None
    else:
        import numpy as np
        keep = (np.ascontiguousarray(vxs, dtype=np.float64), np.ascontiguousarray(vys, dtype=np.float64))
        flow = (D.host_ptr(keep[0]), D.host_ptr(keep[1]))
    # bound of any accumulator cell: every event in one pixel, weight |p| * p_scale (* |dt| for the derivative planes);
    # lets the kernel accumulate in 64-bit fixed point (FORCE["iwe_fixed"] = False keeps float64 accumulation)
    p_bound, dt_bound = 0.0, 0.0
    if FORCE["iwe_fixed"]:
        p_bound = ev.p_absmax() * abs(float(ev.p_scale))
        dt_bound = max(span, abs(ev.t_at(-1) - t_ref))
    head = (D.ptr(bk.records), D.ptr(bk.bucket_start), bk.n, dom_h, dom_w, tw, th, S, win_w, win_h, t_first, t_ref) + \
        flow + (bounds_w, bounds_h, ch, cw, flags | bk.iwe_flag, float(ev.p_scale), p_bound, dt_bound)
    return {"head": head, "staging": staging, "staging_bytes": nbytes, "buckets": bk, "keep": keep}


def iwe_linvel(ev, t_ref, vx, vy, bounds_w, bounds_h, ch, cw, flags, iwe, diwe, impl=None):
    """Fused get_iwe for the linear-flow model on DeviceEvents `ev`, accumulating into iwe (ch, cw) / diwe (2, ch, cw)."""
    import torch
    plan = iwe_plan(ev, t_ref, vx, vy, bounds_w, bounds_h, ch, cw, flags, impl)
    if plan is not None:
        _lib.call("evk_iwe_linvel_tiled_f32", *plan["head"], D.ptr(plan["staging"]), plan["staging_bytes"], D.ptr(iwe),
                  D.ptr(diwe), D.stream())
        return
    fn = "evk_iwe_linvel_f32" if ev.dtype == torch.float32 else "evk_iwe_linvel_f64"
    _lib.call(fn, D.ptr(ev.x), D.ptr(ev.y), D.ptr(ev.t), D.ptr(ev.p), len(ev), t_ref, vx, vy, bounds_w, bounds_h, ch, cw,
              flags, float(ev.p_scale), D.ptr(iwe), D.ptr(diwe), D.stream())


_spill = {}


def _spill_pair(device, planes, ch, cw):
    """The pair of spill images of the fused evaluation (include/evk.h, evk_cmax_variance_tiled_f32): zeroed once, kept
    per stream and image shape -> [tensor, parity of the LAST successful call].  A call uses parity ^ 1 and commits it
    through _spill_call only when it has been enqueued."""
    import torch
    key = (device.index, D.stream_id(device), planes, ch, cw)
    st = _spill.get(key)
    if st is None:
        st = _spill[key] = [torch.zeros(2 * planes * ch * cw, dtype=torch.float32, device=device), 0]
    return st


def _spill_call(st, call):
    """Run one fused evaluation and toggle the spill parity only once it has been enqueued.  If the call fails (launch
    error, argument error, KeyboardInterrupt) the gather that zeroes the other image never ran: both images are zeroed
    and the parity reset, so that no stale out-of-window contributions leak into the next evaluation."""
    try:
        call()
    except BaseException:
        if st is not None:
            st[0].zero_()
            st[1] = 0
        raise
    if st is not None:
        st[1] ^= 1


def _retarget(c, S, win_w, win_h, Dx, Dy, device):
    """A cached evaluation call whose flow needs another window (time slices S, LDS window win_w x win_h) than the one it was
    marshalled for: patch the three geometry arguments and the staging buffer IN PLACE instead of building the plan and its
    ~36 ctypes arguments again.  A BFGS run walks through a handful of window sizes (the flow grows from 0 to the optimum), so
    almost every pass of a cold run used to take the slow path (12 of 14 at configs[2]: tools/bfgs_profile.py).  False when the
    tiled kernels cannot take this flow at all (the caller then goes through iwe_plan, which falls back to the direct kernel)."""
    import math
    tw, th = c["tile"]
    cand = (math.ceil((Dx + win_w) / (1 << tw)) + 1) * (math.ceil((Dy + win_h) / (1 << th)) + 1) * S
    if S > 64 or cand > 128:
        return False
    skey = c["skey_head"] + (S, c["planes"], win_w, win_h)
    nbytes = _staging_bytes.get(skey)
    if nbytes is None:
        nbytes = _staging_bytes[skey] = int(_lib.lib().evk_iwe_tiled_staging_bytes(*skey))
    staging = _buf("iwe_staging", nbytes, device)
    args, i = c["args"], c["i_staging"]
    args[7], args[8], args[9] = S, win_w, win_h
    args[i], args[i + 1] = D.ptr(staging), nbytes
    c["win"], c["staging"], c["staging_bytes"] = (S, win_w, win_h), staging, nbytes
    return True


def cmax_variance(ev, t_ref, vx, vy, bounds_w, bounds_h, ch, cw, flags, weights, radius, post_flags, buf, out, scratch,
                  scratch_bytes, impl=None, host_out=None):
    """One-call objective evaluation (evk_cmax_variance_tiled_f32) into `out` (4 doubles); returns False when the
    tiled plan is not applicable (the caller then composes the direct kernels).  host_out = numpy float64[4]: the call
    also brings the results to the host and synchronises itself.
    The ~36 marshalled arguments of the call are cached on `ev` per (geometry, buffers): a BFGS loop evaluates the same
    events hundreds of times and only vx, vy and the spill parity change, which takes ~10 us of Python off every
    evaluation (84 -> 74 us at 10 M events)."""
    import math
    ckey = (t_ref, bounds_w, bounds_h, ch, cw, flags, radius, post_flags, impl or default_impl(), FORCE["iwe_fixed"],
            FORCE["iwe_records"], ev.p_scale)
    cache = ev.__dict__.setdefault("_cmax_calls", {})
    c = cache.get(ckey)
    if c is not None and c["buf"] is buf and c["out"] is out and c["scratch"] is scratch and c["weights"] is weights \
            and c["host_out"] is host_out and math.isfinite(vx) and math.isfinite(vy):
        span, tw, th, planes = c["geo"]
        S, win_w, win_h = _iwe_window(0.0, 1.0, abs(vx) * span, abs(vy) * span, 1 << tw, 1 << th, planes)
        if ((S, win_w, win_h) == c["win"] and _buf("iwe_staging", c["staging_bytes"], buf.device) is c["staging"]) or \
                _retarget(c, S, win_w, win_h, abs(vx) * span, abs(vy) * span, buf.device):
            args = c["args"]
            args[12], args[13] = vx, vy
            st = c["spill"]
            if st is not None:
                args[c["i_parity"]] = st[1] ^ 1
            args[-1] = D.stream()
            _spill_call(st, lambda: _lib.check(c["fn"](*args), "evk_cmax_variance_tiled_f32"))
            ev.__dict__["_cmax_last_single"] = c
            return True
    plan = iwe_plan(ev, t_ref, vx, vy, bounds_w, bounds_h, ch, cw, flags, impl)
    if plan is None:
        ev._iwe_plans -= 1                # (the caller's direct-kernel route plans the SAME evaluation again: iwe_linvel)
        return False
    planes = 3 if flags & _lib.EVK_IWE_GRADIENT else 1
    sp_state = _spill_pair(buf.device, planes, ch, cw)
    spill, parity = (sp_state[0], sp_state[1] ^ 1) if sp_state is not None else (None, 0)
    args = list(plan["head"]) + [D.host_ptr(weights) if weights is not None else None, radius, post_flags,
                                 D.ptr(plan["staging"]), plan["staging_bytes"], D.ptr(buf), D.ptr(out), D.ptr(scratch),
                                 scratch_bytes, D.ptr(spill), parity,
                                 D.host_ptr(host_out) if host_out is not None else None, D.stream()]
    fn = getattr(_lib.lib(), "evk_cmax_variance_tiled_f32")
    _spill_call(sp_state, lambda: _lib.check(fn(*args), "evk_cmax_variance_tiled_f32"))
    head = plan["head"]
    cache[ckey] = {"fn": fn, "args": args, "buf": buf, "out": out, "scratch": scratch, "weights": weights,
                   "host_out": host_out, "staging": plan["staging"], "staging_bytes": plan["staging_bytes"],
                   "win": (head[7], head[8], head[9]), "geo": (abs(head[10] - head[11]), head[5], head[6], planes),
                   "spill": sp_state, "spill_key": (planes, ch, cw), "tile": (head[5], head[6]), "planes": planes,
                   "skey_head": (plan["buckets"].ntiles, plan["buckets"].n),
                   "i_staging": len(head) + 3, "ckey": ckey,
                   "i_parity": len(args) - 3, "keep": (plan, spill)}
    ev.__dict__["_cmax_last_single"] = cache[ckey]
    return True


def cmax_variance_batch3(ev, t_ref, vxs, vys, bounds_w, bounds_h, ch, cw, flags, weights, radius, buf, out12, scratch,
                         scratch_bytes, impl=None, host_out=None):
    """f at three nearby flows in one pass over the events (evk_cmax_variance_batch3_tiled_f32) -> out12 (3 x 4
    doubles); False when the tiled plan is not applicable.  host_out = numpy float64[12]: the call brings the results to
    the host and synchronises itself.  As cmax_variance, the marshalled arguments are cached on `ev` per (geometry,
    buffers): a line search (events_cmax.evk_bfgs: three step lengths per pass) changes only the six flow components, which
    live in two arrays the call reads in place (round 5: ~50 us of Python off every three-flow pass)."""
    import math
    import numpy as np
    ckey = ("batch3", t_ref, bounds_w, bounds_h, ch, cw, flags, radius, impl or default_impl(), FORCE["iwe_fixed"],
            FORCE["iwe_records"], ev.p_scale)
    cache = ev.__dict__.setdefault("_cmax_calls", {})
    c = cache.get(ckey)
    if c is not None and c["buf"] is buf and c["scratch"] is scratch and c["weights"] is weights \
            and all(math.isfinite(v) for v in tuple(vxs) + tuple(vys)):
        span, tw, th = c["geo"]
        Dx, Dy = max(abs(v) for v in vxs) * span, max(abs(v) for v in vys) * span
        S, win_w, win_h = _iwe_window(0.0, 1.0, Dx, Dy, 1 << tw, 1 << th, 3)
        if ((S, win_w, win_h) == c["win"] and _buf("iwe_staging", c["staging_bytes"], buf.device) is c["staging"]) or \
                _retarget(c, S, win_w, win_h, Dx, Dy, buf.device):
            c["vx"][:] = vxs
            c["vy"][:] = vys
            args, st = c["args"], c["spill"]
            if st is not None:
                args[c["i_parity"]] = st[1] ^ 1
            args[-7] = D.ptr(out12)          # (the samplers hand in a different slice of their result buffer per trio)
            args[-2] = D.host_ptr(host_out) if host_out is not None else None
            c["host_res"] = host_out
            args[-1] = D.stream()
            _spill_call(st, lambda: _lib.check(c["fn"](*args), "evk_cmax_variance_batch3_tiled_f32"))
            ev.__dict__["_cmax_last_b3"] = c
            return True
    plan = iwe_plan(ev, t_ref, None, None, bounds_w, bounds_h, ch, cw, flags, impl, batch=(vxs, vys))
    if plan is None:
        ev._iwe_plans -= 1                # (as cmax_variance: the flows are then evaluated one by one)
        return False
    st = _spill_pair(buf.device, 3, ch, cw)
    spill, parity = (st[0], st[1] ^ 1) if st is not None else (None, 0)
    args = list(plan["head"]) + [D.host_ptr(weights) if weights is not None else None, radius, D.ptr(plan["staging"]),
                                 plan["staging_bytes"], D.ptr(buf), D.ptr(out12), D.ptr(scratch), scratch_bytes, D.ptr(spill),
                                 parity, D.host_ptr(host_out) if host_out is not None else None, D.stream()]
    fn = getattr(_lib.lib(), "evk_cmax_variance_batch3_tiled_f32")
    _spill_call(st, lambda: _lib.check(fn(*args), "evk_cmax_variance_batch3_tiled_f32"))
    head = plan["head"]
    cache[ckey] = {"fn": fn, "args": args, "buf": buf, "scratch": scratch, "weights": weights,
                   "staging": plan["staging"], "staging_bytes": plan["staging_bytes"],
                   "win": (head[7], head[8], head[9]), "geo": (abs(head[10] - head[11]), head[5], head[6]),
                   "spill": st, "i_parity": len(args) - 3, "vx": plan["keep"][0], "vy": plan["keep"][1], "keep": (plan, spill),
                   "tile": (head[5], head[6]), "planes": 3, "skey_head": (plan["buckets"].ntiles, plan["buckets"].n),
                   "spill_key": (3, ch, cw), "i_staging": len(head) + 2, "host_res": host_out, "ckey": ckey}
    ev.__dict__["_cmax_last_b3"] = cache[ckey]
    return True


def cmax_variance_entry(ev, post_flags, single, obj):
    """The cached call of the LAST cmax_variance (single=True: it must be the one with `post_flags`) / cmax_variance_batch3 on
    `ev`, for a loop that repeats it with other flows (objectives.bind_fast); None when there is none, or when it does not
    bring its results to the host itself."""
    c = ev.__dict__.get("_cmax_last_single" if single else "_cmax_last_b3")
    if c is None:
        return None
    if single:
        return c if (c["ckey"][7] == post_flags and c.get("host_out") is not None) else None
    return c if c.get("host_res") is not None else None


def _entry_valid(c):
    """The persistent buffers a cached call was marshalled with are still the ones the layer would hand out now: the image buffer
    (grow-only: a larger image in between replaces it), the spill pair (release_scratch, a failed call) and the reduction
    scratch of the current stream."""
    buf = c["buf"]
    dev = buf.device
    sid = D.stream_id(dev)
    if _persist.get(("iwe_buf", dev.index, sid)) is not buf:
        return False
    sc = D._scratch.get((dev.index, sid, "reduce"))
    if sc is None or sc[0] is not c["scratch"]:
        return False
    st = c["spill"]
    return st is None or _spill.get((dev.index, sid) + c["spill_key"]) is st


def _again(c, name):
    args, st = c["args"], c["spill"]
    if st is not None:
        args[c["i_parity"]] = st[1] ^ 1
    args[-1] = D.stream()
    _spill_call(st, lambda: _lib.check(c["fn"](*args), name))


def cmax_variance_again(c, vx, vy):
    """Repeat the cached evaluation `c` at another flow -> its host result array (4 doubles, filled when the call returns), or
    None when the tiled kernels cannot take this flow or the persistent buffers have been replaced."""
    import math
    if not (math.isfinite(vx) and math.isfinite(vy)) or not _entry_valid(c):
        return None
    span, tw, th, planes = c["geo"]
    Dx, Dy = abs(vx) * span, abs(vy) * span
    S, win_w, win_h = _iwe_window(0.0, 1.0, Dx, Dy, 1 << tw, 1 << th, planes)
    dev = c["buf"].device
    if not (((S, win_w, win_h) == c["win"] and _buf("iwe_staging", c["staging_bytes"], dev) is c["staging"]) or
            _retarget(c, S, win_w, win_h, Dx, Dy, dev)):
        return None
    c["args"][12], c["args"][13] = vx, vy
    _again(c, "evk_cmax_variance_tiled_f32")
    return c["host_out"]


def cmax_variance_batch3_again(c, vxs, vys):
    """The same for the three-flow call -> its host result array (12 doubles)."""
    import math
    if not all(math.isfinite(v) for v in vxs + vys) or not _entry_valid(c):
        return None
    span, tw, th = c["geo"]
    Dx, Dy = max(abs(v) for v in vxs) * span, max(abs(v) for v in vys) * span
    S, win_w, win_h = _iwe_window(0.0, 1.0, Dx, Dy, 1 << tw, 1 << th, 3)
    dev = c["buf"].device
    if not (((S, win_w, win_h) == c["win"] and _buf("iwe_staging", c["staging_bytes"], dev) is c["staging"]) or
            _retarget(c, S, win_w, win_h, Dx, Dy, dev)):
        return None
    c["vx"][:] = vxs
    c["vy"][:] = vys
    _again(c, "evk_cmax_variance_batch3_tiled_f32")
    return c["host_res"]


def cmax_bfgs(ev, t_ref, x0, bounds_w, bounds_h, ch, cw, flags, weights, radius, post_flags, buf, out12, scratch, scratch_bytes,
              opts, trace_cap, impl=None):
    """The whole quasi-Newton optimisation in ONE library call (evk_cmax_bfgs_variance_tiled_f32: events_cmax.evk_bfgs with
    its arithmetic, window planning and result polls in C) -> numpy float64 [x0, x1, f, accepted points, event passes,
    status, then (x0, x1, f, g0, g1) per accepted point], or None when the tiled plan does not apply to these events.
    flags: EVK_IWE_ABS_POLARITY or 0; opts: [xtol, gtol, ftol, maxiter, numeric_grads, unit_first].  The loop may use two time
    slices of the largest LDS window (flows up to 2 x (48 - tile - 4) px of displacement over the stream); a trial flow
    beyond that ends it with status 1 and the caller runs its own loop."""
    import numpy as np
    G = _lib.EVK_IWE_GRADIENT
    plan = iwe_plan(ev, t_ref, float(x0[0]), float(x0[1]), bounds_w, bounds_h, ch, cw, (flags & ~G) | G, impl)
    if plan is None:
        ev._iwe_plans -= 1                # (the caller's own loop evaluates the same start again)
        return None
    head, bk = plan["head"], plan["buckets"]
    cap = max(int(plan["staging_bytes"]),
              int(_lib.lib().evk_iwe_tiled_staging_bytes(bk.ntiles, bk.n, 2, 3, _WIN_MAX[3], _WIN_MAX[3])))
    staging = _buf("iwe_staging", cap, ev.device)
    st = _spill_pair(ev.device, 3, ch, cw)
    parity = np.array([st[1]], dtype=np.int32)
    x0a = np.ascontiguousarray(x0, dtype=np.float64)
    optsa = np.ascontiguousarray(opts, dtype=np.float64)
    res = np.zeros(6 + 5 * trace_cap, dtype=np.float64)
    try:
        _lib.call("evk_cmax_bfgs_variance_tiled_f32", head[0], head[1], head[2], head[3], head[4], head[5], head[6], head[10],
                  head[11], head[14], head[15], head[16], head[17], head[18] & ~G, head[19], head[20], head[21],
                  D.host_ptr(weights) if weights is not None else None, radius, post_flags, D.ptr(staging), staging.numel(),
                  D.ptr(buf), D.ptr(out12), D.ptr(scratch), scratch_bytes, D.ptr(st[0]), D.host_ptr(parity), D.host_ptr(x0a),
                  D.host_ptr(optsa), D.host_ptr(res), trace_cap, D.stream())
    except BaseException:
        # (as _spill_call: a pass that failed half-way leaves the pair in an unknown state)
        st[0].zero_()
        st[1] = 0
        raise
    st[1] = int(parity[0]) & 1
    return res


WARM_MS = 40.0


def _time_ms(fn, reps):
    """Average duration of fn's launches: `reps` back-to-back calls between ONE pair of HIP events on the launch
    stream.  (An event pair around every single launch adds the ~7-10 us the command processor needs between the
    event's timestamp write and the dispatch, which made a 62 us kernel read as 72 us against rocprofv3's
    kernel-trace; back to back the GPU stays fed, the host enqueue being shorter than the kernels timed here.)"""
    import time
    import torch
    # Round 6: the device's clocks ramp for tens of milliseconds after an idle phase (a 50 M-event stream generated on the host
    # is seconds of idleness): the first ~10 launches of a fresh burst ran 15-20 % slower than the steady state (tile kernel at
    # 50 M events / 720p: 128 us in the first 3 ms, 105-107 us from the third burst on; tools/warm_probe.py).  What a loop of
    # such calls sustains is the steady state, so every timing is preceded by WARM_MS of the same launches.
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < WARM_MS:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return float(e0.elapsed_time(e1) / reps)


def time_voxel_kernels(sets, t_first, t_last, B, H, W, impl=None, reps=10):
    """HIP-event timing (on the launch stream) of the kernels one voxel call launches, for bench.py's roofline.  `sets` =
    one or more (x, y, t, p) column tuples: the timed calls rotate over them (several sets that together exceed the
    Infinity Cache make every call read its events from HBM)."""
    import torch
    impl = impl or default_impl()
    xd = sets[0][0]
    out = torch.zeros((B, H, W), dtype=torch.float32, device=xd.device)
    it = [0]

    def cols():
        it[0] += 1
        return sets[it[0] % len(sets)]
    reps = max(reps, len(sets)) // len(sets) * len(sets)
    shape2 = voxel2_shape(H, W, B)
    if not (can_tile(sets[0], impl) and shape2 is not None):
        ms = _time_ms(lambda: voxel_f32(*cols(), t_first, t_last, B, H, W, out, None, impl="direct"), reps)
        return {"impl": "direct", "dominant": "k_voxel_f32", "dominant_ms": ms, "total_ms": ms,
                "kernels_ms": {"k_voxel_f32": round(ms, 4)}, "kernels_ms_exact": {"k_voxel_f32": ms}}
    total = _time_ms(lambda: voxel_f32(*cols(), t_first, t_last, B, H, W, out, None, impl="tiled", fresh=True), reps)
    n = xd.shape[0]
    run2 = lambda stage: voxel2(cols(), None, n, t_first, t_last, B, H, W, *shape2, out, None, True, stage=stage)  # noqa: E731
    # (the tile kernel alone re-reads the records the LAST partition left: time it right behind a partition of every set)
    ms = {"k_part_sorted": _time_ms(lambda: run2(_lib.EVK_VOXEL2_PARTITION_ONLY), reps)}
    ms["k_voxel_tiles2"] = _time_ms(lambda: run2(_lib.EVK_VOXEL2_TILES_ONLY), reps)
    dom = max(ms, key=ms.get)
    return {"impl": "one-pass partition, tiles %dx%d" % shape2, "dominant": dom, "dominant_ms": ms[dom], "total_ms": total,
            "kernels_ms": {k: round(v, 4) for k, v in ms.items()}, "kernels_ms_exact": dict(ms)}
