"""
Event-sharded data parallelism (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI).

Every accumulator of the hot path (event image, voxel grid, IWE, dIWE) is a SUM over events, so any partition of the
events gives partial grids whose element-wise sum is the result.  Each rank therefore processes a
contiguous slice of the time-sorted stream with the kernels of libevk.so and the only exchange step is ONE all-reduce of
the output grid (18.4 MB for a 5x720x1280 voxel grid, 11 MB for IWE+dIWE) -- nothing per-event ever crosses xGMI.
The two scalars every rank must agree on BEFORE its pass (ts[0], ts[-1] of the whole stream: t_norm of the voxel grid,
reference time of the warp) are exchanged once with two scalar all-reduces.
"""
import ctypes
import os

import numpy as np
import torch


def _dist():
    import torch.distributed as dist
    return dist


def is_distributed(group=None):
    dist = _dist()
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def shard_bounds(n, rank, world):
    """Contiguous, balanced slice [lo, hi) of n time-sorted events for `rank` of `world` (sizes differ by <= 1)."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _collective_device(group=None):
    """Tensors handed to a collective must live where the backend works: the GPU for nccl (= RCCL), the host for gloo."""
    dist = _dist()
    if dist.get_backend(group) == "nccl":
        from . import _device as D
        return D.require_gpu()
    return torch.device("cpu")


class _staged:
    """`with _staged(t, group) as h:` -- h is the tensor a collective of `group`'s backend can take: `t` itself for nccl
    (= RCCL, device buffers) or for host tensors; for gloo and a DEVICE tensor a host copy that is written back to `t` on
    exit.  gloo moves host memory only (its device-tensor support covers two collectives and depends on the build), so an
    event-sharded job on gloo -- two ranks sharing one GPU in the tests, machines without RCCL -- stages its grids
    through the host: the same sums, PCIe instead of xGMI."""

    def __init__(self, t, group=None, write_back=True):
        self.t, self.write_back = t, write_back
        self.host = t.is_cuda and _dist().get_backend(group) != "nccl"

    def __enter__(self):
        self.h = self.t.cpu() if self.host else self.t
        return self.h

    def __exit__(self, *exc):
        if self.host and self.write_back and exc[0] is None:
            self.t.copy_(self.h)
        return False


# ---- the C-ABI collective (include/evk.h: evk_comm_*, evk_allreduce_*) -------------------------------------------------
_comms = {}


def collective():
    """'torch' (default): torch.distributed's all_reduce (backend nccl = RCCL).  'evk': libevk.so's own RCCL binding,
    the entry points a caller without torch uses (EVK_COLLECTIVE=evk); torch.distributed then only ships the 128-byte
    communicator id to the ranks once."""
    return os.environ.get("EVK_COLLECTIVE", "torch")


def evk_comm(group=None):
    """libevk communicator of `group` (created on first use: rank 0 makes the id, it is broadcast, every rank joins)."""
    from . import _lib
    key = group if group is not None else 0      # the group OBJECT: keeps it alive, so its identity cannot be recycled
    c = _comms.get(key)
    if c is None:
        dist = _dist()
        L = _lib.lib()
        nbytes = L.evk_comm_unique_id_bytes()
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        else:
            rank, world = 0, 1
        box = [None]
        if rank == 0:
            raw = ctypes.create_string_buffer(nbytes)
            _lib.call("evk_comm_unique_id", raw)
            box[0] = raw.raw
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        handle = ctypes.c_void_p()
        _lib.call("evk_comm_init", ctypes.create_string_buffer(box[0], nbytes), rank, world, ctypes.byref(handle))
        c = _comms[key] = handle
    return c


def evk_comm_destroy(group=None):
    from . import _lib
    c = _comms.pop(group if group is not None else 0, None)
    if c is not None:
        _lib.call("evk_comm_destroy", c)


def global_time_range(t_first_local, t_last_local, group=None, device=None):
    """(min over ranks of t_first, max over ranks of t_last): the ts[0] / ts[-1] of the whole stream
    (voxel_grid.py:133-134; objectives.py:186).  An empty shard passes (+inf, -inf).  ONE collective: MAX of
    (-t_first, t_last)."""
    if not is_distributed(group):
        return float(t_first_local), float(t_last_local)
    dist = _dist()
    v = torch.tensor([-float(t_first_local), float(t_last_local)], dtype=torch.float64, device=_collective_device(group))
    dist.all_reduce(v, op=dist.ReduceOp.MAX, group=group)
    lo, hi = v.tolist()
    return -float(lo), float(hi)


def voxel_collective():
    """How the voxel grid is summed over the ranks: 'allreduce' (default: one RCCL all-reduce, the library picks ring / tree)
    or 'rsag' (EVK_VOXEL_COLLECTIVE=rsag): an explicit reduce-scatter followed by an all-gather -- the all-links form of
    SURVEY.md section 5 for the 18.4 MB grid of configs[4]: every rank reduces 1/N of the grid and all 7 xGMI links of a
    GPU carry traffic in both phases.  bench.py --gpus N reports both."""
    return os.environ.get("EVK_VOXEL_COLLECTIVE", "allreduce")


def reduce_scatter_all_gather_sum_(grid, group=None):
    """In-place SUM of `grid` over the ranks as reduce-scatter + all-gather (the grid is viewed flat and cut into
    world_size equal chunks; a tail that does not divide is handled by a small all-reduce)."""
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if grid.is_cuda and dist.get_backend(group) != "nccl":
        with _staged(grid, group) as h:
            reduce_scatter_all_gather_sum_(h, group)
        return grid
    flat = grid.view(-1)
    chunk = flat.numel() // world
    if chunk:
        body = flat[: chunk * world]
        mine = torch.empty(chunk, dtype=flat.dtype, device=flat.device)
        if dist.get_backend(group) == "nccl":
            dist.reduce_scatter_tensor(mine, body, op=dist.ReduceOp.SUM, group=group)
            dist.all_gather_into_tensor(body, mine, group=group)
        else:   # gloo (CPU tests) has no reduce-scatter: one reduce per destination rank is the same exchange
            for j in range(world):
                part = body[j * chunk:(j + 1) * chunk].clone()
                dist.reduce(part, dst=dist.get_global_rank(group, j) if group is not None else j, op=dist.ReduceOp.SUM, group=group)
                if j == rank:
                    mine.copy_(part)
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine, group=group)
            body.copy_(torch.cat(parts))
    if chunk * world < flat.numel():
        dist.all_reduce(flat[chunk * world:], op=dist.ReduceOp.SUM, group=group)
    return grid


def all_reduce_sum_(grid, group=None, force=False, form=None):
    """In-place SUM all-reduce of an output grid (no-op for a single process unless force=True).  form='rsag': as an explicit
    reduce-scatter + all-gather (voxel_collective())."""
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()) or not (force or is_distributed(group)):
        return grid
    if form == "rsag" and grid.is_contiguous():
        return reduce_scatter_all_gather_sum_(grid, group)
    if collective() == "evk" and grid.is_cuda and grid.is_contiguous() and grid.dtype in (torch.float32, torch.int32):
        from . import _device as D
        from . import _lib
        fn = "evk_allreduce_f32" if grid.dtype == torch.float32 else "evk_allreduce_i32"
        _lib.call(fn, D.ptr(grid), grid.numel(), evk_comm(group), D.stream())
        return grid
    with _staged(grid, group) as h:
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
    return grid


def _raise_everywhere(oob_state, exc_type, msg, group, local_error=None):
    """A data-dependent error must surface on EVERY rank (a rank that raised alone would leave the others waiting in the
    next collective): the per-rank counts of dropped events are summed over the ranks and all of them raise.
    local_error: an exception THIS rank's shard raised before its kernels ran (a shard the library refuses, float64 columns:
    round 6) -- the rank has joined every collective with a zero contribution; here every rank learns of it and raises: the
    failing ranks their own exception, the others a RuntimeError that says so."""
    # (earlier DEFERRED reports of this stream are folded in here, quietly: polling them in OobCounter's constructor could
    # raise on this rank alone, between two collectives, and leave the other ranks waiting in the all-reduce)
    local = oob_state.drain() + oob_state._advance(int(oob_state.counter.item()) & 0xFFFFFFFF)
    cnt = torch.tensor([local, 1 if local_error is not None else 0], dtype=torch.int64, device=oob_state.counter.device)
    total = cnt.to(_collective_device(group)) if is_distributed(group) else cnt
    if is_distributed(group):
        _dist().all_reduce(total, op=_dist().ReduceOp.SUM, group=group)
    total, failed = (int(v) for v in total.tolist())
    if local_error is not None:
        raise local_error
    if failed:
        raise RuntimeError("the sharded call failed on %d other rank(s) before their kernels ran (their shards were refused); "
                           "the collectives were completed with zero contributions from them" % failed)
    if total:
        raise exc_type("%s (%d offending events over all ranks, %d on this one)" % (msg, total, local))


def _local_voxel(xs, ys, ts, ps, B, sensor_size, t_first, t_last, oob=None):
    from . import _device as D
    from . import tiled
    dev = D.require_gpu()
    H, W = int(sensor_size[0]), int(sensor_size[1])
    cols = [D.to_device(a, torch.float32, dev) for a in (xs, ys, ts, ps)]
    if cols[0].shape[0] == 0:
        return torch.zeros((B, H, W), dtype=torch.float32, device=dev)
    out = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    return tiled.voxel_f32(*cols, float(t_first), float(t_last), B, H, W, out, oob, fresh=True)


def banded_exchange(bands, out, group=None):
    """`bands` yields (y_lo, y_hi, band) -- contiguous (B, rows, W) partial grids of this rank, each one yielded as soon as
    its kernel is enqueued.  Every band's SUM all-reduce is issued asynchronously right then (RCCL runs it on its own stream,
    behind the band's kernel), so it overlaps the accumulation of the bands that follow; the reduced bands are copied into
    `out` (B, H, W) at the end.  The same bytes as one all-reduce of the grid, in len(bands) pieces."""
    dist = _dist()
    live = dist.is_available() and dist.is_initialized()
    host = live and dist.get_backend(group) != "nccl"
    pending = []
    for y0, y1, band in bands:
        if host and band.is_cuda:        # gloo: the band goes through the host (_staged); the copy waits for the band's kernel
            band = band.cpu()
        work = dist.all_reduce(band, op=dist.ReduceOp.SUM, group=group, async_op=True) if live else None
        pending.append((y0, y1, band, work))
    for y0, y1, band, work in pending:
        if work is not None:
            work.wait()
        out[:, y0:y1, :].copy_(band)
    return out


def voxel_bands():
    """Row bands of the intra-call overlap of the voxel exchange (EVK_VOXEL_COLLECTIVE=bandsK, K = 2..8; default: one
    all-reduce of the whole grid after the kernels)."""
    v = voxel_collective()
    return int(v[5:]) if v.startswith("bands") and v[5:].isdigit() else 0


def events_to_voxel_torch_sharded(xs, ys, ts, ps, B, sensor_size=(180, 240), group=None, local_fn=None):
    """events_to_voxel_torch over an event stream sharded across ranks: `xs, ys, ts, ps` are THIS rank's slice.
    Returns the full (B, H, W) grid on every rank.  `local_fn(xs, ys, ts, ps, B, sensor_size, t_first, t_last)`
    computes one shard's partial grid (default: the HIP kernels; the CPU tests inject the oracle).  Out-of-range
    coordinates raise IndexError on every rank, after the collectives.
    EVK_VOXEL_COLLECTIVE=bandsK: the exchange overlaps the call's own kernels -- the tile kernel runs in K row bands and band
    k is all-reduced while band k + 1 accumulates (banded_exchange)."""
    n = len(xs)
    inf = float("inf")
    t0 = float(ts[0]) if n else inf
    t1 = float(ts[-1]) if n else -inf
    t_first, t_last = global_time_range(t0, t1, group)
    K = voxel_bands()
    H, W = int(sensor_size[0]), int(sensor_size[1])
    if local_fn is not None:
        part = local_fn(xs, ys, ts, ps, B, sensor_size, t_first, t_last)
        if K >= 2:      # (CPU tests: the banding and the assembly with the oracle's partial grid cut into row bands)
            edges = [k * H // K for k in range(K + 1)]
            out = torch.empty_like(part)
            return banded_exchange(((edges[k], edges[k + 1], part[:, edges[k]:edges[k + 1], :].contiguous()) for k in range(K)
                                    if edges[k + 1] > edges[k]), out, group)
        return all_reduce_sum_(part, group, form=voxel_collective())
    from . import _device as D
    from . import tiled
    oob = D.OobCounter(D.require_gpu(), poll=False)
    bands = None
    # Whether the exchange runs in bands -- K collectives of a band each instead of one of the grid -- must be the SAME
    # decision on every rank, or the ranks' collectives differ in count and size: it depends on K, the grid and the
    # library's tiling only, never on this rank's events.  A rank with an empty shard contributes zero bands with the same
    # edges; columns the one-pass path cannot read in place (views that are not 16-byte aligned, other dtypes) are copied.
    edges = tiled.voxel2_band_rows(H, W, B, K) if (K >= 2 and tiled.default_impl() != "direct") else None
    # Everything that can fail on THIS rank alone -- columns the single-process call refuses (float64 time stamps / polarities:
    # voxel_grid.py's dtype error), a shard the library does not take (more than 4e9 events, a geometry it refuses) -- happens
    # BEFORE the first collective, inside a try: the rank then joins every collective with a zero contribution (the peers must
    # not be left waiting in an all-reduce) and the error surfaces on all ranks afterwards (_raise_everywhere).
    local_error = None
    dev = D.require_gpu()

    def zero_bands():
        return ((y0, y1, torch.zeros((B, y1 - y0, W), dtype=torch.float32, device=dev)) for y0, y1 in edges)
    try:
        for name, col in (("ts", ts), ("ps", ps)):
            if getattr(col, "dtype", None) in (torch.float64, np.float64, np.dtype(np.float64)):
                raise RuntimeError("Index put requires the source and destination dtypes match, got Float for the destination "
                                   "and Double for the source. (%s is float64; events_to_voxel_torch raises the same)" % name)
        if edges is not None:
            if n:
                cols = [D.to_device(a, torch.float32, dev).contiguous() for a in (xs, ys, ts, ps)]
                cols = [c if c.data_ptr() % 16 == 0 else c.clone() for c in cols]
                bands = tiled.voxel2_bands(cols, n, float(t_first), float(t_last), B, H, W, K, oob)
            else:
                bands = zero_bands()
        else:
            part = _local_voxel(xs, ys, ts, ps, B, sensor_size, t_first, t_last, oob)
    except Exception as e:  # noqa: BLE001
        local_error = e
        if edges is not None:
            bands = zero_bands()
        else:
            part = torch.zeros((B, H, W), dtype=torch.float32, device=dev)
    if bands is not None:
        out = banded_exchange(bands, torch.empty((B, H, W), dtype=torch.float32, device=dev), group)
    else:
        out = all_reduce_sum_(part, group, form="rsag" if voxel_collective() == "rsag" else None)
    _raise_everywhere(oob.state, IndexError, "index out of range for voxel grid of size %s"
                      % ((B, int(sensor_size[0]), int(sensor_size[1])),), group, local_error)
    return out


def events_to_image_sharded(xs, ys, ps, sensor_size=(180, 240), group=None, local_fn=None):
    """events_to_image (nearest pixel, integer weights; image.py:28-44) over an event stream sharded across ranks: every
    rank accumulates ITS events on the (H+1, W+1) int32 canvas, ONE int32 all-reduce sums the canvases -- integer
    arithmetic, so the result is bit-identical to the single-process image whatever the sharding -- and the cropped
    float64 (H, W) image is returned on every rank.  `local_fn(xs, ys, ps, canvas_shape) -> int32 tensor` replaces the
    HIP kernel in the CPU tests."""
    xs, ys, ps = np.asarray(xs), np.asarray(ys), np.asarray(ps)
    if not (np.issubdtype(xs.dtype, np.integer) and np.issubdtype(ys.dtype, np.integer)):
        raise TypeError("only int indices permitted")
    if not (np.issubdtype(ps.dtype, np.integer) or ps.dtype == np.bool_):
        raise TypeError("the sharded event image accumulates integer weights (bit-exact int32 all-reduce)")
    shape = (int(sensor_size[0]) + 1, int(sensor_size[1]) + 1)
    if local_fn is not None:
        canvas = local_fn(xs, ys, ps, shape)
        all_reduce_sum_(canvas, group)
    else:
        from . import _device as D
        from . import _lib
        dev = D.require_gpu()
        canvas = torch.zeros(shape, dtype=torch.int32, device=dev)
        oob = D.OobCounter(dev, poll=False)
        if xs.shape[0]:
            xd, yd, wd = (D.to_device(a, torch.int32) for a in (xs, ys, ps))
            _lib.call("evk_image_nearest_i32", D.ptr(xd), D.ptr(yd), D.ptr(wd), xs.shape[0], shape[0], shape[1],
                      D.ptr(canvas), oob.ptr, D.stream())
        all_reduce_sum_(canvas, group)
        _raise_everywhere(oob.state, ValueError, "events outside the (H+1, W+1) canvas %s" % (shape,), group)
    return canvas.cpu().numpy().astype(np.float64)[0:int(sensor_size[0]), 0:int(sensor_size[1])]


def sharded_evaluate(local_iwe, finish, group=None):
    """One event-sharded objective evaluation: `local_iwe()` -> this rank's (1 | 3, H+1, W+1) IWE [+ dIWE] at the GLOBAL
    reference time, ONE in-place all-reduce, `finish(buffer)` -> the scalars (blur + reductions, replicated: identical on
    every rank).  objective_function._one_call runs it with the HIP kernels, the CPU tests with the oracle."""
    img = local_iwe()
    all_reduce_sum_(img, group, force=True)
    return finish(img)


def post_mode():
    """How the blur + reductions of a sharded objective evaluation run: 'replicated' (default: all-reduce the image, every
    rank blurs all of it) or 'rows' (EVK_SHARDED_POST=rows: all-to-all of row blocks -- half the bytes of the all-reduce per
    link --, every rank blurs its block, an 8-double all-reduce of the raw sums)."""
    return os.environ.get("EVK_SHARDED_POST", "replicated")


def row_block(ch, radius, rank, world):
    """Rows of the (ch, cw) image owned by `rank` -> (y0, y1, lo, hi): it sums rows [y0, y1) and needs rows [lo, hi) for
    the blur, i.e. `radius` halo rows on every side that is not an edge of the image."""
    y0, y1 = rank * ch // world, (rank + 1) * ch // world
    return y0, y1, max(0, y0 - radius), min(ch, y1 + radius)


def finalise_sums(sums, n_pixels, mode):
    """Global raw sums of the row blocks -> the 4 scalars of one evaluation, as k_reduce_final computes them (float64):
    mode 0 [mean, var, S v, S v^2]; mode 1 [g0, g1, mean a, S a]; mode 3 [g0, g1, mean v, var v]."""
    s = np.asarray(sums, dtype=np.float64)
    inv = 1.0 / float(n_pixels)
    mean = s[0] * inv
    if mode == 0:
        return np.array([mean, s[1] * inv - mean * mean, s[0], s[1]])
    g = [2.0 * inv * (s[3] - mean * s[1]), 2.0 * inv * (s[4] - mean * s[2])]
    if mode == 1:
        return np.array([g[0], g[1], mean, s[0]])
    mv = s[5] * inv
    return np.array([g[0], g[1], mv, s[6] * inv - mv * mv])


def sharded_evaluate_rows(local_iwe, rows_post, radius, mode, group=None):
    """One event-sharded objective evaluation with a ROW-SHARDED post-pass: `local_iwe()` -> this rank's
    (planes, ch, cw) partial image; every rank receives the partial rows of ITS block (with halo) from all ranks in one
    all-to-all and sums them; `rows_post(block, y_lo, y_hi)` -> 8 raw sums over its own rows (a tensor where the backend
    works); one 8-double all-reduce; finalise_sums on the host.  Identical scalars on every rank."""
    dist = _dist()
    img = local_iwe()
    planes, ch, cw = img.shape
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    y0, y1, lo, hi = row_block(ch, radius, rank, world)
    if world == 1:
        block = img
    else:
        blocks = [row_block(ch, radius, j, world) for j in range(world)]
        send = torch.cat([img[:, b[2]:b[3], :].reshape(-1) for b in blocks])
        in_splits = [planes * (b[3] - b[2]) * cw for b in blocks]
        mine = planes * (hi - lo) * cw
        recv = torch.empty(world * mine, dtype=img.dtype, device=img.device)
        with _staged(recv, group) as hr, _staged(send, group, write_back=False) as hs:
            dist.all_to_all_single(hr, hs, [mine] * world, in_splits, group=group)
        block = recv.view(world, planes, hi - lo, cw).sum(dim=0)   # rank order: the same sum on every run
    sums = rows_post(block.contiguous(), y0 - lo, y1 - lo)
    if world > 1:
        with _staged(sums, group) as h:
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
    return finalise_sums(sums.cpu().numpy(), ch * cw, mode)


def shard_objective(objective, t_last_global, group=None):
    """Configure a contrast-maximisation objective for event-sharded evaluation: every rank warps to the GLOBAL
    reference time ts[-1] and IWE / dIWE are all-reduced before the blur and the scalar reductions, which then run
    replicated (identical values on every rank, so the host BFGS loops stay in lock-step without further traffic)."""
    objective.distributed = True
    objective.process_group = group
    objective.t_ref = float(t_last_global)
    return objective
