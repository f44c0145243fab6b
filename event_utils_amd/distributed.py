"""
Event-sharded data parallelism (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI).

Every accumulator of the hot path (event image, voxel grid, IWE, dIWE) is a SUM over events, so any partition of the
events gives partial grids whose element-wise sum is the result.  Each rank therefore processes a
contiguous slice of the time-sorted stream with the kernels of libevk.so and the only exchange step is ONE all-reduce of
the output grid (18.4 MB for a 5x720x1280 voxel grid, 11 MB for IWE+dIWE) -- nothing per-event ever crosses xGMI.
The two scalars every rank must agree on BEFORE its pass (ts[0], ts[-1] of the whole stream: t_norm of the voxel grid,
reference time of the warp) are exchanged once with two scalar all-reduces.
"""
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


def global_time_range(t_first_local, t_last_local, group=None, device=None):
    """(min over ranks of t_first, max over ranks of t_last): the ts[0] / ts[-1] of the whole stream
    (voxel_grid.py:133-134; objectives.py:186).  An empty shard passes (+inf, -inf)."""
    if not is_distributed(group):
        return float(t_first_local), float(t_last_local)
    dist = _dist()
    lo = torch.tensor([float(t_first_local)], dtype=torch.float64, device=device)
    hi = torch.tensor([float(t_last_local)], dtype=torch.float64, device=device)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return float(lo.item()), float(hi.item())


def all_reduce_sum_(grid, group=None):
    """In-place SUM all-reduce of an output grid (no-op for a single process)."""
    if is_distributed(group):
        dist = _dist()
        dist.all_reduce(grid, op=dist.ReduceOp.SUM, group=group)
    return grid


def _local_voxel(xs, ys, ts, ps, B, sensor_size, t_first, t_last):
    from . import _device as D
    from .representations.voxel_grid import _voxel_f32_device
    dev = D.require_gpu()
    cols = [D.to_device(a, torch.float32, dev) for a in (xs, ys, ts, ps)]
    if cols[0].shape[0] == 0:
        return torch.zeros((B, int(sensor_size[0]), int(sensor_size[1])), dtype=torch.float32, device=dev)
    return _voxel_f32_device(*cols, B, sensor_size, t_first, t_last)


def events_to_voxel_torch_sharded(xs, ys, ts, ps, B, sensor_size=(180, 240), group=None, local_fn=None):
    """events_to_voxel_torch over an event stream sharded across ranks: `xs, ys, ts, ps` are THIS rank's slice.
    Returns the full (B, H, W) grid on every rank.  `local_fn(xs, ys, ts, ps, B, sensor_size, t_first, t_last)`
    computes one shard's partial grid (default: the HIP kernels; the CPU tests inject the oracle)."""
    n = len(xs)
    inf = float("inf")
    t0 = float(ts[0]) if n else inf
    t1 = float(ts[-1]) if n else -inf
    dev = xs.device if isinstance(xs, torch.Tensor) else None
    t_first, t_last = global_time_range(t0, t1, group, device=dev)
    part = (local_fn or _local_voxel)(xs, ys, ts, ps, B, sensor_size, t_first, t_last)
    return all_reduce_sum_(part, group)


def shard_objective(objective, t_last_global, group=None):
    """Configure a contrast-maximisation objective for event-sharded evaluation: every rank warps to the GLOBAL
    reference time ts[-1] and IWE / dIWE are all-reduced before the blur and the scalar reductions, which then run
    replicated (identical values on every rank, so the host BFGS loops stay in lock-step without further traffic)."""
    objective.distributed = True
    objective.process_group = group
    objective.t_ref = float(t_last_global)
    return objective
