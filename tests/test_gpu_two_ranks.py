"""GPU (-m gpu): the event-sharded path on REAL kernels with TWO processes.  The box has one GPU, so both ranks share
cuda:0 (RCCL refuses two ranks on one device; the group is gloo and distributed._staged moves the grids through the host --
the same sums over another wire).  Every rank holds a contiguous half of a 2 M-event stream and each result is compared
with the oracle ON THE WHOLE STREAM (1e-5 of the maximum; the integer image bit for bit), under every exchange form:
one all-reduce, reduce-scatter + all-gather, row bands, and the row-sharded post-pass of the objective.  One out-of-range
event on rank 1 only must raise on BOTH ranks.  (Accumulate sites that make the sharding legal: image.py:37,95,111-114,
132-135; the global reference time: objectives.py:186.)"""
import datetime
import os
import socket
import traceback

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N, H, W, B = 2_000_000, 240, 320, 5


def _stream():
    """A structured scene (edges moving at (40, -25) px/s over a uniform background): the gradient is far from zero."""
    rng = np.random.default_rng(42)
    t = np.sort(rng.uniform(0, 0.1, N))
    kind = rng.integers(0, 3, N)          # 0: background, 1: vertical edges, 2: horizontal edges
    x0 = np.where(kind == 1, rng.choice(np.arange(20, W - 20, 24), N) + rng.normal(0, 0.6, N), rng.uniform(8, W - 8, N))
    y0 = np.where(kind == 2, rng.choice(np.arange(20, H - 20, 24), N) + rng.normal(0, 0.6, N), rng.uniform(8, H - 8, N))
    x = np.clip(x0 + 40.0 * t, 1.0, W - 1.001).astype(np.float32)
    y = np.clip(y0 - 25.0 * t, 1.0, H - 1.001).astype(np.float32)
    p = np.where(kind == 0, rng.integers(0, 2, N) * 2 - 1, 1).astype(np.float32)     # edges fire one polarity
    return x, y, t.astype(np.float32), p


def _worker(rank, world, port, errs):
    try:
        _run(rank, world, port)
    except BaseException:      # (reported to the parent; the other rank runs into its collective's timeout at worst)
        errs.put((rank, traceback.format_exc()))
        raise


def _run(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["EVK_IMPL"] = "tiled"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=240))
    import event_utils_amd as E
    from event_utils_amd import distributed as DD
    from event_utils_amd import tiled
    from event_utils_amd.events import DeviceEvents
    from oracle import reference_np as R
    assert DD.is_distributed() and tiled.share_cu()
    x, y, t, p = _stream()
    lo, hi = DD.shard_bounds(N, rank, world)
    fx, fy = np.floor(x), np.floor(y)
    cols = [torch.from_numpy(a[lo:hi].copy()).cuda() for a in (fx, fy, t, p)]

    # ---- voxel grid: every exchange form against the whole-stream oracle
    ref = R.events_to_voxel_torch(fx, fy, t, p, B, sensor_size=(H, W), accum="f64")
    grids = {}
    for form in ("allreduce", "rsag", "bands2", "bands5"):
        os.environ["EVK_VOXEL_COLLECTIVE"] = form
        v = DD.events_to_voxel_torch_sharded(*cols, B, (H, W)).cpu().numpy()
        assert np.abs(v - ref).max() <= 1e-5 * np.abs(ref).max(), (form, np.abs(v - ref).max())
        grids[form] = v
    # unit polarities: every rank's partial grid is exact integers / 2^31 rounded once, so the forms agree closely
    assert np.abs(grids["rsag"] - grids["allreduce"]).max() <= 2e-6 * np.abs(ref).max()
    # an EMPTY shard on rank 1, with bands (every rank must issue the same collectives) and without
    mine = cols if rank == 0 else [c[:0] for c in cols]
    lo0, hi0 = DD.shard_bounds(N, 0, world)
    half = R.events_to_voxel_torch(fx[lo0:hi0], fy[lo0:hi0], t[lo0:hi0], p[lo0:hi0], B, sensor_size=(H, W), accum="f64")
    for form in ("bands2", "allreduce"):
        os.environ["EVK_VOXEL_COLLECTIVE"] = form
        v = DD.events_to_voxel_torch_sharded(*mine, B, (H, W)).cpu().numpy()
        assert np.abs(v - half).max() <= 1e-5 * np.abs(half).max(), form
    # columns the one-pass path cannot read in place (a view that is not 16-byte aligned) on ONE rank only
    os.environ["EVK_VOXEL_COLLECTIVE"] = "bands2"
    if rank == 1:
        pad = [torch.cat([c[:1], c]) for c in cols]
        odd = [q[1:] for q in pad]
        assert odd[0].data_ptr() % 16 != 0
    else:
        odd = cols
    v = DD.events_to_voxel_torch_sharded(*odd, B, (H, W)).cpu().numpy()
    assert np.abs(v - ref).max() <= 1e-5 * np.abs(ref).max()
    # one out-of-range event on rank 1 only: BOTH ranks raise, after the collectives, under every form
    bad = [c.clone() for c in cols]
    if rank == 1:
        bad[0][12345] = W + 3.0
    for form in ("allreduce", "rsag", "bands2"):
        os.environ["EVK_VOXEL_COLLECTIVE"] = form
        with pytest.raises(IndexError):
            DD.events_to_voxel_torch_sharded(*bad, B, (H, W))
    # a shard that is REFUSED on rank 1 only -- float64 time stamps, the dtype error events_to_voxel_torch raises -- before
    # its kernels run: rank 1 still joins every collective (zero bands / a zero grid), then BOTH ranks raise (round 6)
    f64 = [c.clone() for c in cols]
    if rank == 1:
        f64[2] = f64[2].double()
    for form in ("allreduce", "bands2"):
        os.environ["EVK_VOXEL_COLLECTIVE"] = form
        with pytest.raises(RuntimeError, match="Double for the source" if rank == 1 else "other rank"):
            DD.events_to_voxel_torch_sharded(*f64, B, (H, W))
    os.environ.pop("EVK_VOXEL_COLLECTIVE")
    DD.events_to_voxel_torch_sharded(*cols, B, (H, W))        # ... and the next call is clean
    E.check_errors()

    # ---- integer event image: int32 all-reduce, bit-exact
    xi, yi, pi = fx.astype(np.int64), fy.astype(np.int64), p.astype(np.int64)
    img = DD.events_to_image_sharded(xi[lo:hi], yi[lo:hi], pi[lo:hi], (H, W))
    assert np.array_equal(img, R.events_to_image(xi, yi, pi, sensor_size=(H, W)))
    bx = xi[lo:hi].copy()
    if rank == 1:
        bx[7] = W + 9
    with pytest.raises(ValueError):
        DD.events_to_image_sharded(bx, yi[lo:hi], pi[lo:hi], (H, W))

    # ---- objective: f and gradient at the GLOBAL reference time, replicated and row-sharded post-pass
    d = [a.astype(np.float64) for a in (x, y, t, p)]
    prm = np.array([30.0, -20.0])
    robj = R.variance_objective(); robj.sensor_size = (H, W); robj.accum = "f64"
    fr = float(robj.evaluate_function(prm, *d, R.linvel_warp(), (H, W), 1.0))
    gr = np.asarray(robj.evaluate_gradient(prm, *d, R.linvel_warp(), (H, W), 1.0), np.float64)
    assert np.abs(gr).max() > 1e-3                        # (a scene with a real gradient)
    ev = DeviceEvents.from_arrays(x[lo:hi], y[lo:hi], t[lo:hi], p[lo:hi])
    t_ref = DD.global_time_range(float(t[lo]), float(t[hi - 1]))[1]
    assert t_ref == float(t[-1])
    w = E.linvel_warp()
    for post in ("replicated", "rows"):
        os.environ["EVK_SHARDED_POST"] = post
        obj = DD.shard_objective(E.variance_objective(), t_ref)
        obj.sensor_size = (H, W)
        f = float(obj.evaluate_function(prm, ev, None, None, None, w, (H, W), 1.0))
        g = np.asarray(obj.evaluate_gradient(prm, ev, None, None, None, w, (H, W), 1.0), np.float64)
        assert abs(f - fr) <= 1e-5 * abs(fr), (post, f, fr)
        assert np.abs(g - gr).max() <= 1e-5 * np.abs(gr).max(), (post, g, gr)
        fg = obj.evaluate_function_and_gradient(prm, ev, None, None, None, w, (H, W), 1.0)
        assert abs(float(fg[0]) - f) <= 1e-6 * abs(f)
    os.environ.pop("EVK_SHARDED_POST")
    # the whole optimisation, in lock-step on both ranks (replicated scalars -> identical BFGS trajectories)
    obj = DD.shard_objective(E.variance_objective(), t_ref)
    obj.sensor_size = (H, W)
    obj.reference_exact = False           # the consistent gradient (upstream's is not the gradient of its function: Q5)
    arg = E.optimize_contrast(ev, None, None, None, w, obj, numeric_grads=False, blur_sigma=1.0, img_size=(H, W))
    both = [None, None]
    dist.all_gather_object(both, [float(a) for a in arg])
    assert both[0] == both[1], both
    assert abs(arg[0] - 40.0) < 3.0 and abs(arg[1] + 25.0) < 3.0, arg
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_against_the_whole_stream_oracle():
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    errs = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, errs)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(600)
    hung = [pr for pr in procs if pr.is_alive()]
    for pr in hung:
        pr.kill()
    msgs = []
    while not errs.empty():
        msgs.append("rank %d:\n%s" % errs.get())
    assert not hung and all(pr.exitcode == 0 for pr in procs), "\n".join(msgs) or "a rank hung or died"
