"""GPU (-m gpu): the event-sharded code paths on a real device with a 1-rank RCCL group (the box has one GPU): the
all-reduce is then the identity, but everything else -- global time range, shard objective at a global t_ref, IWE+dIWE
all-reduced in one buffer before the replicated blur / reductions -- runs exactly as with N ranks.  Shard additivity
(the property that makes N > 1 legal) is checked by emulating two ranks sequentially."""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import reference_np as R

pytestmark = pytest.mark.gpu


def f64(a):
    return np.asarray(a, dtype=np.float64)


@pytest.fixture(scope="module")
def pg():
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def _events(seed, n, H, W):
    rng = np.random.default_rng(seed)
    x = rng.uniform(1, W - 1, n).astype(np.float32); y = rng.uniform(1, H - 1, n).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    return x, y, t, p


def test_sharded_voxel_and_objective_with_rccl_group(pg):
    import event_utils_amd as E
    from event_utils_amd import distributed as DD
    from event_utils_amd.events import DeviceEvents
    H, W, B, n = 120, 160, 5, 400_000
    x, y, t, p = _events(0, n, H, W)
    cols = [torch.from_numpy(a).cuda() for a in (np.floor(x), np.floor(y), t, p)]
    vox = DD.events_to_voxel_torch_sharded(*cols, B, (H, W))
    ref = R.events_to_voxel_torch(np.floor(x), np.floor(y), t, p, B, sensor_size=(H, W), accum="f64")
    assert np.abs(vox.cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max()
    # objective configured for event sharding: global t_ref, all-reduce of the (3, H+1, W+1) buffer before blur
    obj = DD.shard_objective(E.variance_objective(), float(t[-1]))
    obj.sensor_size = (H, W)
    ev = DeviceEvents.from_arrays(x, y, t, p)
    w, prm = E.linvel_warp(), np.array([30., -20.])
    f = float(obj.evaluate_function(prm, ev, None, None, None, w, (H, W), 1.0))
    g = f64(obj.evaluate_gradient(prm, ev, None, None, None, w, (H, W), 1.0))
    robj = R.variance_objective(); robj.sensor_size = (H, W); robj.accum = "f64"
    d = [f64(a) for a in (x, y, t, p)]
    fr = float(robj.evaluate_function(prm, *d, R.linvel_warp(), (H, W), 1.0))
    gr = f64(robj.evaluate_gradient(prm, *d, R.linvel_warp(), (H, W), 1.0))
    assert abs(f - fr) <= 1e-5 * abs(fr) and np.abs(g - gr).max() <= 1e-5 * np.abs(gr).max() + 1e-9


def test_voxel_exchange_as_reduce_scatter_all_gather_and_errors_on_every_rank(pg, monkeypatch):
    """EVK_VOXEL_COLLECTIVE=rsag (RCCL reduce_scatter_tensor + all_gather_into_tensor on the flat grid, tail by all-reduce) gives
    the grid of the plain all-reduce; an out-of-range event -- also one that an EARLIER deferred call dropped -- raises from the
    sharded call itself, after its collectives (it must never raise between them on one rank alone)."""
    import event_utils_amd as E
    from event_utils_amd import distributed as DD
    H, W, B, n = 120, 161, 5, 400_000          # 5 * 120 * 161 cells: any world size leaves a tail or not -- both paths run
    x, y, t, p = _events(5, n, H, W)
    cols = [torch.from_numpy(a).cuda() for a in (np.floor(x), np.floor(y), t, p)]
    plain = DD.events_to_voxel_torch_sharded(*cols, B, (H, W))
    monkeypatch.setenv("EVK_VOXEL_COLLECTIVE", "rsag")
    rsag = DD.events_to_voxel_torch_sharded(*cols, B, (H, W))
    # (one rank here: the same sums; over more ranks the two forms add the partial grids in different orders)
    assert torch.allclose(plain, rsag, rtol=1e-6, atol=1e-6 * float(plain.abs().max()))
    # the banded exchange (EVK_VOXEL_COLLECTIVE=bandsK): one partition, the tile kernel in K row bands, every band
    # all-reduced while the next one accumulates -- the same grid (same kernels, same per-tile sums: bit-identical)
    for K in (2, 4, 64):
        monkeypatch.setenv("EVK_VOXEL_COLLECTIVE", "bands%d" % K)
        banded = DD.events_to_voxel_torch_sharded(*cols, B, (H, W))
        assert torch.equal(plain, banded), K
    hot = [c.clone() for c in cols]
    hot[0][: n // 2] = 40.0; hot[1][: n // 2] = 60.0               # a hot pixel: cut tiles inside a band
    monkeypatch.setenv("EVK_VOXEL_COLLECTIVE", "allreduce")
    ref_hot = DD.events_to_voxel_torch_sharded(*hot, B, (H, W))
    monkeypatch.setenv("EVK_VOXEL_COLLECTIVE", "bands3")
    assert torch.equal(ref_hot, DD.events_to_voxel_torch_sharded(*hot, B, (H, W)))
    monkeypatch.setenv("EVK_VOXEL_COLLECTIVE", "rsag")
    odd = torch.arange(1001, dtype=torch.float32, device="cuda")
    assert torch.equal(DD.reduce_scatter_all_gather_sum_(odd.clone()), odd)
    # a deferred report left behind by an earlier call on this stream is folded into the sharded call's own check
    monkeypatch.setenv("EVK_ERRORS", "deferred")
    bad = [c.clone() for c in cols]
    bad[0][7] = W + 3.0
    E.events_to_voxel_torch(*bad, B, sensor_size=(H, W))          # enqueues; the IndexError is pending
    with pytest.raises(IndexError):
        DD.events_to_voxel_torch_sharded(*cols, B, (H, W))        # clean events, but the pending report surfaces HERE
    DD.events_to_voxel_torch_sharded(*cols, B, (H, W))            # ... once
    E.check_errors()


def test_c_abi_collective_and_sharded_integer_image(pg, monkeypatch):
    """libevk's own RCCL binding (evk_comm_*, evk_allreduce_f32 / _i32: what a caller without torch uses) on a 1-rank
    communicator, the sharded integer event image and the sharded voxel grid through it."""
    import event_utils_amd as E
    from event_utils_amd import distributed as DD
    monkeypatch.setenv("EVK_COLLECTIVE", "evk")
    g = torch.arange(5000, dtype=torch.float32, device="cuda") * 0.25
    gi = torch.arange(5000, dtype=torch.int32, device="cuda") - 77
    rf, ri = g.clone(), gi.clone()
    DD.all_reduce_sum_(g, pg.group.WORLD, force=True)
    DD.all_reduce_sum_(gi, pg.group.WORLD, force=True)
    torch.cuda.synchronize()
    assert torch.equal(g, rf) and torch.equal(gi, ri)          # one rank: the sum is the identity
    assert DD._comms                                            # ... and it went through libevk's communicator
    H, W, B, n = 90, 120, 4, 200_000
    rng = np.random.default_rng(3)
    xi, yi = rng.integers(0, W, n), rng.integers(0, H, n)
    pi = rng.integers(0, 2, n) * 2 - 1
    img = DD.events_to_image_sharded(xi, yi, pi, (H, W), group=pg.group.WORLD)
    assert np.array_equal(img, R.events_to_image(xi, yi, pi, sensor_size=(H, W)))
    assert np.array_equal(img, E.events_to_image(xi, yi, pi, sensor_size=(H, W)))
    with pytest.raises(ValueError):
        DD.events_to_image_sharded(np.array([1, W + 5]), np.array([1, 1]), np.array([1, 1]), (H, W), group=pg.group.WORLD)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32)
    cols = [torch.from_numpy(a).cuda() for a in (xi.astype(np.float32), yi.astype(np.float32), t, pi.astype(np.float32))]
    vox = DD.events_to_voxel_torch_sharded(*cols, B, (H, W), group=pg.group.WORLD)
    ref = R.events_to_voxel_torch(xi.astype(np.float32), yi.astype(np.float32), t, pi.astype(np.float32), B,
                                  sensor_size=(H, W), accum="f64")
    assert np.abs(vox.cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max()
    cols[0][7] = W + 3.0
    with pytest.raises(IndexError):                             # raised on every rank, after the collectives
        DD.events_to_voxel_torch_sharded(*cols, B, (H, W), group=pg.group.WORLD)
    DD.evk_comm_destroy(pg.group.WORLD)


def test_two_emulated_ranks_are_additive():
    """rank 0 and rank 1 evaluated one after the other on the same GPU: summing their IWE / dIWE / voxel grids gives the
    single-rank result (what the all-reduce computes), with every rank warping to the GLOBAL reference time."""
    import event_utils_amd as E
    from event_utils_amd import distributed as DD
    from event_utils_amd.contrast_max.objectives import iwe_device
    from event_utils_amd.events import DeviceEvents
    from event_utils_amd.representations.voxel_grid import _voxel_f32_device
    H, W, B, n = 120, 160, 5, 500_000
    x, y, t, p = _events(1, n, H, W)
    prm = np.array([-45., 25.])
    full_ev = DeviceEvents.from_arrays(x, y, t, p)
    iwe_full, d_full = iwe_device(prm, full_ev, (H, W), True, True, (H, W))
    parts = []
    for rank in range(2):
        lo, hi = DD.shard_bounds(n, rank, 2)
        ev = DeviceEvents.from_arrays(x[lo:hi], y[lo:hi], t[lo:hi], p[lo:hi])
        parts.append(iwe_device(prm, ev, (H, W), True, True, (H, W), t_ref=float(t[-1])))
    iwe = parts[0][0] + parts[1][0]
    diwe = parts[0][1] + parts[1][1]
    assert (iwe - iwe_full).abs().max().item() <= 1e-5 * iwe_full.abs().max().item()
    assert (diwe - d_full).abs().max().item() <= 1e-5 * d_full.abs().max().item()
    cols = [torch.from_numpy(a).cuda() for a in (np.floor(x), np.floor(y), t, p)]
    full = _voxel_f32_device(*cols, B, (H, W), float(t[0]), float(t[-1]))
    acc = torch.zeros_like(full)
    for rank in range(2):
        lo, hi = DD.shard_bounds(n, rank, 2)
        acc += _voxel_f32_device(*(c[lo:hi] for c in cols), B, (H, W), float(t[0]), float(t[-1]))
    assert (acc - full).abs().max().item() <= 1e-5 * full.abs().max().item()


@pytest.mark.parametrize("sigma", [1.0, 0.0, 2.0])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_row_sharded_post_pass_equals_the_replicated_one(world, sigma):
    """evk_objective_variance_rows_f32 on the row blocks (with halo) of `world` emulated ranks, raw sums added, finalised
    on the host (distributed.finalise_sums) == the full-image fused post-pass, for the value, the reference's gradient
    (channel-mixing blur, raw IWE) and value + consistent gradient in one pass."""
    from event_utils_amd import _device as D, _lib, distributed as DD
    from event_utils_amd.contrast_max.objectives import _blur_kernel
    ch, cw = 181, 241
    rng = np.random.default_rng(5)
    img = torch.from_numpy(rng.normal(size=(3, ch, cw)).astype(np.float32)).cuda()
    w, radius = _blur_kernel(sigma)
    wp = D.host_ptr(w) if w is not None else None
    dev = img.device
    out, (scratch, nbytes) = D.out4(dev), D.reduce_scratch(dev)
    sums = torch.zeros(8, dtype=torch.float64, device=dev)
    cases = ((0, 0, "evk_objective_variance_f32"), (1, _lib.EVK_POST_MIX, "evk_objective_variance_grad_f32"),
             (3, _lib.EVK_POST_BLUR_IWE, "evk_objective_variance_fg_f32"))
    for mode, flags, fn in cases:
        if mode == 0:
            _lib.call(fn, D.ptr(img), ch, cw, wp, radius, D.ptr(out), D.ptr(scratch), nbytes, D.stream())
        else:
            _lib.call(fn, D.ptr(img), D.ptr(img[1:]), ch, cw, wp, radius, flags, D.ptr(out), D.ptr(scratch), nbytes, D.stream())
        want = out.cpu().numpy().copy()
        total = np.zeros(8)
        for rank in range(world):
            y0, y1, lo, hi = DD.row_block(ch, max(radius, 0), rank, world)
            block = img[:, lo:hi, :].contiguous()
            _lib.call("evk_objective_variance_rows_f32", D.ptr(block), mode, hi - lo, cw, y0 - lo, y1 - lo, wp, radius, flags,
                      D.ptr(sums), D.ptr(scratch), nbytes, D.stream())
            total += sums.cpu().numpy()
        got = DD.finalise_sums(total, ch * cw, mode)
        assert np.abs(got - want).max() <= 1e-11 * np.abs(want).max() + 1e-15, (mode, world, got, want)


def test_shard_objective_with_row_sharded_post(pg, monkeypatch):
    """EVK_SHARDED_POST=rows through shard_objective on the 1-rank RCCL group == the replicated post-pass."""
    import event_utils_amd as E
    from event_utils_amd import distributed as DD
    from event_utils_amd.events import DeviceEvents
    H, W, n = 120, 160, 400_000
    x, y, t, p = _events(3, n, H, W)
    ev = DeviceEvents.from_arrays(x, y, t, p)
    w, prm = E.linvel_warp(), np.array([30., -20.])
    res = {}
    for mode in ("replicated", "rows"):
        monkeypatch.setenv("EVK_SHARDED_POST", mode)
        obj = DD.shard_objective(E.variance_objective(), float(t[-1]))
        obj.sensor_size = (H, W)
        f = float(obj.evaluate_function(prm, ev, None, None, None, w, (H, W), 1.0))
        g = f64(obj.evaluate_gradient(prm, ev, None, None, None, w, (H, W), 1.0))
        fg = obj.evaluate_function_and_gradient(prm, ev, None, None, None, w, (H, W), 1.0)
        res[mode] = np.concatenate([[f], g, [float(fg[0])], f64(fg[1])])
    assert np.abs(res["rows"] - res["replicated"]).max() <= 1e-9 * np.abs(res["replicated"]).max()
