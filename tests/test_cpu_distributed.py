"""CPU, world_size 2, gloo: the event-sharded N>1 path -- shard bounds, agreement on the global time range, one
all-reduce of the partial grids -- with the oracle standing in for the per-shard kernels (no GPU here)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import reference_np as R


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_local(xs, ys, ts, ps, B, sensor_size, t_first, t_last):
    if len(xs) == 0:
        return torch.zeros((B,) + tuple(sensor_size))
    v = R.events_to_voxel_torch(np.asarray(xs), np.asarray(ys), np.asarray(ts), np.asarray(ps), B,
                                sensor_size=sensor_size, accum="f64", t_range=(t_first, t_last))
    return torch.from_numpy(v)


def _worker(rank, world, port, n, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from event_utils_amd import distributed as DD
    H, W, B = 24, 32, 5
    rng = np.random.default_rng(0)
    x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    lo, hi = DD.shard_bounds(n, rank, world)
    assert DD.is_distributed()
    tr = DD.global_time_range(t[lo] if hi > lo else np.inf, t[hi - 1] if hi > lo else -np.inf)
    assert tr == (float(t[0]), float(t[-1]))
    vox = DD.events_to_voxel_torch_sharded(x[lo:hi], y[lo:hi], t[lo:hi], p[lo:hi], B, (H, W), local_fn=_oracle_local)
    ref = R.events_to_voxel_torch(x, y, t, p, B, sensor_size=(H, W), accum="f64")
    err = np.abs(vox.numpy().astype(np.float64) - ref).max()
    # the same grid through the all-links form of the exchange (EVK_VOXEL_COLLECTIVE=rsag: reduce-scatter + all-gather;
    # 5 * 24 * 32 = 3840 cells divide by 2, the (7,) buffer below exercises the tail that does not)
    os.environ["EVK_VOXEL_COLLECTIVE"] = "rsag"
    assert DD.voxel_collective() == "rsag"
    vox2 = DD.events_to_voxel_torch_sharded(x[lo:hi], y[lo:hi], t[lo:hi], p[lo:hi], B, (H, W), local_fn=_oracle_local)
    err = max(err, np.abs(vox2.numpy().astype(np.float64) - ref).max())
    # ... and through the banded exchange (EVK_VOXEL_COLLECTIVE=bands3: three row bands, each all-reduced asynchronously
    # while the next one is produced, assembled at the end)
    os.environ["EVK_VOXEL_COLLECTIVE"] = "bands3"
    assert DD.voxel_bands() == 3
    vox3 = DD.events_to_voxel_torch_sharded(x[lo:hi], y[lo:hi], t[lo:hi], p[lo:hi], B, (H, W), local_fn=_oracle_local)
    os.environ.pop("EVK_VOXEL_COLLECTIVE")
    err = max(err, np.abs(vox3.numpy().astype(np.float64) - ref).max())
    odd = torch.arange(7, dtype=torch.float32) * (rank + 1)
    DD.reduce_scatter_all_gather_sum_(odd)
    assert torch.equal(odd, torch.arange(7, dtype=torch.float32) * sum(range(1, world + 1)))
    # objective: every rank warps to the GLOBAL t_ref and all-reduces IWE+dIWE (here with the oracle as local kernel)
    obj = R.variance_objective()
    iwe_full, d_full = R.get_iwe(np.array([30., -20.]), *(a.astype(np.float64) for a in (x, y, t, p)), R.linvel_warp(),
                                 (H, W), compute_gradient=True, sensor_size=(H, W), accum="f64")

    class shard_warp(R.linvel_warp):           # warp a shard to the global reference time
        def warp(self, xs, ys, ts, ps, t0, params, compute_grad=False):
            return super().warp(xs, ys, ts, ps, float(t[-1]), params, compute_grad)
    iwe, d = R.get_iwe(np.array([30., -20.]), *(a[lo:hi].astype(np.float64) for a in (x, y, t, p)), shard_warp(),
                       (H, W), compute_gradient=True, sensor_size=(H, W), accum="f64")
    buf = torch.from_numpy(np.concatenate([iwe[None], d]))
    DD.all_reduce_sum_(buf)
    err2 = np.abs(buf.numpy() - np.concatenate([iwe_full[None], d_full])).max()
    # the composition objective_function._one_call runs under shard_objective (distributed.sharded_evaluate): local IWE at
    # the global reference time -> one all-reduce -> blur + variance / gradient, here with oracle closures
    d64 = [a.astype(np.float64) for a in (x, y, t, p)]
    robj = R.variance_objective(); robj.sensor_size = (H, W); robj.accum = "f64"
    prm = np.array([30., -20.])
    f_full = float(robj.evaluate_function(prm, *d64, R.linvel_warp(), (H, W), 1.0))
    g_full = np.asarray(robj.evaluate_gradient(prm, *d64, R.linvel_warp(), (H, W), 1.0), dtype=np.float64)

    def local_iwe():
        if hi == lo:
            return torch.zeros((3, H + 1, W + 1), dtype=torch.float64)
        i2, d2 = R.get_iwe(prm, *(a[lo:hi] for a in d64), shard_warp(), (H, W), compute_gradient=True, sensor_size=(H, W),
                           accum="f64")
        return torch.from_numpy(np.concatenate([i2[None], d2]).astype(np.float64))

    def finish(img):
        a = img.numpy()
        return (float(robj.evaluate_function(iwe=a[0].astype(np.float32), blur_sigma=1.0)),
                np.asarray(robj.evaluate_gradient(iwe=a[0].astype(np.float32), d_iwe=a[1:].astype(np.float32), blur_sigma=1.0),
                           dtype=np.float64))
    f_sh, g_sh = DD.sharded_evaluate(local_iwe, finish)
    err3 = max(abs(f_sh - f_full) / abs(f_full), np.abs(g_sh - g_full).max() / np.abs(g_full).max())
    # the same evaluation with the ROW-SHARDED post-pass (distributed.sharded_evaluate_rows): all-to-all of row blocks with
    # halo -> every rank blurs its block (scipy on the block: 'reflect' at the block's own edges, sums over its own rows) ->
    # 8-double all-reduce -> finalise_sums.  Un-mixed blur (reference_exact=False): the oracle's consistent gradient.
    from scipy.ndimage import gaussian_filter

    def rows_post(block, y_lo, y_hi):
        b = block.numpy().astype(np.float32)
        v = gaussian_filter(b[0], 1.0)[y_lo:y_hi].astype(np.float64)
        d0 = gaussian_filter(b[1], 1.0)[y_lo:y_hi].astype(np.float64)
        d1 = gaussian_filter(b[2], 1.0)[y_lo:y_hi].astype(np.float64)
        return torch.tensor([v.sum(), d0.sum(), d1.sum(), (v * d0).sum(), (v * d1).sum(), v.sum(), (v * v).sum(), 0.0],
                            dtype=torch.float64)
    res = DD.sharded_evaluate_rows(lambda: local_iwe().to(torch.float32), rows_post, 4, 3)
    full = np.concatenate([iwe_full[None], d_full]).astype(np.float32)
    vf = gaussian_filter(full[0], 1.0).astype(np.float64)
    gf = [gaussian_filter(full[1 + k], 1.0).astype(np.float64) for k in range(2)]
    npx = vf.size
    want = np.array([2.0 / npx * ((vf * gf[0]).sum() - vf.mean() * gf[0].sum()),
                     2.0 / npx * ((vf * gf[1]).sum() - vf.mean() * gf[1].sum()), vf.mean(), vf.var()])
    err3 = max(err3, float(np.abs(res - want).max() / np.abs(want).max()))
    assert DD.row_block(H + 1, 4, 0, world)[:2] == (0, (H + 1) // 2) and DD.row_block(H + 1, 4, world - 1, world)[1] == H + 1
    both = torch.tensor([f_sh, -f_sh], dtype=torch.float64)           # identical scalars on every rank
    dist.all_reduce(both, op=dist.ReduceOp.MAX)
    same = float(both[0]) == f_sh and float(both[1]) == -f_sh
    # sharded integer event image: int32 all-reduce, bit-exact
    xi, yi, pi = x.astype(np.int64), y.astype(np.int64), p.astype(np.int64)

    def local_img(xs, ys, ps, shape):
        c = np.bincount(np.ravel_multi_index((ys, xs), shape), weights=ps, minlength=shape[0] * shape[1]) if len(xs) else \
            np.zeros(shape[0] * shape[1])
        return torch.from_numpy(c.reshape(shape).astype(np.int32))
    img = DD.events_to_image_sharded(xi[lo:hi], yi[lo:hi], pi[lo:hi], (H, W), local_fn=local_img)
    exact = bool(np.array_equal(img, R.events_to_image(xi, yi, pi, sensor_size=(H, W))))
    np.save(os.path.join(out_dir, "err%d.npy" % rank), np.array([err, np.abs(ref).max(), err2, np.abs(d_full).max(), err3,
                                                               float(same), float(exact)]))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1001, 3])
def test_event_sharded_voxel_and_iwe_world2_gloo(tmp_path, n):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        err, scale, err2, scale2, err3, same, exact = np.load(tmp_path / ("err%d.npy" % r))
        assert err <= 1e-5 * scale and err2 <= 1e-5 * scale2
        assert err3 <= 1e-5 and same == 1.0 and exact == 1.0


def test_shard_bounds_partition():
    from event_utils_amd.distributed import shard_bounds
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_is_a_noop():
    from event_utils_amd import distributed as DD
    assert not DD.is_distributed()
    t = torch.ones(3)
    assert DD.all_reduce_sum_(t) is t and DD.global_time_range(0.1, 0.9) == (0.1, 0.9)
