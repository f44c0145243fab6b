"""
TEST INFRASTRUCTURE ONLY.  Generates tests/golden/f16_voxel_windows.npz by calling the REAL reference's windowed /
split voxelisation entry points (lib/representations/voxel_grid.py: voxel_grids_fixed_n_torch :37-57,
voxel_grids_fixed_t_torch :59-80, events_to_voxel_timesync_torch :82-112, events_to_neg_pos_voxel_torch :155-182,
events_to_neg_pos_voxel :219-243) on one seeded stream.   Run in the build container only:
    python -m oracle.make_golden_windows
"""
import numpy as np
import torch

from . import ref_loader
from .make_golden import save


def main():
    torch.set_num_threads(1)
    V = ref_loader.load().voxel_grid
    rng = np.random.default_rng(160)
    n, H, W, B = 12000, 40, 56, 4
    x = rng.uniform(0, W, n).astype(np.float32); y = rng.uniform(0, H, n).astype(np.float32)   # fractional: .long() truncation
    x[x >= W] = W - 1; y[y >= H] = H - 1
    t = np.sort(rng.uniform(0, 1.0, n)).astype(np.float32)
    p = rng.normal(size=n).astype(np.float32)
    p[::9] = 0.0
    tx, ty, tt, tp = (torch.from_numpy(a) for a in (x, y, t, p))
    out = dict(xs=x, ys=y, ts=t, ps=p, sensor_size=np.array([H, W]), B=np.int64(B))
    wn = V.voxel_grids_fixed_n_torch(tx, ty, tt, tp, B, 2500, sensor_size=(H, W))
    out["fixed_n"] = np.stack([w.numpy() for w in wn]); out["fixed_n_n"] = np.int64(2500)
    wn2 = V.voxel_grids_fixed_n_torch(tx, ty, tt, tp, B, 3000, sensor_size=(H, W))   # n divides len: last window dropped
    out["fixed_n_div"] = np.stack([w.numpy() for w in wn2]); out["fixed_n_div_n"] = np.int64(3000)
    wt = V.voxel_grids_fixed_t_torch(tx, ty, tt, tp, B, 0.13, sensor_size=(H, W))
    out["fixed_t"] = np.stack([w.numpy() for w in wt]); out["fixed_t_t"] = np.float64(0.13)
    out["timesync"] = V.events_to_voxel_timesync_torch(tx, ty, tt, tp, B, 0.2, 0.5, sensor_size=(H, W)).numpy()
    vp, vn = V.events_to_neg_pos_voxel_torch(tx, ty, tt, tp, B, sensor_size=(H, W))
    out["neg_pos_torch_pos"], out["neg_pos_torch_neg"] = vp.numpy(), vn.numpy()
    xi, yi = x.astype(np.int64), y.astype(np.int64)
    vp, vn = V.events_to_neg_pos_voxel(xi, yi, t.astype(np.float64), p, B, sensor_size=(H, W))
    out["neg_pos_numpy_pos"], out["neg_pos_numpy_neg"] = vp, vn
    save("f16_voxel_windows", **out)
    print("  windows: fixed_n %d (+%d), fixed_t %d" % (len(wn), len(wn2), len(wt)))


if __name__ == "__main__":
    main()
