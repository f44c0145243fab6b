"""
TEST INFRASTRUCTURE ONLY.  Generates tests/golden/f15_native_dtypes.npz: events in the reference's ON-DISK dtypes
(lib/data_formats/event_packagers.py:90-93: xs, ys int16, ts float64, ps bool; h5_to_memmap.py:119-121: xy int16
(N, 2), t float64, p uint8) with epoch-scale timestamps, and what the REAL reference computes from them once they are
widened the way its loaders do (memmap_dataset.py:19-24 / hdf5_dataset.py:18-23: float32 coordinates, p * 2.0 - 1.0;
base_dataset.py:306: (ts - ts_0) in float64, then .float()).
Run in the build container only:   python -m oracle.make_golden_native
"""
import numpy as np
import torch

from . import ref_loader
from .make_golden import save


def main():
    torch.set_num_threads(1)
    ref = ref_loader.load()
    V, O, Wp = ref.voxel_grid, ref.objectives, ref.warps
    H, W, n, B = 180, 240, 20000, 5
    rng = np.random.default_rng(150)
    xs = rng.integers(0, W, n).astype(np.int16)
    ys = rng.integers(0, H, n).astype(np.int16)
    ts = 1.6e9 + np.sort(rng.uniform(0.0, 0.5, n))             # float64 epoch seconds: float32 would resolve 128 s
    ps = rng.integers(0, 2, n).astype(bool)
    out = dict(xs=xs, ys=ys, ts=ts, ps=ps, sensor_size=np.array([H, W]), B=np.int64(B))

    # numpy path on the stored integer coordinates / float64 timestamps, polarity as the loaders produce it
    out["voxel_numpy_f64"] = V.events_to_voxel(xs, ys, ts, ps * 2.0 - 1.0, B, sensor_size=(H, W))
    # torch path on the loaders' widened float32 event tensor (base_dataset.py:306)
    ev = torch.from_numpy(np.stack((xs.astype(np.float32), ys.astype(np.float32), ts - ts[0], ps * 2.0 - 1.0), axis=1)).float()
    out["voxel_torch_widened"] = V.events_to_voxel_torch(ev[:, 0], ev[:, 1], ev[:, 2], ev[:, 3], B,
                                                         sensor_size=(H, W)).numpy()
    # torch path fed the narrow dtypes directly (int16 coordinates, float32 pre-offset t, uint8 {0,1} used literally)
    out["voxel_torch_narrow_literal"] = V.events_to_voxel_torch(
        torch.from_numpy(xs), torch.from_numpy(ys), ev[:, 2].contiguous(), torch.from_numpy(ps.astype(np.uint8)), B,
        sensor_size=(H, W)).numpy()
    # contrast maximisation on the widened float64 arrays the demo would pass (events_cmax.py:391-397: ts - ts[0])
    w, obj = Wp.linvel_warp(), O.variance_objective()
    xf, yf, tf, pf = xs.astype(np.float64), ys.astype(np.float64), ts - ts[0], ps * 2.0 - 1.0
    prm = np.array([[0., 0.], [30., -20.], [-120., 75.]])
    out["cmax_params"] = prm
    out["cmax_f"] = np.array([np.float64(obj.evaluate_function(q, xf, yf, tf, pf, w, (H, W), blur_sigma=1.0)) for q in prm])
    out["cmax_g"] = np.array([np.asarray(obj.evaluate_gradient(q, xf, yf, tf, pf, w, (H, W), blur_sigma=1.0),
                                         dtype=np.float64) for q in prm])
    save("f15_native_dtypes", **out)


if __name__ == "__main__":
    main()
