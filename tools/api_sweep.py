"""Every public entry point once, at the reference's own sizes (a DAVIS 240x180 window of 50 k events, host arrays in the
reference's dtypes) and at 1 M events: wall time per call -- a smoke screen for host-side outliers (a call that takes
milliseconds where its neighbours take a hundred microseconds).      python tools/api_sweep.py"""
import os
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import event_utils_amd as E  # noqa: E402
from event_utils_amd.contrast_max import events_cmax as C, objectives as O  # noqa: E402
from event_utils_amd.representations import image as I, voxel_grid as V  # noqa: E402

warnings.simplefilter("ignore")


def med(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


for n, H, W in ((50_000, 180, 240), (1_000_000, 180, 240)):
    x, y, t, p = bench.structured_scene(3, n, H, W)
    xi, yi = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)
    x64, y64, p64 = xi.astype(np.float64), yi.astype(np.float64), p.astype(np.float64)
    t64 = 1_600_000_000.0 + np.round(t.astype(np.float64) * 1e6) / 1e6
    tx, ty, tp = (torch.from_numpy(a.astype(np.float32)) for a in (x64, y64, p64))
    tt = torch.from_numpy((t64 - t64[0]).astype(np.float32))
    w = E.linvel_warp()
    q = np.array([38.0, -24.0])

    def obj(cls=O.variance_objective, **kw):
        o = cls(**kw)
        o.sensor_size = (H, W)
        return o
    rows = [
        ("events_to_image(int arrays)", lambda: I.events_to_image(xi, yi, p.astype(np.int64), sensor_size=(H, W))),
        ("events_to_image(bilinear)", lambda: I.events_to_image(x64, y64, p64, sensor_size=(H, W), interpolation="bilinear")),
        ("events_to_image_torch(cpu tensors)", lambda: I.events_to_image_torch(tx, ty, tp, sensor_size=(H, W))),
        ("events_to_image_torch(bilinear)", lambda: I.events_to_image_torch(tx, ty, tp, sensor_size=(H, W), interpolation="bilinear")),
        ("events_to_timestamp_image", lambda: I.events_to_timestamp_image(x64, y64, t64 - t64[0], p64, sensor_size=(H, W))),
        ("events_to_timestamp_image_torch", lambda: I.events_to_timestamp_image_torch(tx, ty, tt, tp, sensor_size=(H, W))),
        ("events_to_voxel(numpy)", lambda: V.events_to_voxel(xi, yi, t64, p64, 5, sensor_size=(H, W))),
        ("events_to_voxel_torch", lambda: V.events_to_voxel_torch(tx, ty, tt, tp, 5, sensor_size=(H, W))),
        ("events_to_neg_pos_voxel_torch", lambda: V.events_to_neg_pos_voxel_torch(tx, ty, tt, tp, 5, sensor_size=(H, W))),
        ("events_to_neg_pos_voxel(numpy)", lambda: V.events_to_neg_pos_voxel(xi, yi, t64, p64, 5, sensor_size=(H, W))),
        ("voxel_grids_fixed_n_torch(n/10)", lambda: V.voxel_grids_fixed_n_torch(tx, ty, tt, tp, 5, n // 10, sensor_size=(H, W))),
        ("voxel_grids_fixed_t_torch(T/10)", lambda: V.voxel_grids_fixed_t_torch(tx, ty, tt, tp, 5, float(tt[-1]) / 10, sensor_size=(H, W))),
        ("linvel_warp.warp(numpy)", lambda: w.warp(x64, y64, t64, p64, t64[-1], q, compute_grad=True)),
        ("events_bounds_mask", lambda: E.events_bounds_mask(x64, y64, 0, W, 0, H)),
        ("get_iwe(numpy, grad)", lambda: O.get_iwe(q, x64, y64, t64, p64, w, (H, W), compute_gradient=True, sensor_size=(H, W))),
        ("variance.evaluate_function(numpy)", lambda: obj().evaluate_function(q, x64, y64, t64, p64, w, (H, W), 1.0)),
        ("variance.evaluate_gradient(numpy)", lambda: obj().evaluate_gradient(q, x64, y64, t64, p64, w, (H, W), 1.0)),
        ("sos.evaluate_function(numpy)", lambda: obj(O.sos_objective).evaluate_function(q, x64, y64, t64, p64, w, (H, W), 1.0)),
        ("r1.evaluate_function(numpy)", lambda: obj(O.r1_objective).evaluate_function(q, x64, y64, t64, p64, w, (H, W), 1.0)),
        ("optimize(numeric, scipy default)", lambda: C.optimize(x64, y64, t64, p64, w, obj(), numeric_grads=True, img_size=(H, W))),
        ("optimize(analytic)", lambda: C.optimize(x64, y64, t64, p64, w, obj(), numeric_grads=False, img_size=(H, W))),
        ("optimize_contrast(evk_bfgs)", lambda: C.optimize_contrast(x64, y64, t64, p64, w, obj(), optimizer="evk_bfgs", numeric_grads=False, blur_sigma=1.0, img_size=(H, W))),
        ("optimize(adaptive lifespan)", lambda: C.optimize(x64, y64, t64, p64, w, obj(adaptive_lifespan=True, minimum_events=1000), numeric_grads=False, img_size=(H, W))),
        ("optimize(sos objective)", lambda: C.optimize(x64, y64, t64, p64, w, obj(O.sos_objective), numeric_grads=True, img_size=(H, W))),
        ("optimize_r2", lambda: C.optimize_r2(x64, y64, t64, p64, w, obj(), numeric_grads=False)),
        ("grid_search_optimisation", lambda: C.grid_search_optimisation(x64, y64, t64, p64, w, obj(), (H, W), param_ranges=[[-100, 100], [-100, 100]], log_scale=False)),
        ("objective_landscape(20x20)", lambda: C.objective_landscape(x64, y64, t64, p64, obj(), w, img_size=(H, W))),
    ]
    for name, fn in rows:
        try:
            ms = med(fn, 3)
            print("n=%-8d %-40s %9.3f ms" % (n, name, ms), flush=True)
        except Exception as e:  # noqa: BLE001
            print("n=%-8d %-40s FAILED %r" % (n, name, e), flush=True)
