"""Every public entry point that accepts DEVICE tensors / resident events, at 10 M events on the 640x480 sensor (and 1 M on
240x180): wall time per call and events per second -- a screen for functions still far below their neighbours (a global-atomic
kernel where the others run on LDS tiles).      python tools/device_sweep.py"""
import os
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import event_utils_amd as E  # noqa: E402
from event_utils_amd.contrast_max import objectives as O  # noqa: E402
from event_utils_amd.representations import image as I, voxel_grid as V  # noqa: E402
from event_utils_amd.transforms import optic_flow as F  # noqa: E402

warnings.simplefilter("ignore")


def med(fn, reps=7):
    fn(); fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        del r
    return float(np.median(ts)) * 1e3


dev = torch.device("cuda:0")
for n, H, W in ((1_000_000, 180, 240), (10_000_000, 480, 640)):
    x, y, t, p = bench.structured_scene(3, n, H, W)
    x = np.clip(x, 0.5, W - 1.5).astype(np.float32); y = np.clip(y, 0.5, H - 1.5).astype(np.float32)
    xd, yd, td, pd = (torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev) for a in (x, y, t, p))
    xi, yi = xd.floor(), yd.floor()
    ev = E.DeviceEvents(xd, yd, td, pd)
    w = E.linvel_warp()
    q = np.array([38.0, -24.0])
    flow = torch.randn(2, H, W, device=dev)
    pxs, pys = xd.floor().long(), yd.floor().long()
    dxs, dys = xd - xd.floor(), yd - yd.floor()
    img = torch.zeros(H + 1, W + 1, device=dev)
    dimg = torch.zeros(2, H + 1, W + 1, device=dev)
    w12 = torch.randn(2, n, device=dev)
    gimg = np.random.default_rng(0).normal(size=(H, W)).astype(np.float32)

    def obj(cls, **kw):
        o = cls(**kw)
        o.sensor_size = (H, W)
        return o
    objs = {c.__name__: obj(c) for c in (O.variance_objective, O.sos_objective, O.soe_objective, O.moa_objective, O.isoa_objective,
                                         O.sosa_objective, O.r1_objective, O.rms_objective)}
    tsi, evi = I.TimestampImage((H, W)), I.EventImage((H, W))
    rows = [
        ("events_to_image_torch nearest", lambda: I.events_to_image_torch(xi, yi, pd, sensor_size=(H, W), padding=False)),
        ("events_to_image_torch bilinear", lambda: I.events_to_image_torch(xd, yd, pd, sensor_size=(H, W), interpolation="bilinear")),
        ("events_to_timestamp_image_torch", lambda: I.events_to_timestamp_image_torch(xd, yd, td, pd, sensor_size=(H, W))),
        ("events_to_voxel_torch B=5", lambda: V.events_to_voxel_torch(xi, yi, td, pd, 5, sensor_size=(H, W))),
        ("events_to_neg_pos_voxel_torch B=5", lambda: V.events_to_neg_pos_voxel_torch(xi, yi, td, pd, 5, sensor_size=(H, W))),
        ("voxel_grids_fixed_n_torch 10 windows", lambda: V.voxel_grids_fixed_n_torch(xi, yi, td, pd, 5, n // 10, sensor_size=(H, W))),
        ("voxel_grids_fixed_n_torch 100 windows", lambda: V.voxel_grids_fixed_n_torch(xi, yi, td, pd, 5, n // 100, sensor_size=(H, W))),
        ("voxel_grids_fixed_t_torch 20 windows", lambda: V.voxel_grids_fixed_t_torch(xi, yi, td, pd, 5, float(t[-1] - t[0]) / 20, sensor_size=(H, W))),
        ("events_to_voxel_timesync_torch", lambda: V.events_to_voxel_timesync_torch(xi, yi, td, pd, 5, float(t[n // 4]), float(t[3 * n // 4]), sensor_size=(H, W))),
        ("warp_events_flow_torch", lambda: F.warp_events_flow_torch(xd, yd, td, pd, flow)),
        ("interpolate_to_image", lambda: I.interpolate_to_image(pxs, pys, dxs, dys, pd, img)),
        ("interpolate_to_derivative_img", lambda: I.interpolate_to_derivative_img(pxs, pys, dxs, dys, dimg, w12, w12)),
        ("linvel_warp.warp (device, grad)", lambda: w.warp(xd.double(), yd.double(), td.double(), pd, float(t[-1]), q, compute_grad=True)),
        ("get_iwe(DeviceEvents, grad)", lambda: O.iwe_device(q, ev, (H, W), compute_gradient=True, sensor_size=(H, W))),
        ("TimestampImage.add_events + get_image", lambda: (tsi.add_events(xd, yd, td, pd), tsi.get_image())),
        ("EventImage.add_events + get_image", lambda: (evi.add_events(xd, yd, td, pd), evi.get_image())),
    ]
    for name, o in objs.items():
        rows.append(("%s.evaluate_function" % name, lambda o=o: o.evaluate_function(q, ev, None, None, None, w, (H, W), 1.0)))
        if o.has_derivative:
            rows.append(("%s.evaluate_gradient" % name, lambda o=o: o.evaluate_gradient(q, ev, None, None, None, w, (H, W), 1.0)))
    for name, fn in rows:
        try:
            ms = med(fn, 5)
            print("n=%-9d %-44s %9.3f ms  %8.1f Mev/s" % (n, name, ms, n / ms / 1e3), flush=True)
        except Exception as e:  # noqa: BLE001
            print("n=%-9d %-44s FAILED %r" % (n, name, e), flush=True)
    del ev, objs, rows
    torch.cuda.empty_cache()
