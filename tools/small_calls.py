"""The data loader's calls at its item sizes (base_dataset.py:448,451: events_to_voxel_torch / events_to_neg_pos_voxel_torch on
30 k - 100 k events, 5 bins) and the event images: WALL time per public call on device-resident tensors, back to back (at these
sizes the host's ~30 us per call is the bound, not the kernels), per EVK_IMPL.   usage: python tools/small_calls.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import event_utils_amd as E  # noqa: E402

B = 5
for H, W in ((180, 240), (480, 640)):
    for n in (30_000, 100_000, 300_000):
        rng = np.random.default_rng(n)
        xs, ys = rng.integers(0, W, n).astype(np.int16), rng.integers(0, H, n).astype(np.int16)
        ts = np.sort(rng.uniform(0, 0.05, n)); ps = rng.integers(0, 2, n).astype(np.uint8)
        f32 = [torch.from_numpy(a).cuda() for a in (xs.astype(np.float32), ys.astype(np.float32), ts.astype(np.float32), ps.astype(np.float32) * 2 - 1)]
        xr = [torch.from_numpy(rng.uniform(0, W - 1, n).astype(np.float32)).cuda(), torch.from_numpy(rng.uniform(0, H - 1, n).astype(np.float32)).cuda()]
        ev = E.DeviceEvents.from_native(xs, ys, ts, ps)
        ops = (("voxel f32", lambda: E.events_to_voxel_torch(*f32, B, sensor_size=(H, W))),
               ("voxel on-disk", lambda: E.events_to_voxel_torch(ev, None, None, None, B, sensor_size=(H, W))),
               ("neg/pos f32", lambda: E.events_to_neg_pos_voxel_torch(*f32, B, sensor_size=(H, W))),
               ("neg/pos on-disk", lambda: E.events_to_neg_pos_voxel_torch(ev, None, None, None, B, sensor_size=(H, W))),
               ("image nearest", lambda: E.events_to_image_torch(f32[0], f32[1], f32[3], sensor_size=(H, W))),
               ("image bilinear", lambda: E.events_to_image_torch(xr[0], xr[1], f32[3], sensor_size=(H, W), interpolation="bilinear")))
        for name, fn in ops:
            res = []
            for impl in ("direct", "tiled", "auto"):
                os.environ["EVK_IMPL"] = impl
                for _ in range(20):
                    fn()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(300):
                    fn()
                torch.cuda.synchronize()
                res.append((time.perf_counter() - t0) / 300 * 1e6)
            os.environ.pop("EVK_IMPL")
            print("%dx%d n=%6d  %-16s direct %6.1f us   one-pass %6.1f us   auto %6.1f us" % (H, W, n, name, *res), flush=True)
        E.check_errors()
