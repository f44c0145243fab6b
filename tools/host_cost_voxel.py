import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import event_utils_amd as E
from event_utils_amd.representations.voxel_grid import _voxel_f32_device
torch.cuda.set_device(0)
n, H, W, B = 10_000_000, 480, 640, 5
rng = np.random.default_rng(1)
x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32); p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
c = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
small = [a[:1000].contiguous() for a in c]
for _ in range(50): E.events_to_voxel_torch(*c, B, sensor_size=(H, W))
torch.cuda.synchronize()
# host cost: time to ENQUEUE k calls (no sync inside), GPU far behind
for label, fn in (("public", lambda: E.events_to_voxel_torch(*c, B, sensor_size=(H, W))),):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(label, "enqueue %.1f us/call, total %.1f us/call" % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(300): E.events_to_voxel_torch(*c, B, sensor_size=(H, W))
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
