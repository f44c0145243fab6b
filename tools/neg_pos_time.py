import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from event_utils_amd.representations import voxel_grid as V
from event_utils_amd import tiled
H, W, B, n = 480, 640, 5, 10_000_000
rng = np.random.default_rng(1)
cols = [torch.from_numpy(a).cuda() for a in (rng.integers(0, W, n).astype(np.float32), rng.integers(0, H, n).astype(np.float32),
        np.sort(rng.uniform(0, 0.1, n)).astype(np.float32), (rng.integers(0, 2, n) * 2 - 1).astype(np.float32))]
def two_calls():
    pw = torch.where(cols[3] > 0, 1.0, 0.0).to(torch.float32); nw = torch.where(cols[3] <= 0, 1.0, 0.0).to(torch.float32)
    return V.events_to_voxel_torch(cols[0], cols[1], cols[2], pw, B, sensor_size=(H, W)), V.events_to_voxel_torch(cols[0], cols[1], cols[2], nw, B, sensor_size=(H, W))
for name, fn in (("one pass (EVK_VOXEL_SPLIT_POLARITY)", lambda: V.events_to_neg_pos_voxel_torch(*cols, B, sensor_size=(H, W))), ("two voxelisations", two_calls)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); print("%-38s %.3f ms per call" % (name, (time.perf_counter() - t0) / 20 * 1e3))
