"""Host (Python) time of one voxel call vs its device time: is the bench loop host-bound?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from event_utils_amd.representations.voxel_grid import _voxel_f32_device
H, W, B, n = 480, 640, 5, 10_000_000
rng = np.random.default_rng(1)
cols = [torch.from_numpy(a).cuda() for a in (rng.integers(0, W, n).astype(np.float32), rng.integers(0, H, n).astype(np.float32),
        np.sort(rng.uniform(0, 0.1, n)).astype(np.float32), (rng.integers(0, 2, n) * 2 - 1).astype(np.float32))]
out = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
call = lambda: _voxel_f32_device(*cols, B, (H, W), 0.0, 0.1, out=out, check=False, impl="tiled", fresh=True)
for _ in range(5):
    call()
torch.cuda.synchronize()
# host time: enqueue K calls without waiting (the queue is deep enough not to block for K = 50)
K = 50
t0 = time.perf_counter()
for _ in range(K):
    call()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.1f us/call; wall incl. drain %.1f us/call" % ((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    call()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
