"""End-to-end time of events_to_voxel_torch when the caller hands HOST tensors (the reference's usual situation):
where does it go -- upload, kernels, download?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import event_utils_amd as E
H, W, B, n = 480, 640, 5, 10_000_000
rng = np.random.default_rng(1)
cols = [torch.from_numpy(a) for a in (rng.integers(0, W, n).astype(np.float32), rng.integers(0, H, n).astype(np.float32),
        np.sort(rng.uniform(0, 0.1, n)).astype(np.float32), (rng.integers(0, 2, n) * 2 - 1).astype(np.float32))]
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("events_to_voxel_torch(host tensors) end to end: %.2f ms" % timed(lambda: E.events_to_voxel_torch(*cols, B, sensor_size=(H, W))))
print("  pageable .to(cuda) of the 4 columns:          %.2f ms (%.1f GB/s)" % ((lambda ms: (ms, 0.16 / ms * 1e3))(timed(lambda: [c.to("cuda") for c in cols]))))
pinned = [c.pin_memory() for c in cols]
print("  pinned   .to(cuda) of the 4 columns:          %.2f ms (%.1f GB/s)" % ((lambda ms: (ms, 0.16 / ms * 1e3))(timed(lambda: [c.to("cuda", non_blocking=True) for c in pinned]))))
dcols = [c.cuda() for c in cols]
print("  device tensors:                               %.2f ms" % timed(lambda: E.events_to_voxel_torch(*dcols, B, sensor_size=(H, W))))
out = E.events_to_voxel_torch(*dcols, B, sensor_size=(H, W))
print("  grid .cpu():                                  %.2f ms" % timed(lambda: out.cpu()))
stage = torch.empty(n, dtype=torch.float32).pin_memory()
def staged():
    for c in cols:
        stage.copy_(c); stage.to("cuda", non_blocking=True); torch.cuda.current_stream().synchronize()
print("  pageable -> pinned staging -> cuda:           %.2f ms" % timed(staged))
