import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import tile_attrib as T
from event_utils_amd import tiled
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
H, W, n = 720, 1280, 50_000_000
sets = [T.stream(100, n, H, W, dev)]
tiled.FORCE["rec"] = 4
for i in range(6):
    k = tiled.time_voxel_kernels(sets, 0.0, 0.1, 5, H, W, impl="tiled", reps=10 if i < 3 else 50)
    print(i, k["total_ms"], k["kernels_ms"], flush=True)
# busy the GPU for 2 s then measure again
x = torch.randn(8192, 8192, device=dev)
t0 = time.time()
while time.time() - t0 < 2.0:
    y = x @ x
torch.cuda.synchronize()
for i in range(3):
    k = tiled.time_voxel_kernels(sets, 0.0, 0.1, 5, H, W, impl="tiled", reps=20)
    print("after matmul burn", i, k["total_ms"], k["kernels_ms"], flush=True)
