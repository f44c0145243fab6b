"""The 50 M-event / 720p voxel call in a loop long enough for steady clocks (300 calls), for rocprofv3 --kernel-trace --stats:
the in-situ durations of the two kernels (bash tools/kstats.sh c5 python tools/c5_loop.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tile_attrib as T  # noqa: E402
from event_utils_amd import tiled  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
spec = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "720x1280x50000000").split("x")]
H, W, n = spec[:3]
nsets = 4 if n * 16 <= (256 << 20) else 1
sets = [T.stream(100 + k, n, H, W, dev) for k in range(nsets)]
out = torch.empty((5, H, W), dtype=torch.float32, device=dev)
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 300):
    tiled.voxel_f32(*sets[i % nsets], 0.0, 0.1, 5, H, W, out, None, impl="tiled", fresh=True)
torch.cuda.synchronize()
print("done")
