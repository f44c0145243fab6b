"""A few EVK_VOXEL2_LIVE calls for rocprofv3 --kernel-trace (timeline of partition / consumer / tile kernel), and how many
tiles the consumers finished (status words of the last call).
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/live_trace -- python tools/live_trace.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_utils_amd import tiled  # noqa: E402
from event_utils_amd.representations.voxel_grid import _voxel_f32_device  # noqa: E402
from tools.voxel_sweep import synth  # noqa: E402

H, W, B = 480, 640, 5
n = int(os.environ.get("N", 10_000_000))
torch.cuda.set_device(0)
sets = []
for k in range(4):
    x, y, t, p = synth(1000 * k + 1, n, H, W)
    sets.append([torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (x, y, t, p)])
out = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
for live in (False, True):
    tiled.FORCE["live"] = live
    for i in range(6):
        _voxel_f32_device(*sets[i % 4], B, (H, W), None, None, out=out, check=False, impl="tiled", fresh=True)
    torch.cuda.synchronize()
idx = [v for k, v in tiled._zpersist.items() if k[0] == "voxel2_index"][0].cpu().numpy().astype(np.uint32)
V2_HDR, MAXT = 16, 2048
prog = idx[V2_HDR + MAXT: V2_HDR + MAXT + 256]
status = idx[V2_HDR + MAXT + 256: V2_HDR + MAXT + 256 + 512]
print("progress words: epoch %s runs %s" % (np.unique(prog >> 8), np.unique(prog & 255)))
print("status: epochs %s, states (1 flushing, 2 done, 3 left) %s" % (np.unique(status >> 2), dict(zip(*np.unique(status & 3, return_counts=True)))))
