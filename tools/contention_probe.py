"""How sensitive is one voxel call (10 M events, 640x480x5) to another kernel holding k CUs -- the situation of an
overlapped RCCL all-reduce?  A spin kernel of k workgroups (512 threads, 4-64 KB LDS) runs on a side stream while the
call is timed on the main stream.   python tools/contention_probe.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_utils_amd.representations.voxel_grid import _voxel_f32_device  # noqa: E402
from event_utils_amd import tiled  # noqa: E402

spin = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libspin.so"))
spin.spin_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
H, W, B, n = 480, 640, 5, 10_000_000
rng = np.random.default_rng(1)
cols = [torch.from_numpy(a).cuda() for a in (rng.integers(0, W, n).astype(np.float32), rng.integers(0, H, n).astype(np.float32),
                                            np.sort(rng.uniform(0, 0.1, n)).astype(np.float32),
                                            (rng.integers(0, 2, n) * 2 - 1).astype(np.float32))]
out = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
sink = torch.zeros(1, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()
call = lambda: _voxel_f32_device(*cols, B, (H, W), 0.0, 0.1, out=out, check=False, impl="tiled", fresh=True)
for _ in range(3):
    call()
torch.cuda.synchronize()
for share in ("0", "1"):
    tiled.FORCE["share_cu"] = share == "1"          # EVK_VOXEL2_SHARE_CU: 64 KB instead of 128 KB of LDS per partition workgroup
    for lds_kb in (4, 24, 48, 64):
        for k in (0, 8, 64):
            reps, ts = 10, []
            for _ in range(reps):
                torch.cuda.synchronize()
                if k:
                    spin.spin_launch(k, 400.0, ctypes.c_void_p(side.cuda_stream), ctypes.c_void_p(sink.data_ptr()), lds_kb * 1024)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                call()
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            print("share_cu=%s: %3d CUs held by another kernel with %2d KB of LDS each: voxel call %.3f ms (min %.3f)"
                  % (share, k, lds_kb, float(np.median(ts)), min(ts)), flush=True)
