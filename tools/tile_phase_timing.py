"""Where a tile workgroup spends its time: per-phase shader cycles of k_voxel_tiles2 from an experiments build with
-DV2_PHASE_TIMING (tools/ab_build.sh "phase:-DEVK_EXPERIMENTS -DV2_PHASE_TIMING")."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EVK_LIB_PATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp", "libevk_phase.so"))
from event_utils_amd import _lib, tiled  # noqa: E402
from event_utils_amd.representations.voxel_grid import _voxel_f32_device  # noqa: E402

torch.cuda.set_device(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "uniform"       # uniform | edges | blob (tools/voxel_sweep.py)
H, W, B = (480, 640, 5) if n <= 20_000_000 else (720, 1280, 5)
if kind == "uniform":
    rng = np.random.default_rng(1)
    x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32); p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
else:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import voxel_sweep
    x, y, t, p = [np.ascontiguousarray(a) for a in voxel_sweep.scene(kind, n, H, W)]
cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
out = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
run = lambda: _voxel_f32_device(*cols, B, (H, W), float(t[0]), float(t[-1]), out=out, check=False, impl="tiled", fresh=True)
for _ in range(5):
    run()
torch.cuda.synchronize()
L = _lib.lib()
buf = (ctypes.c_ulonglong * 16)()
L.evk_debug_tile_cycles(buf)
reps = 10
for _ in range(reps):
    run()
torch.cuda.synchronize()
L.evk_debug_tile_cycles(buf)
names = ["plan loads", "zero accumulators", "table entry wait + scan", "barrier 1", "list build", "barrier 2", "chunk rounds",
         "long segments", "final barrier", "flush"]
tw, th = tiled.voxel2_shape(H, W, B)

waves = L.evk_voxel2_num_tiles(H, W, tw, th) * 8
print("scene %s, %d tiles" % (kind, waves // 8))
tot = 0.0
for i, nm in enumerate(names):
    cyc = buf[i] / reps / waves        # average cycles per wave per call
    tot += cyc
    print("%-28s %9.0f cycles/wave  = %6.2f us at 2.4 GHz" % (nm, cyc, cyc / 2400.0))
print("%-28s %9.0f cycles/wave  = %6.2f us   (%d waves)" % ("sum", tot, tot / 2400.0, waves))
