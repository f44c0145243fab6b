"""Per-phase shader-clock sums of the voxel tile kernel (a -DV2_PHASE_TIMING build: EVK_LIB_PATH=tools/exp/libevk_phase.so).
Every wave adds the cycles it spends between two V2_U() marks to v2_tile_cycles[]; evk_debug_tile_cycles reads and clears them.
The timer reads drain the wave's LDS queue, so the SHARES matter, not the sum.
    python tools/tile_phases.py [HxWxN[xREC]]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from event_utils_amd import _lib, tiled  # noqa: E402
import tile_attrib as T  # noqa: E402

NAMES = ["0 plan / setup", "1 zero accumulators", "2 table entries + scan", "3 (barrier)", "4 list build", "5 (barrier)",
         "6 chunk rounds", "7 long segments / batch tail", "8 final barrier", "9 flush", "10", "11"]

if __name__ == "__main__":
    torch.cuda.set_device(0)
    spec = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "720x1280x50000000x4").split("x")]
    H, W, n = spec[:3]
    tiled.FORCE["rec"] = spec[3] if len(spec) > 3 else None
    dev = torch.device("cuda", 0)
    cols = T.stream(100, n, H, W, dev)
    out = torch.zeros((5, H, W), dtype=torch.float32, device=dev)
    shape2 = tiled.voxel2_shape(H, W, 5)
    run = lambda stage: tiled.voxel2(cols, None, n, 0.0, 0.1, 5, H, W, *shape2, out, None, True, stage=stage)  # noqa: E731
    run(0)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    L = _lib.lib()
    L.evk_debug_tile_cycles(buf)
    reps = 20
    for _ in range(reps):
        run(_lib.EVK_VOXEL2_TILES_ONLY)
    torch.cuda.synchronize()
    L.evk_debug_tile_cycles(buf)
    c = np.array(list(buf)[:12], dtype=np.float64) / reps
    tot = c.sum()
    print("tile kernel phases, %dx%d n=%d rec=%s tiles %s: total %.3g wave-cycles per launch" % (W, H, n, tiled.FORCE["rec"], shape2, tot))
    for name, v in zip(NAMES, c):
        if v:
            print("  %-32s %6.2f %%" % (name, 100.0 * v / tot))
