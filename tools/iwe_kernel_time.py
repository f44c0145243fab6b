"""Device time of the tiled IWE (k_iwe_tiled + gather) alone, f and grad, full and compact records, on sensor-pixel
events: C3 (10 M, 640x480) and, with --big, C4 (50 M, 1280x720).  EVK_LIB_PATH selects an ablation build."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from event_utils_amd import tiled, _lib  # noqa: E402
from event_utils_amd.events import DeviceEvents  # noqa: E402


def run(x, y, t, p, H, W, reps):
    ch, cw = H + 1, W + 1
    for mode in ("full", "compact"):
        tiled.FORCE["iwe_records"] = mode
        ev = DeviceEvents.from_arrays(x, y, t, p, precision="f32")
        buf = torch.zeros(3 * ch * cw, dtype=torch.float32, device="cuda")
        for name, flags in (("f", 0), ("grad", _lib.EVK_IWE_GRADIENT)):
            go = lambda: tiled.iwe_linvel(ev, 0.1, 30.0, -20.0, float(W), float(H), ch, cw, flags, buf, buf[ch * cw:])
            for _ in range(3):
                go()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                go()
            b.record()
            torch.cuda.synchronize()
            print("%s %dx%d n=%d %-7s %-4s %.4f ms" % (os.environ.get("EVK_LIB_PATH", "lib")[-14:], W, H, len(x), mode, name,
                                                    a.elapsed_time(b) / reps), flush=True)
        del ev
        torch.cuda.empty_cache()


if __name__ == "__main__":
    torch.cuda.set_device(0)
    if "--big" not in sys.argv:
        x, y, t, p = bench.synth(2, 10_000_000, 0.0, 0.1, real_xy=False)
        run(x, y, t, p, 480, 640, 20)
    else:
        x, y, t, p = bench.structured_scene(3, 50_000_000, 720, 1280)
        run(np.floor(x), np.floor(y), t, p, 720, 1280, 10)
