"""Public events_to_voxel_torch on device tensors, 10 M events 640x480x5, four rotating event streams: EVK_ERRORS=strict (the
default: the call waits for its partition kernel's report) against deferred, alternating; wall time per call."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import event_utils_amd as E
from tools.voxel_sweep import synth
torch.cuda.set_device(0)
n, H, W, B = 10_000_000, 480, 640, 5
sets = [[torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in synth(1000 * k + 1, n, H, W)] for k in range(4)]
keep = [None]
def run(k):
    for i in range(k):
        keep[0] = E.events_to_voxel_torch(*sets[i % 4], B, sensor_size=(H, W))
for mode in ("strict", "deferred") * 3:
    os.environ["EVK_ERRORS"] = mode
    run(300); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(200); torch.cuda.synchronize(); el = time.perf_counter() - t0
    E.check_errors()
    print("%-8s %.4f ms per call" % (mode, el / 200 * 1e3), flush=True)
