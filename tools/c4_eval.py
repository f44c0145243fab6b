"""C4-shaped objective evaluations only (50M events, 1280x720) -- target for rocprofv3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import event_utils_amd as E
from event_utils_amd.events import DeviceEvents
import bench
n = int(os.environ.get("N", 50_000_000)); H, W = 720, 1280
x, y, t, p = bench.structured_scene(3, n, H, W)
ev = DeviceEvents.from_arrays(x, y, t, p, precision="f32")
obj, w = E.variance_objective(), E.linvel_warp()
obj.sensor_size = (H, W)
prm = np.array([30.0, -20.0])
for fn, name in ((obj.evaluate_function, "f"), (obj.evaluate_gradient, "grad"), (obj.evaluate_numeric_gradient, "batch3")):
    for _ in range(2): fn(prm, ev, None, None, None, w, (H, W), 1.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn(prm, ev, None, None, None, w, (H, W), 1.0)
    torch.cuda.synchronize(); print(name, "ms", (time.perf_counter() - t0) / 5 * 1e3)
