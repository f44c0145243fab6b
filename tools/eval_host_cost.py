"""Host side of one objective evaluation at configs[2] (10 M events, 640x480): cProfile of evaluate_function / evaluate_gradient."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import event_utils_amd as E
from event_utils_amd.events import DeviceEvents
torch.cuda.set_device(0)
n, H, W = 10_000_000, 480, 640
rng = np.random.default_rng(2)
x = rng.uniform(1, W - 1, n).astype(np.float32); y = rng.uniform(1, H - 1, n).astype(np.float32)
t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32); p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
ev = DeviceEvents.from_arrays(x, y, t, p, precision="f32")
obj, w = E.variance_objective(), E.linvel_warp()
obj.sensor_size = (H, W)
prm = np.array([30.0, -20.0])
for _ in range(20): obj.evaluate_function(prm, ev, None, None, None, w, (H, W), 1.0)
t0 = time.perf_counter()
for _ in range(200): obj.evaluate_function(prm, ev, None, None, None, w, (H, W), 1.0)
print("evaluate_function %.1f us per call" % ((time.perf_counter() - t0) / 200 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(300): obj.evaluate_function(prm, ev, None, None, None, w, (H, W), 1.0)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
