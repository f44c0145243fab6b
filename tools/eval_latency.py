"""Where does one objective evaluation's latency go?  (10M events, 640x480)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import event_utils_amd as E
from event_utils_amd import tiled, _device as D, _lib
from event_utils_amd.events import DeviceEvents
from event_utils_amd.contrast_max.objectives import _blur_kernel
import bench
H, W, n = 480, 640, 10_000_000
x, y, t, p = bench.synth(2, n, 0.0, 0.1, real_xy=True)
ev = DeviceEvents.from_arrays(x, y, t, p, precision="f32")
obj, w = E.variance_objective(), E.linvel_warp(); obj.sensor_size = (H, W)
prm = np.array([30.0, -20.0])
for _ in range(3): obj.evaluate_function(prm, ev, None, None, None, w, (H, W), 1.0)
dev = ev.x.device; ch, cw = H + 1, W + 1
wts, radius = _blur_kernel(1.0)
buf = tiled._buf("iwe_buf", ch * cw * 4, dev); out, (scratch, nbytes) = D.out4(dev), D.reduce_scratch(dev)
def launch():
    return tiled.cmax_variance(ev, 0.1, 30.0, -20.0, float(W), float(H), ch, cw, 0, wts, radius, 0, buf, out, scratch, nbytes)
host = np.empty(4)
def launch_sync():
    return tiled.cmax_variance(ev, 0.1, 30.0, -20.0, float(W), float(H), ch, cw, 0, wts, radius, 0, buf, out, scratch, nbytes, host_out=host)
N = 200
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): launch()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("async: host %.1f us/call, incl. drain %.1f us/call" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): launch(); out.cpu()
print("launch + out.cpu(): %.1f us" % ((time.perf_counter() - t0) / N * 1e6))
t0 = time.perf_counter()
for _ in range(N): launch(); torch.cuda.synchronize()
print("launch + synchronize: %.1f us" % ((time.perf_counter() - t0) / N * 1e6))
t0 = time.perf_counter()
for _ in range(N): obj.evaluate_function(prm, ev, None, None, None, w, (H, W), 1.0)
print("evaluate_function: %.1f us" % ((time.perf_counter() - t0) / N * 1e6))
t0 = time.perf_counter()
for _ in range(N): tiled.iwe_plan(ev, 0.1, 30.0, -20.0, float(W), float(H), ch, cw, 0)
print("iwe_plan only (host): %.1f us" % ((time.perf_counter() - t0) / N * 1e6))
t0 = time.perf_counter()
for _ in range(N): launch_sync()
print("launch with host_out (self-synchronising): %.1f us" % ((time.perf_counter() - t0) / N * 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(300): obj.evaluate_function(prm, ev, None, None, None, w, (H, W), 1.0)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
