"""Index-width check of the tile-bucketed voxel path at event counts beyond 2^31 (a 288 GB MI355X holds them):
mass conservation and additivity of two half streams.   python tools/big_n_check.py [n]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_utils_amd.representations.voxel_grid import _voxel_f32_device  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_200_000_000
H, W, B = 720, 1280, 5
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.empty(n, dtype=torch.float32, device="cuda").uniform_(0, W, generator=g).floor_().clamp_(0, W - 1)
y = torch.empty(n, dtype=torch.float32, device="cuda").uniform_(0, H, generator=g).floor_().clamp_(0, H - 1)
t = torch.linspace(0.0, 0.1, n, device="cuda")                      # sorted by construction
p = torch.empty(n, dtype=torch.float32, device="cuda").uniform_(0, 1, generator=g).round_().mul_(2).sub_(1)
psum = p.double().sum().item()
torch.cuda.synchronize()
t0 = time.perf_counter()
v = _voxel_f32_device(x, y, t, p, B, (H, W), 0.0, 0.1, impl="tiled")
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("n = %d events (%.1f GB of columns): %.1f ms, %.1f Gev/s" % (n, 16 * n / 1e9, dt * 1e3, n / dt / 1e9))
print("mass: sum(grid) = %.1f, sum(p) = %.1f" % (v.double().sum().item(), psum))
assert abs(v.double().sum().item() - psum) <= 1e-3 * n ** 0.5 + 1.0
half = (n // 2) & ~3
va = _voxel_f32_device(x[:half], y[:half], t[:half], p[:half], B, (H, W), 0.0, 0.1, impl="tiled")
vb = _voxel_f32_device(x[half:], y[half:], t[half:], p[half:], B, (H, W), 0.0, 0.1, impl="tiled")
err = (va + vb - v).abs().max().item() / v.abs().max().item()
print("additivity of the two halves: rel. err %.2e" % err)
assert err <= 1e-5
print("ok")

# ---- the IWE / objective path on the same stream (real-valued coordinates inside the sensor) ----------------------
import numpy as np  # noqa: E402
import event_utils_amd as E  # noqa: E402
from event_utils_amd.contrast_max.objectives import iwe_device  # noqa: E402
del v, va, vb
x.add_(0.37).clamp_(4, W - 4)
y.add_(0.61).clamp_(4, H - 4)
ev = E.DeviceEvents(x, y, t, p)
prm = np.array([25.0, -15.0])
torch.cuda.synchronize()
t0 = time.perf_counter()
iwe, diwe = iwe_device(prm, ev, (H, W), True, True, (H, W), impl="tiled")
torch.cuda.synchronize()
print("IWE + dIWE of %d events: %.1f ms (bucketing included)" % (n, (time.perf_counter() - t0) * 1e3))
print("mass: sum(iwe) = %.1f, sum(p) = %.1f, sum(diwe) = %.3f %.3f" % (iwe.double().sum().item(), psum,
                                                                   diwe[0].double().sum().item(), diwe[1].double().sum().item()))
assert abs(iwe.double().sum().item() - psum) <= 1e-3 * n ** 0.5 + 1.0
h = (n // 2) & ~3
a = iwe_device(prm, ev.slice(0, h), (H, W), False, True, (H, W), impl="tiled", t_ref=ev.t_at(-1))[0]
b = iwe_device(prm, ev.slice(h, n), (H, W), False, True, (H, W), impl="tiled", t_ref=ev.t_at(-1))[0]
err = (a + b - iwe).abs().max().item() / iwe.abs().max().item()
print("additivity of the two halves: rel. err %.2e" % err)
assert err <= 1e-5
print("ok")
