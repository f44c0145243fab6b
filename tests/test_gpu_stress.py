"""GPU (-m gpu): randomized tiled-vs-direct equivalence (both are HIP paths; the direct one is the simplest possible
kernel and is itself pinned to the oracle in test_gpu_parity.py) over odd sensor sizes, bin counts, event counts, flows,
clustered and degenerate inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(a, b, tol=1e-5):
    a, b = a.double(), b.double()
    scale = max(b.abs().max().item(), 1e-30)
    assert (a - b).abs().max().item() <= tol * scale


@pytest.mark.parametrize("seed", range(12))
def test_random_voxel_tiled_equals_direct(seed):
    from event_utils_amd.representations.voxel_grid import _voxel_f32_device
    rng = np.random.default_rng(1000 + seed)
    H, W = int(rng.integers(5, 300)), int(rng.integers(5, 400))
    B = int(rng.integers(1, 10))
    n = int(rng.choice([17, 300, 4099, 70_001, 333_333]))
    kind = seed % 4
    x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
    if kind == 1:       # clustered
        hot = rng.random(n) < 0.8
        x[hot] = W // 2; y[hot] = H // 2
    if kind == 2:       # fractional + a few negative (wrap) coordinates
        x = np.clip(x + rng.random(n).astype(np.float32), 0, W - 1).astype(np.float32)
        x[:3] = -1.0
    t = np.sort(rng.uniform(5.0, 5.1, n)).astype(np.float32)
    p = rng.normal(size=n).astype(np.float32) if kind == 3 else (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    a = _voxel_f32_device(*cols, B, (H, W), float(t[0]), float(t[-1]), impl="tiled")
    b = _voxel_f32_device(*cols, B, (H, W), float(t[0]), float(t[-1]), impl="direct")
    _close(a, b)
    assert abs(a.double().sum().item() - float(p.astype(np.float64).sum())) <= 1e-4 * max(1.0, np.abs(p).sum())


@pytest.mark.parametrize("seed", range(12))
def test_random_iwe_tiled_equals_direct(seed):
    from event_utils_amd.contrast_max.objectives import iwe_device
    from event_utils_amd.events import DeviceEvents
    rng = np.random.default_rng(2000 + seed)
    H, W = int(rng.integers(20, 300)), int(rng.integers(20, 400))
    n = int(rng.choice([33, 2000, 50_001, 250_000]))
    x = rng.uniform(-2, W + 2, n).astype(np.float32); y = rng.uniform(-2, H + 2, n).astype(np.float32)
    if seed % 3 == 1:
        hot = rng.random(n) < 0.85
        x[hot] = (W / 3 + rng.random(hot.sum())).astype(np.float32); y[hot] = (H / 2 + rng.random(hot.sum())).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.2, n)).astype(np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32) * (1 if seed % 4 else 37.5)
    prm = rng.normal(0, [10, 100, 1000][seed % 3], 2)
    ev = DeviceEvents.from_arrays(x, y, t, p)
    if seed % 5 == 0:
        ev = ev.scaled(100.0)
    img_size = (H, W) if seed % 2 else (H + 7, W + 11)
    for grad in (False, True):
        for pol in (True, False):
            a = iwe_device(prm, ev, img_size, grad, pol, (H, W), impl="tiled")
            b = iwe_device(prm, ev, img_size, grad, pol, (H, W), impl="direct")
            _close(a[0], b[0])
            if grad:
                _close(a[1], b[1])


def test_objective_values_are_reproducible_run_to_run():
    """Fixed-point LDS windows + sorted gather lists + two-stage reductions: repeated evaluations are bit-identical."""
    import event_utils_amd as E
    from event_utils_amd.events import DeviceEvents
    rng = np.random.default_rng(7)
    H, W, n = 240, 320, 600_000
    x = rng.uniform(1, W - 1, n).astype(np.float32); y = rng.uniform(1, H - 1, n).astype(np.float32)
    t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32); p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    ev = DeviceEvents.from_arrays(x, y, t, p)
    obj, w = E.variance_objective(), E.linvel_warp()
    obj.sensor_size = (H, W)
    prm = np.array([33.3, -21.7])
    fs = {float(obj.evaluate_function(prm, ev, None, None, None, w, (H, W), 1.0)) for _ in range(5)}
    gs = {tuple(obj.evaluate_gradient(prm, ev, None, None, None, w, (H, W), 1.0).tolist()) for _ in range(5)}
    assert len(fs) == 1 and len(gs) == 1
