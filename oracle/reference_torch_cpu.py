"""
TEST INFRASTRUCTURE ONLY -- never imported by the product package.

The reference's TORCH path restated with the same torch primitives, on the CPU, all host threads: what
"the reference on this box's host cores" costs when it is allowed to use every core ("a multi-threaded
variant (torch CPU index_put_, threads = all cores) as the best-effort CPU").  Used by bench.py's cpu_baseline leg and
pinned to the golden vectors in tests/test_oracle_golden.py; the numpy restatement (reference_np.py) stays THE oracle.
"""
import torch


def events_to_image_torch_nearest(xs, ys, ps, sensor_size):
    """image.py:46-100, nearest branch with clip_out_of_range=False (the voxel path's call, voxel_grid.py:140-142):
    float coordinates truncated toward zero (.long(), :90-91), img.index_put_((ys, xs), ps, accumulate=True) (:95)."""
    img = torch.zeros(tuple(int(v) for v in sensor_size), dtype=torch.float32)
    img.index_put_((ys.long(), xs.long()), ps, accumulate=True)
    return img


def events_to_voxel_torch(xs, ys, ts, ps, B, sensor_size=(180, 240)):
    """voxel_grid.py:114-153 (temporal_bilinear=True): B passes, each a full nearest-pixel accumulate of
    ps * max(0, 1 - |t_norm - b|) with t_norm = (ts - ts[0]) / (ts[-1] - ts[0]) * (B - 1) in float32."""
    dt = ts[-1] - ts[0]
    t_norm = (ts - ts[0]) / dt * (B - 1)
    zeros = torch.zeros(t_norm.size())
    bins = []
    for bi in range(B):
        weights = ps * torch.max(zeros, 1.0 - torch.abs(t_norm - bi))
        bins.append(events_to_image_torch_nearest(xs, ys, weights, sensor_size))
    return torch.stack(bins)
