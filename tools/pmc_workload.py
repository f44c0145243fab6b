"""The voxel call of one bench workload, alone, for the rocprofv3 PMC passes (tools/profile_round.sh):
    python tools/pmc_workload.py c2        10 M events, 640x480x5   (configs[1], the headline)
    python tools/pmc_workload.py c5_share  50 M events, 1280x720x5  (one rank's share of configs[4])
    python tools/pmc_workload.py img_nearest | img_bilinear | img_timestamp   10 M events, 640x480: events_to_image (int32) /
                                           events_to_image_torch(bilinear) / the average-timestamp planes through the one-pass path
                                           (evk_image2.hip)
Runs the internal entry point (resident grid, no per-call checks) 8 times so that the counters see exactly the kernels
of the call: k_part_sorted and k_voxel_tiles2 (evk_voxel2.hip)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_utils_amd.representations.voxel_grid import _voxel_f32_device  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "c2"
CALLS = int(sys.argv[2]) if len(sys.argv) > 2 else 8   # (PMC passes: 8; kernel-trace stats at steady clocks: a few hundred)
if tag.startswith("img_"):     # the event images of bench.py's image_10m block: 10 M events, 640x480, one-pass path
    from event_utils_amd import tiled  # noqa: E402
    n, H, W = 10_000_000, 480, 640
    rng = np.random.default_rng(5)
    x = rng.uniform(0, W - 1, n).astype(np.float32); y = rng.uniform(0, H - 1, n).astype(np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    xd, yd, pd = (torch.from_numpy(a).cuda() for a in (x, y, p))
    inf = float("inf")
    if tag == "img_timestamp":   # bench.py's image_10m.events_to_timestamp_image_torch: the four planes, one-pass path (round 6)
        td = torch.sort(torch.rand(n, device="cuda") * 0.1).values.contiguous()
        planes = torch.zeros((4, H + 1, W + 1), dtype=torch.float32, device="cuda")
        for _ in range(CALLS):
            assert tiled.timestamp_images2(xd, yd, td, pd, n, H + 1, W + 1, float(W), float(H), 0, 0.0, 0.1, planes, None)
        torch.cuda.synchronize()
        print("done", tag)
        sys.exit(0)
    if tag == "img_nearest":
        cols, kind, img = (xd.int(), yd.int(), pd.int()), "i32", torch.zeros((H, W), dtype=torch.int32, device="cuda")
    else:
        cols, kind, img = (xd, yd, pd), "bilinear", torch.zeros((H, W), dtype=torch.float32, device="cuda")
    for _ in range(CALLS):
        assert tiled.image2(kind, *cols, n, H, W, inf, inf, img, None, fresh=(kind != "bilinear"))
    torch.cuda.synchronize()
    print("done", tag)
    sys.exit(0)
if tag == "prebucketed":   # bench.py's `prebucketed` block: the north_star kernel on records bucketed once (k_voxel_tiled, 16 B/event)
    from event_utils_amd import tiled, _lib, _device as D  # noqa: E402
    n, H, W, B, tw, th = 10_000_000, 480, 640, 5, 4, 4
    rng = np.random.default_rng(1)
    x = rng.integers(0, W, n).astype(np.float32); y = rng.integers(0, H, n).astype(np.float32)
    t = np.sort(rng.uniform(0.0, 0.1, n)).astype(np.float32); p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    bk = tiled.bucket_events(*cols, 0, H, W, tw, th)
    nbytes = int(_lib.lib().evk_voxel_tiled_staging_bytes(bk.ntiles, n, B, tw, th))
    staging = torch.empty(max(nbytes, 1), dtype=torch.uint8, device="cuda")
    out = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
    for _ in range(CALLS):
        _lib.call("evk_voxel_tiled_f32", D.ptr(bk.records), D.ptr(bk.bucket_start), n, H, W, tw, th, float(t[0]), float(t[-1]), B,
                  _lib.EVK_VOXEL_OVERWRITE, D.ptr(out), D.ptr(staging), nbytes, D.stream())
    torch.cuda.synchronize()
    print("done", tag)
    sys.exit(0)
n, H, W, B = (10_000_000, 480, 640, 5) if tag == "c2" else (50_000_000, 720, 1280, 5)
rng = np.random.default_rng(1 if tag == "c2" else 40)
x = rng.integers(0, W, n).astype(np.float32)
y = rng.integers(0, H, n).astype(np.float32)
t = np.sort(rng.uniform(0.0, 0.1, n)).astype(np.float32)
p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
out = torch.empty((B, H, W), dtype=torch.float32, device="cuda")
for _ in range(CALLS):
    _voxel_f32_device(*cols, B, (H, W), float(t[0]), float(t[-1]), out=out, check=False, fresh=True)
torch.cuda.synchronize()
print("done", tag)
