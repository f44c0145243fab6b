"""The north_star kernel on pre-bucketed records (evk_voxel_tiled_f32, 16-byte records): kernel time against the tile shape,
10 M events 640x480x5, four rotating bucketed streams (HBM-resident)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tile_attrib as T  # noqa: E402
from event_utils_amd import _lib, tiled, _device as D  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
n, H, W, B = 10_000_000, 480, 640, 5
sets = [T.stream(100 + k, n, H, W, dev) for k in range(4)]
L = _lib.lib()
for tw, th in ((5, 4), (4, 4), (5, 3), (4, 5), (6, 3), (5, 5), (3, 4), (4, 3)):
    if L.evk_bucket_num_tiles(H, W, tw, th) <= 0:
        continue
    bks = [tiled.bucket_events(*c, 0, H, W, tw, th) for c in sets]
    nbytes = int(L.evk_voxel_tiled_staging_bytes(bks[0].ntiles, n, B, tw, th))
    staging = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    out = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    it = [0]

    def kernel():
        it[0] += 1
        bk = bks[it[0] % 4]
        _lib.call("evk_voxel_tiled_f32", D.ptr(bk.records), D.ptr(bk.bucket_start), n, H, W, tw, th, 0.0, 0.1, B,
                  _lib.EVK_VOXEL_OVERWRITE, D.ptr(out), D.ptr(staging), nbytes, D.stream())
    ms = tiled._time_ms(kernel, 20)
    alg = 16.0 * n + B * H * W * 4
    print("tiles %3dx%-3d (%4d): %.4f ms  frac %.3f" % (1 << tw, 1 << th, bks[0].ntiles, ms, alg / (ms * 1e-3) / 8e12), flush=True)
    del bks
    torch.cuda.empty_cache()
