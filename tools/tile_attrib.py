"""Attribution table of the one-pass voxel call's two kernels (round 6): sensor / tiling (640x480 -> 512
tiles of 40x15, 1280x720 -> 1020 tiles of 38x24) x record size (4 | 8 bytes) x event count (10 M | 50 M), uniform events with
unit polarities generated ON the device, every timed call reading its events from HBM (10 M-event cases rotate over four
streams).  Also the counting modes (count / count2 on and off) per case.

    python tools/tile_attrib.py [--quick] [--only 720p50]        (EVK_LIB_PATH selects an A/B build)
--case HxWxN[xREC] times one explicit case (e.g. 720x1280x50000000x4), --tile TWxTH forces a tile size."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from event_utils_amd import tiled  # noqa: E402


def stream(seed, n, H, W, dev):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = torch.randint(0, W, (n,), device=dev, generator=g).float()
    y = torch.randint(0, H, (n,), device=dev, generator=g).float()
    t = (torch.rand(n, device=dev, generator=g) * 0.1).sort().values
    t[0], t[-1] = 0.0, 0.1
    p = (torch.randint(0, 2, (n,), device=dev, generator=g) * 2 - 1).float()
    return [x.contiguous(), y.contiguous(), t.contiguous(), p.contiguous()]


def case(H, W, n, rec, B=5, reps=10, modes=("default",)):
    dev = torch.device("cuda", 0)
    nsets = 4 if n * 16 <= (256 << 20) else 1
    sets = [stream(100 + k, n, H, W, dev) for k in range(nsets)]
    out = []
    for mode in modes:
        tiled.FORCE["rec"] = rec
        tiled.FORCE["count"] = mode != "f64"          # (EVK_VOXEL2_NO_COUNT switches both integer modes off)
        try:
            k = tiled.time_voxel_kernels(sets, 0.0, 0.1, B, H, W, impl="tiled", reps=reps)
        finally:
            tiled.FORCE["rec"], tiled.FORCE["count"] = None, True
        alg = 16.0 * n + B * H * W * 4
        out.append((mode, k))
        print("%4dx%-4d n=%-9d rec=%s %-8s total %.4f ms (frac %.3f)  part %.4f  tiles %.4f   [%s]" % (
            W, H, n, rec or "-", mode, k["total_ms"], alg / (k["total_ms"] * 1e-3) / 8e12, k["kernels_ms"]["k_part_sorted"],
            k["kernels_ms"]["k_voxel_tiles2"], k["impl"]), flush=True)
    del sets
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    torch.cuda.set_device(0)
    print("lib: %s" % os.environ.get("EVK_LIB_PATH", "product"), flush=True)
    if "--tile" in sys.argv:
        tiled.FORCE["tile"] = tuple(int(v) for v in sys.argv[sys.argv.index("--tile") + 1].split("x"))
    if "--case" in sys.argv:
        spec = [int(v) for v in sys.argv[sys.argv.index("--case") + 1].split("x")]
        case(spec[0], spec[1], spec[2], spec[3] if len(spec) > 3 else None, modes=("default",))
        sys.exit(0)
    quick = "--quick" in sys.argv
    for (H, W) in ((480, 640), (720, 1280)):
        for n in (10_000_000, 50_000_000):
            for rec in (4, 8):
                case(H, W, n, rec, modes=("default",) if quick else ("default", "f64"))
