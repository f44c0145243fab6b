"""Soak of the in-kernel hand-overs of the one-pass voxel path (cut tiles: agent-scope stores / loads + a relaxed ticket; the
partition's ticket): structured scenes at three sizes, both record formats, 150 launches each on warm caches with other
traffic in between -- unit polarities accumulate integers, so every launch must give the same bits as the first."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import event_utils_amd as E
from event_utils_amd import tiled
import voxel_sweep as V
torch.cuda.set_device(0)
H, W, B = 480, 640, 5
filler = torch.empty(96 << 20, dtype=torch.uint8, device="cuda")
bad = 0
for rec in ("8", "4"):
    tiled.FORCE["rec"] = int(rec)
    for kind in ("blob", "edges"):
        for n in (3_000_000, 6_000_000, 11_000_000):
            x, y, t, p = [np.ascontiguousarray(a) for a in V.scene(kind, n, H, W)]
            cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
            first = E.events_to_voxel_torch(*cols, B, sensor_size=(H, W))
            for i in range(150):
                if i % 7 == 0: filler.random_(0, 255)
                if not torch.equal(E.events_to_voxel_torch(*cols, B, sensor_size=(H, W)), first):
                    bad += 1
            print(rec, kind, n, "mismatches so far", bad, flush=True)
# the event images (evk_image2.hip): integer image (int32 partial tiles) and unit-weight nearest image, cut tiles on the blob scene
tiled.FORCE["rec"] = None
os.environ["EVK_IMPL"] = "tiled"
for n in (3_000_000, 11_000_000):
    x, y, t, p = [np.ascontiguousarray(a) for a in V.scene("blob", n, H, W)]
    xi, yi, pi = x.astype(np.int64), y.astype(np.int64), p.astype(np.int64)
    xd, yd, pd = (torch.from_numpy(a).cuda() for a in (x, y, p))
    first_i = E.events_to_image(xi, yi, pi, sensor_size=(H, W))
    first_f = E.events_to_image_torch(xd, yd, pd, sensor_size=(H, W), interpolation=None, padding=False)
    for i in range(100):
        if i % 7 == 0: filler.random_(0, 255)
        if i % 10 == 0 and not np.array_equal(E.events_to_image(xi, yi, pi, sensor_size=(H, W)), first_i):
            bad += 1
        if not torch.equal(E.events_to_image_torch(xd, yd, pd, sensor_size=(H, W), interpolation=None, padding=False), first_f):
            bad += 1
    print("image blob", n, "mismatches so far", bad, flush=True)
print("SOAK", "FAILED" if bad else "ok")
