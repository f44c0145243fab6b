"""Soak of the in-kernel hand-overs of the one-pass voxel path (cut tiles: agent-scope stores / loads + a relaxed ticket; the
partition's ticket): structured scenes at three sizes, both record formats, 150 launches each on warm caches with other
traffic in between -- unit polarities accumulate integers, so every launch must give the same bits as the first.
  --neighbour   (round 5) the same next to a SECOND PROCESS that keeps the GPU busy the whole time: it streams 2 x 256 MB
                buffers through the caches and holds an uneven 37 CUs with spin workgroups (24 KB of LDS each), so that the
                hand-overs run under uneven load with the L2s churned by foreign traffic; 60 launches per case, plus the live
                call (evk_voxel_live.h: consumer kernel on a second stream) and the 1280x720 counting mode."""
import os, sys, subprocess, time, numpy as np, torch
if "--as-neighbour" in sys.argv:      # the second process: stream + spin until the deadline
    import ctypes
    torch.cuda.set_device(0)
    spin = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libspin.so"))
    spin.spin_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    a = torch.empty(64 << 20, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
    sink = torch.zeros(1, dtype=torch.int32, device="cuda"); side = torch.cuda.Stream()
    a.uniform_()
    end = time.time() + float(sys.argv[sys.argv.index("--as-neighbour") + 1])
    k = 0
    while time.time() < end:
        spin.spin_launch(37, 300.0, ctypes.c_void_p(side.cuda_stream), ctypes.c_void_p(sink.data_ptr()), 24 * 1024)
        for _ in range(8):
            b.copy_(a); a.mul_(1.0000001)
        k += 1
        if k % 16 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print("neighbour: %d rounds" % k, flush=True)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import event_utils_amd as E
from event_utils_amd import tiled
import voxel_sweep as V
torch.cuda.set_device(0)
H, W, B = 480, 640, 5
filler = torch.empty(96 << 20, dtype=torch.uint8, device="cuda")
bad = 0
NEIGHBOUR = "--neighbour" in sys.argv
LAUNCHES = 60 if NEIGHBOUR else 150
nb = None
if NEIGHBOUR:
    nb = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--as-neighbour", "600"])
    time.sleep(8.0)      # (its first import of torch and its warm-up)
    print("neighbour process %d running" % nb.pid, flush=True)
for rec in ("8", "4"):
    tiled.FORCE["rec"] = int(rec)
    for kind in ("blob", "edges"):
        for n in (3_000_000, 6_000_000, 11_000_000):
            x, y, t, p = [np.ascontiguousarray(a) for a in V.scene(kind, n, H, W)]
            cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
            first = E.events_to_voxel_torch(*cols, B, sensor_size=(H, W))
            for i in range(LAUNCHES):
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
if NEIGHBOUR:
    # the live call (uniform events: every tile finished by a consumer; blob: hot tiles left to the tile kernel proper) ...
    for kind in ("uniform", "blob"):
        n = 8_000_000
        x, y, t, p = [np.ascontiguousarray(a) for a in (V.synth(5, n, H, W) if kind == "uniform" else V.scene(kind, n, H, W))]
        cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
        tiled.FORCE["live"] = False
        first = E.events_to_voxel_torch(*cols, B, sensor_size=(H, W))
        tiled.FORCE["live"] = True
        for i in range(LAUNCHES):
            if i % 7 == 0: filler.random_(0, 255)
            if not torch.equal(E.events_to_voxel_torch(*cols, B, sensor_size=(H, W)), first):
                bad += 1
        tiled.FORCE["live"] = None
        print("live", kind, n, "mismatches so far", bad, flush=True)
    # ... and the counting mode in the float64 mode's planes (1280x720), cut tiles handing over exact int64 cells
    x, y, t, p = [np.ascontiguousarray(a) for a in V.scene("blob", 9_000_000, 720, 1280)]
    cols = [torch.from_numpy(a).cuda() for a in (x, y, t, p)]
    first = E.events_to_voxel_torch(*cols, B, sensor_size=(720, 1280))
    for i in range(LAUNCHES):
        if i % 7 == 0: filler.random_(0, 255)
        if not torch.equal(E.events_to_voxel_torch(*cols, B, sensor_size=(720, 1280)), first):
            bad += 1
    print("720p blob", "mismatches so far", bad, flush=True)
    nb.kill(); nb.wait()
print("SOAK", "FAILED" if bad else "ok")
