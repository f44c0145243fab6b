"""Times the partition stages alone (no tile kernel): used for A/B experiments on the scatter kernel via EVK_LIB_PATH."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from event_utils_amd import tiled
H, W, n = 480, 640, 10_000_000
rng = np.random.default_rng(1)
cols = [torch.from_numpy(a).cuda() for a in (rng.integers(0, W, n).astype(np.float32), rng.integers(0, H, n).astype(np.float32),
        np.sort(rng.uniform(0, 0.1, n)).astype(np.float32), (rng.integers(0, 2, n) * 2 - 1).astype(np.float32))]
bk = tiled.bucket_events(*cols, 0, H, W, 5, 4)
run = lambda st: tiled.bucket_events(*cols, 0, H, W, 5, 4, stages=st, into=bk)
run(7)
print("scatter alone: %.1f us" % (tiled._time_ms(lambda: run(4), 20) * 1e3))
print("hist+scan+scatter: %.1f us" % (tiled._time_ms(lambda: run(7), 20) * 1e3))
