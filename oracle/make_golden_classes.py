"""
TEST INFRASTRUCTURE ONLY.  Generates tests/golden/f17_image_classes.npz: the two stateful image classes of
lib/representations/image.py:355-396 -- TimestampImage (last event per pixel wins, dense-rank normalisation) and EventImage
(polarity accumulation; its add_events passes p = 0, so only add_event changes the image) -- driven by the REAL reference on
seeded events: the images after every step and the get_image() outputs.
Run in the build container only:   python -m oracle.make_golden_classes
"""
import numpy as np

from . import ref_loader
from .make_golden import save


def main():
    ref = ref_loader.load()
    I = ref.image
    H, W, n = 60, 80, 20000
    rng = np.random.default_rng(170)
    # float coordinates (int() truncates toward zero), a few negative ones (numpy indexing wraps them once), many events per pixel
    xs = rng.uniform(0, W, n)
    ys = rng.uniform(0, H, n)
    xs[:40] = -rng.uniform(0.0, 3.0, 40)
    ys[40:90] = -rng.uniform(0.0, 5.0, 50)
    ts = np.sort(rng.uniform(0.0, 0.5, n))
    ts[100:140] = ts[100]                               # equal time stamps: one dense rank
    ps = rng.integers(0, 2, n) * 2.0 - 1.0
    out = dict(xs=xs, ys=ys, ts=ts, ps=ps, sensor_size=np.array([H, W]))

    ti = I.TimestampImage((H, W))
    out["ts_init_image"] = ti.get_image()               # all ones: 0 / 0
    ti.set_init(-1.0)
    half = n // 2
    ti.add_events(xs[:half], ys[:half], ts[:half], ps[:half])
    out["ts_image_half"] = ti.image.copy()
    out["ts_get_half"] = ti.get_image()
    ti.add_events(xs[half:], ys[half:], ts[half:], ps[half:])
    ti.add_event(3.7, 5.2, 0.123, 1)
    out["ts_image_full"] = ti.image.copy()
    out["ts_get_full"] = ti.get_image()
    # a sparse case: most pixels keep the initial value (they share rank 0)
    ts2 = I.TimestampImage((H, W))
    ts2.add_events(xs[:300], ys[:300], ts[:300] + 2.0, ps[:300])
    out["ts_sparse_image"] = ts2.image.copy()
    out["ts_sparse_get"] = ts2.get_image()

    ei = I.EventImage((H, W))
    ei.add_events(xs, ys, ts, ps)                       # upstream passes p = 0: the image stays at ones
    out["ev_image_after_add_events"] = ei.image.copy()
    for k in range(0, 2000):
        ei.add_event(xs[k], ys[k], ts[k], ps[k])
    out["ev_image"] = ei.image.copy()
    out["ev_get"] = ei.get_image()
    save("f17_image_classes", **out)


if __name__ == "__main__":
    main()
