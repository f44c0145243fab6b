"""The IWE bucketing of a cold optimize() (evk_bucket_events_f32): stage times of the round-6 LDS-sorting scatter against the
write-combining ring scatter of rounds 1-5 (+ its separate compaction pass), 10 M events 640x480 and 50 M events 1280x720,
real-valued and sensor-pixel coordinates; and a bit-for-bit comparison of the two scatters' records and index.
    python tools/bucket_time.py [--big]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from event_utils_amd import _lib, tiled  # noqa: E402


def run(n, H, W, pixel):
    x, y, t, p = bench.structured_scene(3, n, H, W)
    if pixel:
        x, y = np.floor(x), np.floor(y)
    cols = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (x, y, t, p)]
    dom_h, dom_w = H + 1, W + 1
    tw, th = tiled.iwe_tile_shape(dom_h, dom_w)
    res = {}
    for legacy in (True, False):
        tiled.FORCE["legacy_scatter"] = legacy
        wc = tiled.want_compact(1, n, tw, th) and not legacy
        kw = dict(stats=not legacy, compact=wc)
        bk = tiled.bucket_events(*cols, 1, dom_h, dom_w, tw, th, **kw)
        full = lambda: tiled.bucket_events(*cols, 1, dom_h, dom_w, tw, th, into=bk0, **kw)   # noqa: E731
        bk0 = tiled.bucket_events(*cols, 1, dom_h, dom_w, tw, th, **kw)
        ms = {"all": tiled._time_ms(full, 10)}
        # (the scan works IN PLACE on the histogram's table: a stage can only be timed behind the stages before it)
        acc = {st: tiled._time_ms(lambda st=st: tiled.bucket_events(*cols, 1, dom_h, dom_w, tw, th, into=bk0, stages=st, **kw), 10)
               for st in (1, 3)}
        ms["hist"], ms["scan"], ms["scatter"] = acc[1], acc[3] - acc[1], ms["all"] - acc[3]
        if legacy:
            out = torch.empty(int(_lib.lib().evk_compact_records_bytes(n)) // 8, dtype=torch.int64, device="cuda")
            verdict = torch.zeros(1, dtype=torch.int32, device="cuda")
            from event_utils_amd import _device as D
            ms["compact_pass"] = tiled._time_ms(lambda: _lib.call("evk_compact_records_f32", D.ptr(bk0.records), n, dom_h, dom_w, tw, th,
                                                                  D.ptr(out), D.ptr(verdict), D.stream()), 10) if tiled.want_compact(1, n, tw, th) else 0.0
        bk = bk.compact()
        res[legacy] = (bk, ms)
        print("n=%d %dx%d %s %-6s: %s  -> records %s" % (n, W, H, "pixel" if pixel else "real ", "legacy" if legacy else "sorted",
              {k: round(v, 4) for k, v in ms.items()}, "compact 8 B" if bk.iwe_flag else "full 16 B"), flush=True)
    tiled.FORCE["legacy_scatter"] = False
    a, b = res[True][0], res[False][0]
    ia, ib = a.bucket_start.cpu().numpy(), b.bucket_start.cpu().numpy()
    used = 3 * a.ntiles + 2 + int(ia[2 * a.ntiles + 1])     # offsets, item offsets, counters, the USED item -> tile entries
    same_idx = bool(np.array_equal(ia[:used], ib[:used]))
    # the records of a tile are the same SET in both (their order inside one block's share of a tile is the order of LDS
    # atomics in either scatter): compare them sorted within every tile
    T = a.ntiles
    starts = a.bucket_start[: T + 1].cpu().numpy().astype(np.int64)
    tile_of = np.repeat(np.arange(T), np.diff(starts))

    def canon(bk):
        if bk.iwe_flag:
            r = bk.records[: bk.n].cpu().numpy().view(np.uint32).reshape(-1, 2)
            keys = (r[:, 1], r[:, 0], tile_of)
        else:
            r = bk.records.cpu().numpy().reshape(-1, 4).view(np.uint32)[: bk.n]
            keys = (r[:, 3], r[:, 1], r[:, 0], r[:, 2], tile_of)
        return r[np.lexsort(keys)]
    same_rec = a.iwe_flag == b.iwe_flag and np.array_equal(canon(a), canon(b))
    print("   identical index: %s, identical records per tile: %s" % (same_idx, same_rec), flush=True)
    assert same_idx and same_rec


if __name__ == "__main__":
    torch.cuda.set_device(0)
    for pixel in (False, True):
        run(10_000_000, 480, 640, pixel)
    tiled.FORCE["iwe_records"] = "compact"
    run(10_000_000, 480, 640, True)
    tiled.FORCE["iwe_records"] = "auto"
    if "--big" in sys.argv:
        for pixel in (False, True):
            run(50_000_000, 720, 1280, pixel)
