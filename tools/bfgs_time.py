"""optimize_contrast: scipy's fmin_bfgs (the reference's optimiser) against optimizer='evk_bfgs' -- wall time, event
passes and argmax -- on the moving-edge scene at configs[2] (10 M events, 640x480) and configs[3] (50 M, 1280x720) size,
analytic (consistent) and numeric gradients.  usage: python tools/bfgs_time.py [--big]"""
import os
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import event_utils_amd as E  # noqa: E402
from event_utils_amd.contrast_max.events_cmax import optimize_contrast  # noqa: E402


def run(n, H, W):
    x, y, t, p = bench.structured_scene(3, n, H, W)
    ev = E.DeviceEvents.from_arrays(x, y, t, p, precision="f32")
    w = E.linvel_warp()
    for numeric, exact in ((False, False), (True, True)):
        res = {}
        for optimizer in ("scipy", "evk_bfgs"):
            o = E.variance_objective()
            o.sensor_size, o.reference_exact = (H, W), exact
            passes = [0]
            for name in ("evaluate_function", "evaluate_gradient", "evaluate_function_and_gradient",
                         "evaluate_function_and_numeric_gradient", "evaluate_numeric_gradient"):
                fn = getattr(o, name)

                def wrapped(*a, _fn=fn, **k):
                    passes[0] += 1
                    return _fn(*a, **k)
                setattr(o, name, wrapped)
            fb = o.evaluate_function_batch

            def fbw(*a, **k):
                r = fb(*a, **k)
                passes[0] += o.batch_passes
                return r
            o.evaluate_function_batch = fbw
            kw = {} if optimizer == "scipy" else {"optimizer": "evk_bfgs"}
            best = None
            for rep in range(3):
                passes[0] = 0
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    a = optimize_contrast(ev, None, None, None, w, o, numeric_grads=numeric, blur_sigma=1.0, img_size=(H, W), **kw)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            res[optimizer] = (best, passes[0], np.asarray(a, dtype=float))
        s, e = res["scipy"], res["evk_bfgs"]
        print("n=%d %dx%d %s: scipy %.2f ms, %d passes -> (%.3f, %.3f) | evk_bfgs %.2f ms, %d passes -> (%.3f, %.3f) | x%.2f, |d argmax| %.4f px/s"
              % (n, W, H, "numeric grads (reference default)" if numeric else "analytic consistent grad", s[0] * 1e3, s[1], s[2][0], s[2][1],
                 e[0] * 1e3, e[1], e[2][0], e[2][1], s[0] / e[0], float(np.linalg.norm(s[2] - e[2]))), flush=True)


if __name__ == "__main__":
    run(10_000_000, 480, 640)
    if "--big" in sys.argv:
        run(50_000_000, 720, 1280)
