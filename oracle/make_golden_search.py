"""
TEST INFRASTRUCTURE ONLY.  Generates tests/golden/f14_search.npz by calling the REAL reference's parameter-space
samplers (events_cmax.py: grid_search_initial :241, grid_search_optimisation :186, find_new_range :162,
draw_objective_function :103, segmentation_mask_from_d_iwe :78; objectives.py: rms_objective :266,
cut_events_to_lifespan :143) on the structured scene of f8_objective.
Run in the build container only:   python -m oracle.make_golden_search

Two things the reference needs from its environment to run at all, supplied here WITHOUT touching its source:
 * numpy < 1.16 semantics for np.vstack(<map object>) (events_cmax.py:294): the module's `np` global is replaced
   by a proxy whose vstack materialises the iterator first;
 * the name `recursive_search` (events_cmax.py:233, never defined): bound to grid_search_optimisation itself;
 * the name `plt` (events_cmax.py:133-158, never imported): bound to matplotlib.pyplot, with imshow capturing the image.
"""
import contextlib
import io
import types

import numpy as np
import torch

from . import ref_loader
from .make_golden import gen_structured, save


class _NumpyCompat(types.ModuleType):
    def __init__(self):
        super().__init__("numpy_compat")

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def vstack(tup, **kw):
        return np.vstack(list(tup), **kw)


def main():
    torch.set_num_threads(1)
    ref = ref_loader.load()
    O, C, Wp = ref.objectives, ref.events_cmax, ref.warps
    C.np = _NumpyCompat()
    C.recursive_search = C.grid_search_optimisation
    w = Wp.linvel_warp()
    H, Wd, n = 180, 240, 30000
    x, y, t, p = gen_structured(80, n, H, Wd)          # identical to f8_objective's events
    out = dict(events_from=np.array("f8_objective"), img_size=np.array([H, Wd]))
    quiet = contextlib.redirect_stdout(io.StringIO())

    with quiet:
        for tag, kw in (("log5", dict(log_scale=True, num_samples_per_param=5)),
                        ("lin7", dict(log_scale=False, num_samples_per_param=7, param_ranges=[[-60, 60], [-90, 30]]))):
            r = C.grid_search_initial(x, y, t, p, w, O.variance_objective(), (H, Wd), **kw)
            out["gsi_%s_params" % tag] = np.array(r["params"], dtype=np.float64)
            out["gsi_%s_eval" % tag] = np.array(r["eval"], dtype=np.float64)
            out["gsi_%s_axes" % tag] = np.array(r["search_axes"], dtype=np.float64)
            out["gsi_%s_min_params" % tag] = np.array(r["min_params"], dtype=np.float64)
            out["gsi_%s_min_eval" % tag] = np.float64(r["min_func_eval"])
        calls = []
        gsi = C.grid_search_initial

        def counted(*a, **k):
            r = gsi(*a, **k)
            calls.append(np.array(r["min_params"], dtype=np.float64))
            return r
        C.grid_search_initial = counted
        r = C.grid_search_optimisation(x, y, t, p, w, O.variance_objective(), (H, Wd), log_scale=False)
        C.grid_search_initial = gsi
        out["gso_min_params"] = np.array(r["min_params"], dtype=np.float64)
        out["gso_min_eval"] = np.float64(r["min_func_eval"])
        out["gso_level_min_params"] = np.array(calls)
        print_levels = len(calls)

    axes = np.array([-150., -75., 0., 75., 150.])
    prm = np.array([-200., -150., -100., -75., 0., 10., 75., 149., 150., 400.])
    out["fnr_axes"], out["fnr_params"] = axes, prm
    out["fnr_ranges"] = np.array([C.find_new_range(axes, q) for q in prm], dtype=np.float64)

    captured = []
    import matplotlib.pyplot as plt       # events_cmax.py uses `plt` (:133-158) without importing it: supply it
    C.plt = plt
    imshow = C.plt.imshow
    C.plt.imshow = lambda img, **k: captured.append(np.array(img))
    try:
        with quiet:
            C.draw_objective_function(x, y, t, p, O.variance_objective(minimum_events=1), w, x_range=(-100, 100),
                                      y_range=(-80, 60), resolution=20, img_size=(H, Wd), show=False)
    finally:
        C.plt.imshow = imshow
        C.plt.close("all")
    out["landscape"] = captured[0]
    out["landscape_args"] = np.array([-100, 100, -80, 60, 20], dtype=np.float64)

    params12 = np.array([[0, 0], [40, -25], [-10, 60]], dtype=np.float64)
    rms = O.rms_objective()
    out["rms_params"] = params12
    out["rms_f"] = np.array([np.float64(rms.evaluate_function(q, x, y, t, p, w, (H, Wd), blur_sigma=s))
                             for q in params12 for s in (None, 0.0)])
    out["rms_g"] = np.array([np.asarray(rms.evaluate_gradient(q, x, y, t, p, w, (H, Wd), blur_sigma=s), dtype=np.float64)
                             for q in params12 for s in (None, 0.0)])

    with quiet:
        cut = O.cut_events_to_lifespan(x, y, t, p, np.array([400., -250.]), 5, minimum_events=5000)
        cut2 = O.cut_events_to_lifespan(x, y, t, p, np.array([4000., -2500.]), 5, minimum_events=5000)
    out["cut_first_t"] = np.array([cut[2][0], cut2[2][0]])
    out["cut_len"] = np.array([len(cut[0]), len(cut2[0])])

    _, d_iwe = O.get_iwe(np.array([40., -25.]), x, y, t, p, w, (H, Wd), compute_gradient=True)
    out["seg_d_iwe"] = d_iwe
    out["seg_mask"] = C.segmentation_mask_from_d_iwe(d_iwe).astype(np.uint8)
    out["seg_mask_th"] = C.segmentation_mask_from_d_iwe(d_iwe, th=0.05).astype(np.uint8)
    save("f14_search", **out)
    print("  grid_search_optimisation: %d levels, min_params=%s" % (print_levels, out["gso_min_params"]))


if __name__ == "__main__":
    main()
