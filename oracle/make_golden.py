"""
TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz by calling the REAL reference
(/root/reference, imported through oracle/ref_loader.py) on seeded synthetic inputs.
Run in the build container only:   python -m oracle.make_golden
Fixtures hold data only: inputs (stored in their narrowest lossless dtype) and the reference's outputs.
"""
import os
import numpy as np
import torch

from . import ref_loader

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def gen_events(seed, n, H, W, real_xy=False, t_hi=0.1, f32_exact=True, margin=1.0):
    """Synthetic events per SURVEY §8(d).  With f32_exact the float columns are representable in
    float32 (so they can be stored as f32 and fed to f32 device columns losslessly)."""
    rng = np.random.default_rng(seed)
    if real_xy:
        x = rng.uniform(margin, W - margin, n)
        y = rng.uniform(margin, H - margin, n)
    else:
        x = rng.integers(0, W, n).astype(np.float64)
        y = rng.integers(0, H, n).astype(np.float64)
    t = np.sort(rng.uniform(0.0, t_hi, n))
    p = rng.integers(0, 2, n).astype(np.float64) * 2 - 1
    if f32_exact:
        x = x.astype(np.float32).astype(np.float64)
        y = y.astype(np.float32).astype(np.float64)
        t = np.sort(t.astype(np.float32).astype(np.float64))
    return x, y, t, p


def gen_structured(seed, n, H, W, flow=(40.0, -25.0), t_hi=0.25):
    """Moving vertical (+1) / horizontal (-1) edges: a contrast surface with a clear optimum at `flow`
    for the objective / optimize fixtures."""
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(0.0, t_hi, n))
    vert = rng.random(n) < 0.5
    ex = rng.choice(np.arange(30, W - 30, 24), n).astype(np.float64) + rng.normal(0, 0.3, n)
    ey = rng.choice(np.arange(30, H - 30, 24), n).astype(np.float64) + rng.normal(0, 0.3, n)
    x0 = np.where(vert, ex, rng.uniform(25, W - 25, n))
    y0 = np.where(vert, rng.uniform(25, H - 25, n), ey)
    # an event observed at time t sits where its edge has moved to; warping back with `flow` re-aligns it
    x = x0 + (t - t[-1]) * flow[0]
    y = y0 + (t - t[-1]) * flow[1]
    p = np.where(vert, 1.0, -1.0)
    x = x.astype(np.float32).astype(np.float64)
    y = y.astype(np.float32).astype(np.float64)
    t = np.sort(t.astype(np.float32).astype(np.float64))
    return x, y, t, p


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-28s %8.1f kB" % (name, os.path.getsize(path) / 1e3))


def main():
    torch.set_num_threads(1)
    ref = ref_loader.load()
    I, V, U, Wp, O, C = ref.image, ref.voxel_grid, ref.event_util, ref.warps, ref.objectives, ref.events_cmax
    versions = np.array(["numpy " + np.__version__, "torch " + torch.__version__,
                         "scipy " + __import__("scipy").__version__])

    # ---- F1 events_to_image, nearest, integer (bit-exact target) --------------------------------
    H, Wd, n = 180, 240, 20000
    x, y, t, p = gen_events(10, n, H, Wd)
    xi, yi, pi = x.astype(np.int64), y.astype(np.int64), p.astype(np.int64)
    # edge cases: the pad row/column (x==W, y==H) is legal and cropped off
    xi[:7] = Wd
    yi[7:13] = H
    out = dict(xs=xi.astype(np.int16), ys=yi.astype(np.int16), ps=pi.astype(np.int8),
               sensor_size=np.array([H, Wd]), versions=versions)
    out["img_pm"] = I.events_to_image(xi, yi, pi, sensor_size=(H, Wd))
    out["img_cnt"] = I.events_to_image(xi, yi, np.ones_like(pi), sensor_size=(H, Wd))
    out["img_mean"] = I.events_to_image(xi, yi, pi, sensor_size=(H, Wd), meanval=True)
    out["img_mean_default"] = I.events_to_image(xi, yi, pi, sensor_size=(H, Wd), meanval=True, default=7)
    wf = np.random.default_rng(11).uniform(-2, 2, n)
    out["wf"] = wf
    out["img_wf"] = I.events_to_image(xi, yi, wf, sensor_size=(H, Wd))
    save("f1_image_nearest_int", **out)

    # ---- F2 events_to_voxel numpy ------------------------------------------------------------------
    out = dict(versions=versions)
    for tag, (H, Wd, n, Bs) in {"small": (48, 64, 20000, (1, 2, 5, 9)), "dvs": (180, 240, 20000, (5,))}.items():
        x, y, t, p = gen_events(20 + len(tag), n, H, Wd, f32_exact=False)
        xi, yi = x.astype(np.int64), y.astype(np.int64)
        out[tag + "_xs"], out[tag + "_ys"] = xi.astype(np.int16), yi.astype(np.int16)
        out[tag + "_ts"], out[tag + "_ps"] = t, p.astype(np.int8)
        out[tag + "_sensor_size"] = np.array([H, Wd])
        for B in Bs:
            out["%s_voxel_B%d" % (tag, B)] = V.events_to_voxel(xi, yi, t, p, B, sensor_size=(H, Wd))
    save("f2_voxel_numpy", **out)

    # ---- F3 events_to_voxel_torch (f32, fractional coords pin .long() truncation) ------------------
    out = dict(versions=versions)
    for tag, (H, Wd, n, Bs) in {"small": (48, 64, 20000, (1, 2, 5, 9)), "dvs": (180, 240, 20000, (5,))}.items():
        rng = np.random.default_rng(30 + len(tag))
        x = rng.uniform(0, Wd, n).astype(np.float32)
        y = rng.uniform(0, H, n).astype(np.float32)
        x[x >= Wd] = Wd - 1
        y[y >= H] = H - 1
        t = np.sort(rng.uniform(0, 0.1, n)).astype(np.float32)
        p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
        out[tag + "_xs"], out[tag + "_ys"], out[tag + "_ts"], out[tag + "_ps"] = x, y, t, p.astype(np.int8)
        out[tag + "_sensor_size"] = np.array([H, Wd])
        for B in Bs:
            v = V.events_to_voxel_torch(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(t),
                                        torch.from_numpy(p), B, sensor_size=(H, Wd))
            out["%s_voxel_B%d" % (tag, B)] = v.numpy()
    # integer (long) coordinates as the data loaders would pass them
    xl, yl = torch.from_numpy(x.astype(np.int64)), torch.from_numpy(y.astype(np.int64))
    out["dvs_voxel_B5_long"] = V.events_to_voxel_torch(xl, yl, torch.from_numpy(t), torch.from_numpy(p), 5,
                                                       sensor_size=(180, 240)).numpy()
    save("f3_voxel_torch", **out)

    # ---- F4 events_to_image_torch: bilinear (padding T/F) and nearest incl. clip quirk Q8 ---------
    H, Wd, n = 180, 240, 12000
    rng = np.random.default_rng(40)
    x = rng.uniform(0, Wd + 3, n).astype(np.float32)     # some events beyond the clip threshold
    y = rng.uniform(0, H + 3, n).astype(np.float32)
    p = rng.uniform(-2, 2, n).astype(np.float32)
    out = dict(xs=x, ys=y, ps=p, sensor_size=np.array([H, Wd]), versions=versions)
    tx, ty, tp = torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(p)
    out["bil_pad"] = I.events_to_image_torch(tx, ty, tp, sensor_size=(H, Wd), interpolation='bilinear', padding=True).numpy()
    out["bil_nopad"] = I.events_to_image_torch(tx, ty, tp, sensor_size=(H, Wd), interpolation='bilinear', padding=False).numpy()
    out["near_pad"] = I.events_to_image_torch(tx, ty, tp, sensor_size=(H, Wd), interpolation=None, padding=True).numpy()
    out["near_nopad"] = I.events_to_image_torch(tx, ty, tp, sensor_size=(H, Wd), interpolation=None, padding=False).numpy()
    out["near_default3"] = I.events_to_image_torch(tx, ty, tp, sensor_size=(H, Wd), interpolation=None, padding=False, default=3).numpy()
    # numpy front door with interpolation='bilinear' (sensor_size not forwarded -> (180,240); img==0 -> default)
    out["np_bil"] = I.events_to_image(x.astype(np.float64), y.astype(np.float64), p.astype(np.float64),
                                      sensor_size=(H, Wd), interpolation='bilinear', padding=False, default=0)
    out["np_bil_pad"] = I.events_to_image(x.astype(np.float64), y.astype(np.float64), p.astype(np.float64),
                                          sensor_size=(H, Wd), interpolation='bilinear', padding=True, default=0)
    save("f4_image_torch", **out)

    # ---- F5 linvel_warp.warp + events_bounds_mask --------------------------------------------------
    H, Wd, n = 180, 240, 10000
    x, y, t, p = gen_events(50, n, H, Wd, real_xy=True)
    out = dict(xs=x.astype(np.float32), ys=y.astype(np.float32), ts=t.astype(np.float32), ps=p.astype(np.int8),
               versions=versions, params=np.array([[0, 0], [30, -20], [-150, 150]], dtype=np.float64))
    w = Wp.linvel_warp()
    for i, prm in enumerate(out["params"]):
        xp, yp, jx, jy = w.warp(x, y, t, p, t[-1], prm, compute_grad=True)
        out["xp%d" % i], out["yp%d" % i], out["jx%d" % i], out["jy%d" % i] = xp, yp, jx, jy
        out["mask%d" % i] = U.events_bounds_mask(xp, yp, 0, Wd, 0, H)
    save("f5_warp", **out)

    # ---- F6 get_iwe verbatim (reference-exact, (181,241) canvas whatever img_size) -----------------
    H, Wd, n = 180, 240, 12000
    x, y, t, p = gen_events(60, n, H, Wd, real_xy=True)
    out = dict(xs=x.astype(np.float32), ys=y.astype(np.float32), ts=t.astype(np.float32), ps=p.astype(np.int8),
               versions=versions, params=np.array([[30, -20], [-600, 450]], dtype=np.float64),
               img_size=np.array([H, Wd]))
    for i, prm in enumerate(out["params"]):
        iwe, diwe = O.get_iwe(prm, x, y, t, p, w, (H, Wd), compute_gradient=True, use_polarity=True)
        out["iwe%d" % i], out["diwe%d" % i] = iwe, diwe
        if i == 0:
            iwe_np, _ = O.get_iwe(prm, x, y, t, p, w, (H, Wd), compute_gradient=False, use_polarity=False)
            out["iwe_nopol%d" % i] = iwe_np
    # Q1: img_size larger than the hard-wired (180,240) canvas
    H2, W2 = 200, 300
    x2, y2, t2, p2 = gen_events(61, n, 170, 230, real_xy=True)
    iwe, diwe = O.get_iwe(np.array([30., -20.]), x2, y2, t2, p2, w, (H2, W2), compute_gradient=True)
    out.update(q1_xs=x2.astype(np.float32), q1_ys=y2.astype(np.float32), q1_ts=t2.astype(np.float32),
               q1_ps=p2.astype(np.int8), q1_img_size=np.array([H2, W2]), q1_iwe=iwe, q1_diwe=diwe)
    save("f6_get_iwe", **out)

    # ---- F7 sized IWE: warp + bounds mask + events_to_image_drv(sensor_size=...) -------------------
    out = dict(versions=versions, params=np.array([30., -20.]))
    for tag, (H, Wd, n) in {"s48": (48, 64, 20000), "vga": (480, 640, 6000)}.items():
        x, y, t, p = gen_events(70 + len(tag), n, H, Wd, real_xy=True)
        prm = out["params"]
        xp, yp, jx, jy = w.warp(x, y, t, p, t[-1], prm, compute_grad=True)
        m = U.events_bounds_mask(xp, yp, 0, Wd, 0, H)
        iwe, diwe = I.events_to_image_drv(xp * m, yp * m, p * m, jx * m, jy * m, sensor_size=(H, Wd),
                                          interpolation='bilinear', compute_gradient=True)
        out[tag + "_xs"], out[tag + "_ys"], out[tag + "_ts"], out[tag + "_ps"] = \
            x.astype(np.float32), y.astype(np.float32), t.astype(np.float32), p.astype(np.int8)
        out[tag + "_sensor_size"] = np.array([H, Wd])
        out[tag + "_iwe"], out[tag + "_diwe"] = iwe, diwe
    save("f7_iwe_sized", **out)

    # ---- F11 gather / per-event contrast / timestamp images ("next" rows of the scope table) --------
    H, Wd, n = 180, 240, 8000
    x, y, t, p = gen_events(110, n, H, Wd, real_xy=True)
    prm = np.array([30., -20.])
    out = dict(xs=x.astype(np.float32), ys=y.astype(np.float32), ts=t.astype(np.float32), ps=p.astype(np.int8),
               versions=versions, params=prm)
    r = O.get_iwe(prm, x, y, t, p, w, (H, Wd), return_events=True, return_per_event_contrast=True)
    out["iwe"], out["ev_x"], out["ev_y"], out["contrast"] = r[0], r[2][0], r[2][1], r[3]
    rng = np.random.default_rng(111)
    gimg = rng.normal(size=(H + 1, Wd + 1)).astype(np.float32)
    gx = rng.uniform(0, Wd + 2, 3000)
    gy = rng.uniform(0, H + 2, 3000)
    out["g_img"], out["g_x"], out["g_y"] = gimg, gx, gy
    out["g_w"] = I.image_to_event_weights(gx, gy, gimg)
    xi = rng.uniform(0, Wd + 1.5, n)
    yi = rng.uniform(0, H + 1.5, n)
    out["ti_x"], out["ti_y"] = xi.astype(np.float32), yi.astype(np.float32)
    tsn = t + 3.0
    out["ti_ts64"] = tsn
    a, b = I.events_to_timestamp_image(out["ti_x"].astype(np.float64), out["ti_y"].astype(np.float64), tsn, p)
    out["ti_np_pos"], out["ti_np_neg"] = a, b
    a, b = I.events_to_timestamp_image(out["ti_x"].astype(np.float64), out["ti_y"].astype(np.float64), tsn, p,
                                       padding=False, normalize_timestamps=False)
    out["ti_np_nopad_pos"], out["ti_np_nopad_neg"] = a, b
    tt = [torch.from_numpy(v) for v in (out["ti_x"], out["ti_y"], tsn.astype(np.float32), p.astype(np.float32))]
    for rev in (False, True):
        a, b = I.events_to_timestamp_image_torch(*tt, timestamp_reverse=rev)
        out["ti_t_pos_rev%d" % rev], out["ti_t_neg_rev%d" % rev] = a.numpy(), b.numpy()
    save("f11_gather_timestamp", **out)

    # ---- F13 dense-flow warp (lib/transforms/optic_flow.py) ---------------------------------------------
    H, Wd, n = 60, 80, 5000
    rng = np.random.default_rng(130)
    fx_, fy_, ft_ = (rng.uniform(-3, Wd + 2, n).astype(np.float32), rng.uniform(-3, H + 2, n).astype(np.float32),
                     np.sort(rng.uniform(0, 0.1, n)).astype(np.float32))
    flow = rng.normal(0, 30, size=(2, H, Wd)).astype(np.float32)
    xw, yw = ref.optic_flow.warp_events_flow_torch(torch.from_numpy(fx_), torch.from_numpy(fy_), torch.from_numpy(ft_),
                                                   torch.ones(n), torch.from_numpy(flow))
    xw2, yw2 = ref.optic_flow.warp_events_flow_torch(torch.from_numpy(fx_), torch.from_numpy(fy_), torch.from_numpy(ft_),
                                                     torch.ones(n), torch.from_numpy(flow), t0=0.02)
    save("f13_flow_warp", xs=fx_, ys=fy_, ts=ft_, flow=flow, xw=xw.numpy(), yw=yw.numpy(), xw_t0=xw2.numpy(),
         yw_t0=yw2.numpy(), versions=versions)

    # ---- F10 gaussian_filter (2-D and 3-D two-channel) ----------------------------------------------
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(100)
    a3 = rng.normal(size=(2, 37, 53)).astype(np.float32)
    out = dict(a3=a3, versions=versions)
    for s in (1.0, 2.0, 0.5):
        out["blur3_s%g" % s] = gaussian_filter(a3, s)
        out["blur2_s%g" % s] = gaussian_filter(a3[0], s)
    small = rng.normal(size=(3, 5)).astype(np.float32)    # axis shorter than the kernel radius
    out["small"] = small
    out["small_blur_s1"] = gaussian_filter(small, 1.0)
    save("f10_blur", **out)

    # ---- F8 variance objective: f and grad at several params / sigmas ------------------------------
    H, Wd, n = 180, 240, 30000
    x, y, t, p = gen_structured(80, n, H, Wd)
    params = np.array([[0, 0], [40, -25], [30, -20], [-10, 60], [100, 100]], dtype=np.float64)
    sig = np.array([0.0, 1.0, 2.0])
    out = dict(xs=x.astype(np.float32), ys=y.astype(np.float32), ts=t.astype(np.float32), ps=p.astype(np.int8),
               params=params, sigmas=sig, img_size=np.array([H, Wd]), versions=versions)
    obj = O.variance_objective()
    fv = np.zeros((len(params), len(sig)), dtype=np.float64)
    gv = np.zeros((len(params), len(sig), 2), dtype=np.float64)
    for i, prm in enumerate(params):
        for j, s in enumerate(sig):
            fv[i, j] = obj.evaluate_function(prm, x, y, t, p, w, (H, Wd), blur_sigma=s)
            gv[i, j] = obj.evaluate_gradient(prm, x, y, t, p, w, (H, Wd), blur_sigma=s)
    out["f"], out["grad"] = fv, gv
    # adaptive lifespan path (Q10)
    obj_al = O.variance_objective(adaptive_lifespan=True, minimum_events=5000)
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        obj_al.iter_update(np.array([400., -250.]))
        out["al_f"] = np.float64(obj_al.evaluate_function(np.array([40., -25.]), x, y, t, p, w, (H, Wd), blur_sigma=1.0))
        out["al_s_idx"] = np.int64(obj_al.s_idx)
        obj_al.iter_update(np.array([400., -250.]))
        out["al_grad"] = obj_al.evaluate_gradient(np.array([40., -25.]), x, y, t, p, w, (H, Wd), blur_sigma=1.0)
    save("f8_objective", **out)

    # ---- F12 the other objectives (reductions of the same IWE; "next" row) --------------------------
    params12 = np.array([[0, 0], [40, -25], [-10, 60]], dtype=np.float64)
    out = dict(params=params12, versions=versions)
    objs = {"sos": O.sos_objective(), "soe": O.soe_objective(), "moa": O.moa_objective(), "isoa": O.isoa_objective(),
            "sosa": O.sosa_objective()}
    for name, ob in objs.items():
        fvals, gvals = [], []
        for prm in params12:
            for s in (None, 0.0):
                fvals.append(np.float64(ob.evaluate_function(prm, x, y, t, p, w, (H, Wd), blur_sigma=s)))
                if ob.has_derivative:
                    if name == "sos":
                        ob.pixel_crossings = 5      # evaluate_gradient calls an undefined find_lifespan (:345): patch in
                        O.find_lifespan = lambda ts_, prm_, pc_: (None, None)
                    gvals.append(np.asarray(ob.evaluate_gradient(prm, x, y, t, p, w, (H, Wd), blur_sigma=s), dtype=np.float64))
        out[name + "_f"] = np.array(fvals)
        if gvals:
            out[name + "_g"] = np.array(gvals)
    r1 = O.r1_objective()
    out["r1_f"] = np.array([np.float64(r1.evaluate_function(prm, x, y, t, p, w, (H, Wd))) for prm in
                            (params12[0], params12[1], params12[1], params12[2])])
    save("f12_other_objectives", **out)

    # ---- F9 optimize trace ---------------------------------------------------------------------------
    out = dict(versions=versions)
    for mode in ("numeric", "analytic"):
        trace = []
        obj = O.variance_objective()
        f_orig, g_orig = obj.evaluate_function, obj.evaluate_gradient

        def f_rec(prm, *a, **k):
            v = f_orig(prm, *a, **k)
            trace.append(("f", np.array(prm, dtype=np.float64), np.float64(v), np.zeros(2)))
            return v

        def g_rec(prm, *a, **k):
            g = g_orig(prm, *a, **k)
            trace.append(("g", np.array(prm, dtype=np.float64), np.float64(0), np.array(g, dtype=np.float64)))
            return g
        obj.evaluate_function, obj.evaluate_gradient = f_rec, g_rec
        with contextlib.redirect_stdout(io.StringIO()):
            argmax = C.optimize_contrast(x, y, t, p, w, obj, numeric_grads=(mode == "numeric"),
                                         blur_sigma=1.0, img_size=(H, Wd))
        out[mode + "_argmax"] = np.array(argmax)
        out[mode + "_kind"] = np.array([k for k, _, _, _ in trace])
        out[mode + "_params"] = np.array([q for _, q, _, _ in trace])
        out[mode + "_f"] = np.array([v for _, _, v, _ in trace])
        out[mode + "_g"] = np.array([g for _, _, _, g in trace])
        print("  optimize[%s]: %d evals, argmax=%s" % (mode, len(trace), argmax))
    save("f9_optimize_trace", **out)


if __name__ == "__main__":
    main()
