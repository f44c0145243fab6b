"""
TEST INFRASTRUCTURE ONLY -- never imported by the product package.

The reference's TORCH path restated with the same torch primitives, on the CPU, all host threads: what
"the reference on this box's host cores" costs when it is allowed to use every core ("a multi-threaded
variant (torch CPU index_put_, threads = all cores) as the best-effort CPU").  Used by bench.py's cpu_baseline leg and
pinned to the golden vectors in tests/test_oracle_golden.py; the numpy restatement (reference_np.py) stays THE oracle.
"""
import torch


def events_to_image_torch_nearest(xs, ys, ps, sensor_size):
    """image.py:46-100, nearest branch with clip_out_of_range=False (the voxel path's call, voxel_grid.py:140-142):
    float coordinates truncated toward zero (.long(), :90-91), img.index_put_((ys, xs), ps, accumulate=True) (:95)."""
    img = torch.zeros(tuple(int(v) for v in sensor_size), dtype=torch.float32)
    img.index_put_((ys.long(), xs.long()), ps, accumulate=True)
    return img


def events_to_voxel_torch(xs, ys, ts, ps, B, sensor_size=(180, 240)):
    """voxel_grid.py:114-153 (temporal_bilinear=True): B passes, each a full nearest-pixel accumulate of
    ps * max(0, 1 - |t_norm - b|) with t_norm = (ts - ts[0]) / (ts[-1] - ts[0]) * (B - 1) in float32."""
    dt = ts[-1] - ts[0]
    t_norm = (ts - ts[0]) / dt * (B - 1)
    zeros = torch.zeros(t_norm.size())
    bins = []
    for bi in range(B):
        weights = ps * torch.max(zeros, 1.0 - torch.abs(t_norm - bi))
        bins.append(events_to_image_torch_nearest(xs, ys, weights, sensor_size))
    return torch.stack(bins)


# ---- contrast maximisation: the reference's own mix of numpy (warp, mask), torch (splat) and scipy (blur) ------------
import numpy as np  # noqa: E402
from scipy.ndimage import gaussian_filter  # noqa: E402


def _splat(pxs, pys, dxs, dys, weights, img):
    """interpolate_to_image, image.py:102-115: four index_put_(accumulate=True) passes."""
    img.index_put_((pys, pxs), weights * (1.0 - dxs) * (1.0 - dys), accumulate=True)
    img.index_put_((pys, pxs + 1), weights * dxs * (1.0 - dys), accumulate=True)
    img.index_put_((pys + 1, pxs), weights * (1.0 - dxs) * dys, accumulate=True)
    img.index_put_((pys + 1, pxs + 1), weights * dxs * dys, accumulate=True)


def _splat_drv(pxs, pys, dxs, dys, d_img, w1, w2):
    """interpolate_to_derivative_img, image.py:117-136: eight index_put_ passes."""
    for i in range(d_img.shape[0]):
        d_img[i].index_put_((pys, pxs), w1[i] * (-(1.0 - dys)) + w2[i] * (-(1.0 - dxs)), accumulate=True)
        d_img[i].index_put_((pys, pxs + 1), w1[i] * (1.0 - dys) + w2[i] * (-dxs), accumulate=True)
        d_img[i].index_put_((pys + 1, pxs), w1[i] * (-dys) + w2[i] * (1.0 - dxs), accumulate=True)
        d_img[i].index_put_((pys + 1, pxs + 1), w1[i] * dys + w2[i] * dxs, accumulate=True)


def get_iwe(params, xs, ys, ts, ps, img_size, sensor_size, compute_gradient=False):
    """objectives.py:165-199 for linvel_warp (warps.py:51-61), with an explicit sensor_size:
    float64 numpy warp and bounds mask (event_util.py:15-28), then events_to_image_drv (image.py:162-217)."""
    dt = ts - ts[-1]
    x, y = xs - dt * params[0], ys - dt * params[1]
    mask = np.where((x <= 0) | (x > img_size[1]), 0.0, 1.0) * np.where((y <= 0) | (y > img_size[0]), 0.0, 1.0)
    x, y, p = x * mask, y * mask, ps * mask
    xt, yt, pt = (torch.from_numpy(a).float() for a in (x, y, p))
    size = (int(sensor_size[0]) + 1, int(sensor_size[1]) + 1)
    m = torch.where(xt >= size[1] - 1, 0.0, 1.0) * torch.where(yt >= size[0] - 1, 0.0, 1.0)
    pxs, pys = xt.floor(), yt.floor()
    dxs, dys = xt - pxs, yt - pys
    pxs, pys, mp = (pxs * m).long(), (pys * m).long(), pt * m
    img = torch.zeros(size)
    _splat(pxs, pys, dxs, dys, mp, img)
    d_img = None
    if compute_gradient:
        jx = np.zeros((2, len(xs))); jy = np.zeros((2, len(xs)))
        jx[0], jy[1] = -dt * mask, -dt * mask
        d_img = torch.zeros((2, *size))
        _splat_drv(pxs, pys, dxs, dys, d_img, torch.from_numpy(jx).float() * mp, torch.from_numpy(jy).float() * mp)
        d_img = d_img.numpy()
    return img.numpy(), d_img


def variance_f(params, xs, ys, ts, ps, img_size, sensor_size, blur_sigma=1.0):
    """variance_objective.evaluate_function, objectives.py:211-236."""
    iwe, _ = get_iwe(params, xs, ys, ts, ps, img_size, sensor_size, False)
    if blur_sigma > 0:
        iwe = gaussian_filter(iwe, blur_sigma)
    return -np.var(iwe - np.mean(iwe))


def variance_grad(params, xs, ys, ts, ps, img_size, sensor_size, blur_sigma=1.0):
    """variance_objective.evaluate_gradient, objectives.py:238-264 (3-D blur of dIWE, un-blurred IWE: Q4, Q5)."""
    iwe, d_iwe = get_iwe(params, xs, ys, ts, ps, img_size, sensor_size, True)
    if blur_sigma > 0:
        d_iwe = gaussian_filter(d_iwe, blur_sigma)
    c = 2.0 * (iwe - np.mean(iwe))
    return -np.array([np.mean(c * d_iwe[k]) for k in range(2)])
