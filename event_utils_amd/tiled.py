"""Kernel-variant dispatch for the two streaming hot paths (voxel grid, fused linear-flow IWE).
impl: 'direct' = one global atomic per contribution (evk_scatter.hip);
      'tiled'  = tile-bucketed events + per-tile LDS accumulate + one flush (evk_tiled.hip).
Default comes from EVK_IMPL (env) or 'auto'."""
import os

from . import _device as D
from . import _lib


def default_impl():
    return os.environ.get("EVK_IMPL", "auto")


def voxel_f32(xd, yd, td, pd, t_first, t_last, B, H, W, out, oob=None, impl=None):
    impl = impl or default_impl()
    _lib.call("evk_voxel_f32", D.ptr(xd), D.ptr(yd), D.ptr(td), D.ptr(pd), xd.shape[0], t_first, t_last, B, H, W,
              D.ptr(out), oob.ptr if oob is not None else None, D.stream())
    return out


def iwe_linvel(ev, t_ref, vx, vy, bounds_w, bounds_h, ch, cw, flags, iwe, diwe, impl=None):
    """Fused get_iwe for the linear-flow model on DeviceEvents `ev`, accumulating into iwe (ch, cw) / diwe (2, ch, cw)."""
    import torch
    impl = impl or default_impl()
    fn = "evk_iwe_linvel_f32" if ev.dtype == torch.float32 else "evk_iwe_linvel_f64"
    _lib.call(fn, D.ptr(ev.x), D.ptr(ev.y), D.ptr(ev.t), D.ptr(ev.p), len(ev), t_ref, vx, vy, bounds_w, bounds_h, ch, cw,
              flags, float(ev.p_scale), D.ptr(iwe), D.ptr(diwe), D.stream())


def _time_ms(fn, reps):
    import torch
    fn()
    torch.cuda.synchronize()
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    for i in range(reps):
        e0[i].record()
        fn()
        e1[i].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in zip(e0, e1))
    return float(sum(ts) / len(ts))


def time_voxel_kernels(xd, yd, td, pd, t_first, t_last, B, H, W, impl=None, reps=10):
    """HIP-event timing (on the launch stream) of the kernels one voxel call launches, for bench.py's roofline."""
    import torch
    impl = impl or default_impl()
    out = torch.zeros((B, H, W), dtype=torch.float32, device=xd.device)
    ms = _time_ms(lambda: voxel_f32(xd, yd, td, pd, t_first, t_last, B, H, W, out, None, impl="direct"), reps)
    return {"impl": "direct", "dominant": "k_voxel_f32<VEC>", "dominant_ms": ms, "total_ms": ms,
            "kernels_ms": {"k_voxel_f32": round(ms, 4)}}
