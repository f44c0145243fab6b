"""Reference: lib/transforms/optic_flow.py."""
import torch

from .. import _device as D
from .. import _lib


def warp_events_flow_torch(xt, yt, tt, pt, flow_field, t0=None, batched=False, batch_indices=None):
    """
    Warp events by a dense per-pixel flow field (reference: optic_flow.py:5-46): the (2, H, W) field is sampled
    bilinearly at every event (F.grid_sample, align_corners=True, zero padding) and x' = x + flow_x * (t - t0),
    y' = y + flow_y * (t - t0), t0 defaulting to the last timestamp.  Tensors in, float32 tensors out on xt.device.
    (`batched` / `batch_indices` are accepted and ignored, as upstream.)
    """
    device = xt.device
    if len(xt.shape) > 1:
        xt, yt, tt, pt = xt.squeeze(), yt.squeeze(), tt.squeeze(), pt.squeeze()
    dev = D.require_gpu()
    xd, yd, td = (D.to_device(a, torch.float32, dev) for a in (xt, yt, tt))
    if t0 is None:
        t0 = td[-1].item()
    flow = D.to_device(flow_field, torch.float32, dev)
    H, W = flow.shape[-2:]
    flow = flow.reshape(-1, H, W)
    if flow.shape[0] != 2:
        raise ValueError("flow_field must hold 2 channels (x and y flow), got shape %s" % (tuple(flow_field.shape),))
    xo, yo = torch.empty_like(xd), torch.empty_like(yd)
    _lib.call("evk_warp_flow_field_f32", D.ptr(xd), D.ptr(yd), D.ptr(td), xd.shape[0], D.ptr(flow), H, W, float(t0),
              D.ptr(xo), D.ptr(yo), D.stream())
    return xo.to(device), yo.to(device)
