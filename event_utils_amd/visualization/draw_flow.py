"""Reference: lib/visualization/draw_flow.py -- only the array computation of motion_compensate (:15-26); the plotting
functions of that file are out of scope."""
import numpy as np
import torch

from ..representations.image import events_to_image_torch
from ..transforms.optic_flow import warp_events_flow_torch


def motion_compensate(xs, ys, ts, ps, flow, fname=None, crop=None):
    """
    Motion-compensated event image for a dense flow field (reference: draw_flow.py:15-26): numpy events and a
    (2, H, W) / (H, W, 2)-free flow array -> warp every event by the flow sampled at its position
    (warp_events_flow_torch, :18) -> bilinear event image of the flow's size (:21) -> flip both axes (:22) -> min-max
    normalise to [0, 255] (:23, cv.normalize NORM_MINMAX: (img - min) * 255 / (max - min), float32) -> crop
    [y0:y1, x0:x1] (:24-25).  Both per-event steps run in libevk.so.  Returns the float32 image; the reference writes it
    to `fname` with OpenCV and returns nothing -- with fname given and cv2 importable it is written here as well.
    """
    xt, yt, tt, pt, ft = (torch.from_numpy(np.asarray(a)).type(torch.float32) for a in (xs, ys, ts, ps, flow))
    xw, yw = warp_events_flow_torch(xt, yt, tt, pt, ft)
    img_size = list(ft.shape)
    img_size.remove(2)
    img = events_to_image_torch(xw, yw, pt, sensor_size=img_size, interpolation='bilinear')
    img = np.flip(np.flip(img.numpy(), axis=0), axis=1)
    lo, hi = np.float32(img.min()), np.float32(img.max())
    scale = np.float32(255.0) / (hi - lo) if hi > lo else np.float32(0.0)
    img = ((img - lo) * scale).astype(np.float32)
    if crop is not None:
        img = img[crop[0]:crop[1], crop[2]:crop[3]]
    if fname is not None:
        try:
            import cv2 as cv
            cv.imwrite(fname, img)
        except ImportError:
            pass
    return img
