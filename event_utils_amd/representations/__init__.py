from .image import (events_to_image, events_to_image_torch, interpolate_to_image,  # noqa: F401
                    interpolate_to_derivative_img, events_to_image_drv, image_to_event_weights,
                    events_to_timestamp_image, events_to_timestamp_image_torch, TimestampImage, EventImage)
from .voxel_grid import (events_to_voxel, events_to_voxel_torch, events_to_neg_pos_voxel,  # noqa: F401
                         events_to_neg_pos_voxel_torch, voxel_grids_fixed_n_torch, voxel_grids_fixed_t_torch,
                         events_to_voxel_timesync_torch)
