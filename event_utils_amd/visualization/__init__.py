from .draw_flow import motion_compensate  # noqa: F401
