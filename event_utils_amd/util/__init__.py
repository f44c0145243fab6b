from .event_util import events_bounds_mask  # noqa: F401
