from .events_cmax import *  # noqa: F401,F403
from .warps import *  # noqa: F401,F403
from .objectives import *  # noqa: F401,F403
