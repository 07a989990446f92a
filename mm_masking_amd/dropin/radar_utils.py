"""Drop-in for the reference's flat module name (`from radar_utils import ...`)."""
from mm_masking_amd.radar_utils import *  # noqa: F401,F403
from mm_masking_amd.radar_utils import __all__  # noqa: F401
