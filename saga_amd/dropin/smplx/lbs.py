from saga_amd.smplx.lbs import *      # noqa: F401,F403
from saga_amd.smplx.lbs import batch_rodrigues  # noqa: F401
