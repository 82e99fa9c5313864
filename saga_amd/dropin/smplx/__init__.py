"""Put ``saga_amd/dropin`` on PYTHONPATH and the reference's ``import smplx`` resolves here."""
from saga_amd.smplx import *          # noqa: F401,F403
from saga_amd.smplx import lbs        # noqa: F401
