"""Put ``saga_amd/dropin`` on PYTHONPATH and the reference's ``import chamfer_distance as chd`` resolves here."""
from saga_amd.chamfer_distance import *   # noqa: F401,F403
