from .base import Regressor
from .hip_gp import HipGP
from .hip_gp import HipGP as GP

__all__ = ["Regressor", "GP", "HipGP"]
