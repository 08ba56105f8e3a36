"""gumbi_amd -- MI355X-native Gaussian-process regression behind Gumbi's ``GP`` API.

``GP(ds).fit(...)``, ``prepare_grid()``, ``predict_grid()`` and the ``DataSet`` / ``parray`` /
``uparray`` plumbing keep the reference's names and behaviour (JohnGoertz/Gumbi v0.4.1); the
covariance build, Cholesky factorisation and posterior solves run in ``libgumbi_hip.so``
(hand-written HIP for gfx950) through a ctypes C ABI (``include/gumbi_hip.h``).
"""

from .aggregation import DataSet, Standardizer, TidyData, WideData
from .array_utils import make_deltas_parray
from .arrays import (
    LayeredArray,
    MVUncertainParameterArray,
    ParameterArray,
    UncertainArray,
    UncertainParameterArray,
)
from .regression import GP, HipGP, Regressor

__version__ = "0.1.0"

# Aliases (reference gumbi/__init__.py:13-17)
parray = ParameterArray
uarray = UncertainArray
uparray = UncertainParameterArray
mvuparray = MVUncertainParameterArray

__all__ = [
    "DataSet", "Standardizer", "TidyData", "WideData", "LayeredArray", "ParameterArray", "UncertainArray",
    "UncertainParameterArray", "MVUncertainParameterArray", "GP", "HipGP", "Regressor", "parray", "uarray",
    "uparray", "mvuparray", "make_deltas_parray",
]
