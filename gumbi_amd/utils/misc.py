"""Small argument-handling helpers used by the data containers and the Regressor.

Counterparts of the one-liners the hot path touches in ``gumbi/utils/misc.py``
(``listify`` :33-45, ``first`` :23-25, ``skip`` :79-81, ``assert_in`` / ``assert_is_subset``).
"""

from collections.abc import Iterable, Iterator

import numpy as np

__all__ = ["listify", "first", "identity", "skip", "assert_in", "assert_is_subset"]


def listify(x):
    """``None`` -> ``[]``, a string or scalar -> one-element list, any other iterable -> list."""
    if x is None:
        return []
    if isinstance(x, list):
        return x
    if isinstance(x, (str, bytes)):
        return [x]
    if isinstance(x, (Iterable, Iterator)):
        return list(x)
    return [x]


def first(seq):
    return listify(seq)[0]


def identity(x):
    """The "no transform" transform."""
    return x


skip = identity  # reference spelling


def assert_in(label, value, allowed):
    """Raise ``ValueError`` unless ``value`` is one of ``allowed``."""
    allowed = listify(allowed)
    if value not in allowed:
        raise ValueError(f"{label} must be one of {allowed}, got {value!r}")


def assert_is_subset(label, subset, superset):
    """Raise ``ValueError`` unless every element of ``subset`` is found in ``superset``.

    Set membership, as the reference does (``gumbi/utils/misc.py:100-107``): ``specify_model`` calls
    this with the N observed values of every continuous dimension, so a list scan would be O(N^2)."""
    sub = listify(subset)
    sup = superset
    if hasattr(sup, "tolist") and not isinstance(sup, (list, tuple, set, frozenset, dict)):
        sup = sup.tolist()  # numpy array / pandas Index / Series -> python scalars
    try:
        have = set(sup)
        missing = [item for item in sub if item not in have]
    except TypeError:  # unhashable elements: fall back to equality scans
        sup = listify(sup)
        missing = [item for item in sub if item not in sup]
    if missing:
        raise ValueError(f"{label} {missing} not found among the allowed values")
