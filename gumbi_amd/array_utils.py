"""Helpers that build parrays for the model-declaration knobs of ``GP.fit``.

``make_deltas_parray`` is the counterpart of ``gumbi/array_utils.py:36-126``: it turns per-dimension
*differences* ("this input varies on scales of at least 0.5 standard deviations", "... of at most 40 hp")
into the standardized ``(lower, upper)`` lengthscale bounds that ``fit(ls_bounds=...)`` /
``build_model(ls_bounds=...)`` consume (``gumbi/regression/pymc/GP.py:630-650``: the bounds replace the
pairwise-distance defaults of ``parse_ls_limits`` before the InverseGamma prior is fitted).  The
stacking helpers of the reference's module are convenience wrappers outside the fit / predict path
 and are not restated.
"""

from __future__ import annotations

import numpy as np

from .arrays import ParameterArray
from .utils.misc import assert_in

__all__ = ["make_deltas_parray"]


def _standardized_step(stdzr, dim, delta, scale):
    """Standardized distance between the points ``delta`` and ``2 * delta`` of ``dim``'s own axis, ``delta``
    being given on ``scale``.  For an untransformed variable on the natural scale this is ``delta / sigma``;
    for a log variable it is ``log(2) / sigma`` whatever ``delta`` -- the reference measures the step
    between v and 2v after mapping both to the natural scale (``array_utils.py:15-30``), and so does this."""
    pair = np.array([delta, 2.0 * delta], dtype=float)
    if scale == "transformed":
        pair = np.asarray(stdzr.untransform(dim, pair), dtype=float)
    elif scale == "standardized":
        pair = np.asarray(stdzr.unstdz(dim, pair), dtype=float)
    z = np.asarray(stdzr.stdz(dim, pair), dtype=float)
    return z[1:] - z[:-1]


def make_deltas_parray(*, stdzr, scale, **deltas):
    """A standardized parray of lengthscale bounds, one ``[lower, upper]`` pair per named dimension.

    ``deltas``: ``dim=[lower, upper]`` with either entry ``None`` for "keep the default" (NaN in the result,
    which ``_prepare_lengthscales`` turns back into ``None``).  ``scale`` says on which of the variable's three
    scales -- ``"natural"``, ``"transformed"`` or ``"standardized"`` -- the numbers are given.
    """
    assert_in("scale", scale, ["transformed", "standardized", "natural"])
    cols = {}
    for dim, bounds in deltas.items():
        cols[dim] = [np.array([np.nan]) if b is None else _standardized_step(stdzr, dim, b, scale) for b in bounds]
    return ParameterArray(**cols, stdzr=stdzr, stdzd=True)
