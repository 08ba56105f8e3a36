"""Lengthscale-prior construction from pairwise-distance extrema.

Counterpart of ``gumbi/utils/gp_utils.py`` (``parse_ls_limits`` :15-48, ``get_ls_prior`` :51-87).
Same rule -- ``lower = max(user, min non-zero pairwise distance, 0.01)``, ``upper = max pairwise
distance`` per group, then an ``InverseGamma(alpha, beta)`` holding ``mass`` of its probability in
``[lower, upper]`` -- but not the reference's algorithms, which allocate all N(N-1)/2 distances
per dimension (40 GB at N = 1e5) and sum them in pure Python:

* ARD (one group per column, 1-D distances): the extrema of |x_i - x_j| are the smallest
  positive gap between sorted neighbours and ``max - min`` -- exact, O(N log N), host;
* joint (``ARD=False``): a true d-dimensional pairwise reduction, done on the GPU by
  ``gmb_ls_limits`` (``covariance.hpp: ls_limits_kernel``).

``find_constrained_prior`` restates ``pm.find_constrained_prior`` (call site ``gp_utils.py:64-70``):
PyMC minimises the squared distance of ``cdf(lower)`` from ``(1-mass)/2`` subject to
``cdf(upper) - cdf(lower) = mass``; for the two-parameter InverseGamma the optimum is the unique
root of ``cdf(lower) = (1-mass)/2, cdf(upper) = (1+mass)/2``, which is solved for directly here.
"""

from __future__ import annotations

from warnings import warn

import numpy as np
from scipy.optimize import brentq
from scipy.special import gammaincc, gammainccinv

from .misc import listify

__all__ = ["parse_ls_limits", "get_ls_prior", "find_constrained_prior"]


def _broadcast_bounds(bound, n, label):
    vals = [None] if bound is None else listify(bound)
    if len(vals) == 1:
        vals = vals * n
    if len(vals) != n:
        raise ValueError(f"Number of {label} bounds must match number of dimensions")
    return list(vals)


def _extrema_1d(x):
    s = np.sort(np.asarray(x, dtype=float).ravel())
    gaps = np.diff(s)
    pos = gaps[gaps > 0]
    if pos.size == 0:
        return None, None
    return float(pos.min()), float(s[-1] - s[0])


def parse_ls_limits(X, *, ARD, lower=None, upper=None, device=0):
    """Per-group ``(lowers, uppers)`` for the lengthscale prior (reference ``gp_utils.py:15-48``)."""
    X = np.asarray(X, dtype=float)
    if X.ndim == 1:
        X = X[:, None]
    n_groups = X.shape[1] if ARD else 1
    lowers = _broadcast_bounds(lower, n_groups, "lower")
    uppers = _broadcast_bounds(upper, n_groups, "upper")

    if ARD or X.shape[1] == 1:
        extrema = [_extrema_1d(X[:, j]) for j in range(n_groups)]
    else:
        from ..engine import ls_limits as _gpu_ls_limits

        lo, hi = _gpu_ls_limits(X, ard=False, device=device)
        extrema = [(None, None) if lo[0] < 0 else (float(lo[0]), float(hi[0]))]

    for g, (dmin, dmax) in enumerate(extrema):
        default_lower = dmin if dmin is not None else 0.01
        lo = default_lower if lowers[g] is None else lowers[g]
        lowers[g] = max(lo, default_lower, 0.01)
        if uppers[g] is None:
            uppers[g] = dmax if dmax is not None else 1
    return lowers, uppers


def find_constrained_prior(lower, upper, mass=0.98):
    """``{"alpha", "beta"}`` of the InverseGamma with ``P(l < lower) = P(l > upper) = (1-mass)/2``.

    CDF of InverseGamma(a, b) at x is Q(a, b/x) (regularised upper incomplete gamma).  For a fixed
    ``a`` the lower condition gives ``b = lower * Qinv(a, q_lo)``; the upper condition is then a
    monotone 1-D root in ``a``.  Raises ``ValueError`` (as PyMC does) if no root is bracketed.
    """
    lower, upper = float(lower), float(upper)
    if not (0 < lower < upper):
        raise ValueError(f"need 0 < lower < upper, got {lower}, {upper}")
    q_lo, q_hi = (1.0 - mass) / 2.0, (1.0 + mass) / 2.0

    def beta_of(a):
        return lower * gammainccinv(a, q_lo)

    def resid(log_a):
        a = np.exp(log_a)
        return gammaincc(a, beta_of(a) / upper) - q_hi

    lo_a, hi_a = np.log(1e-3), np.log(1e6)
    f_lo, f_hi = resid(lo_a), resid(hi_a)
    if not (np.isfinite(f_lo) and np.isfinite(f_hi)) or f_lo * f_hi > 0:
        raise ValueError("Optimization of parameters failed: no InverseGamma matches the requested mass")
    a = float(np.exp(brentq(resid, lo_a, hi_a, xtol=1e-14, rtol=1e-13, maxiter=500)))
    return {"alpha": a, "beta": float(beta_of(a))}


def get_ls_prior(X, *, ARD, lower=None, upper=None, mass=0.98, dist="InverseGamma", device=0):
    """``{"alpha": [...], "beta": [...]}`` per lengthscale (reference ``gp_utils.py:51-87``),
    lowering ``mass`` by 0.01 and warning whenever a fit fails, as the reference does."""
    if dist != "InverseGamma":
        raise NotImplementedError("only the InverseGamma lengthscale prior is implemented")
    lowers, uppers = parse_ls_limits(X, ARD=ARD, lower=lower, upper=upper, device=device)
    params = []
    for i, (lo, up) in enumerate(zip(lowers, uppers)):
        mass_ = mass
        while True:
            try:
                fitted = find_constrained_prior(lo, up, mass=mass_)
                break
            except ValueError as err:
                if "Optimization of parameters failed" not in str(err) or mass_ <= 0.5:
                    raise
                mass_ -= 0.01
        if mass_ != mass:
            warn(f"Mass of constrained lengthscale prior was reduced from {mass:.3f} to {mass_:.3f} "
                 f"to enable convergence for dimension {i}.")
        params.append(fitted)
    return {k: [p[k] for p in params] for k in ("alpha", "beta")}
