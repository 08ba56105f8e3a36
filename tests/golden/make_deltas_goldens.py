"""Goldens for ``make_deltas_parray`` and the ``ls_bounds`` -> ``parse_ls_limits`` chain, captured from the
REFERENCE (``gumbi/array_utils.py:8-33``, ``gumbi/regression/pymc/GP.py:630-650``,
``gumbi/utils/gp_utils.py:15-48``) executed in the build container with its third-party imports stubbed
(see make_plumbing_goldens.py).  Only arrays are written (``deltas_goldens.npz``).

Usage:  python tests/golden/make_deltas_goldens.py
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from make_plumbing_goldens import EXAMPLE_STDZR, LOG_VARS, LOGIT_VARS, import_reference  # noqa: E402

# (scale, {dim: [lower, upper]})
CASES = {
    "standardized_XY": ("standardized", {"X": [0.5, None], "Y": [0.25, 6.0]}),
    # natural-scale deltas of UNtransformed variables raise a TypeError in the reference (list - float in
    # aggregation.py:393); only log / logit variables can be captured on that scale
    "natural_log_logit": ("natural", {"Y": [20.0, None], "X": [0.05, 0.3], "d": [None, 0.5]}),
    "transformed_mix": ("transformed", {"X": [0.2, 1.5], "Y": [0.1, 2.0], "d": [0.5, 3.0]}),
    "standardized_plain": ("standardized", {"lg10_Z": [0.5, 3.0], "a": [None, 2.0]}),
}


def main():
    gumbi = import_reference()
    from gumbi.utils.gp_utils import parse_ls_limits

    stdzr = gumbi.Standardizer(**EXAMPLE_STDZR, log_vars=LOG_VARS, logit_vars=LOGIT_VARS)
    out = {}
    rng = np.random.default_rng(7)
    pts = rng.standard_normal((40, 3))
    out["points"] = pts
    for case, (scale, deltas) in CASES.items():
        pa = gumbi.make_deltas_parray(stdzr=stdzr, scale=scale, **deltas)
        out[f"{case}/shape"] = np.array(pa.shape)
        for dim in deltas:
            out[f"{case}/{dim}"] = np.asarray(pa[dim].values(), dtype=float)
        # the bound extraction of PymcGP._prepare_lengthscales (pymc/GP.py:633-641) followed by parse_ls_limits
        zb = [[b if not np.isnan(b) else None for b in pa[dim].z.values().squeeze()] for dim in deltas]
        lower, upper = list(zip(*zb))
        lo, up = parse_ls_limits(pts[:, : len(deltas)], ARD=True, lower=list(lower), upper=list(upper))
        out[f"{case}/ls_lower"] = np.array(lo, dtype=float)
        out[f"{case}/ls_upper"] = np.array(up, dtype=float)
    np.savez_compressed(HERE / "deltas_goldens.npz", **out)
    print(f"wrote {len(out)} arrays")


if __name__ == "__main__":
    main()
