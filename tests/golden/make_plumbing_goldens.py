"""Capture plumbing goldens from the REFERENCE implementation (runs only in the build container).

The reference (JohnGoertz/Gumbi, mounted read-only at /root/reference) is pure Python but imports
pymc / pytensor / gpytorch / seaborn / uncertainties at module import, none of which is installed
here.  Its data plumbing (DataSet, Standardizer, parray, uparray, Regressor.specify_model /
get_shaped_data / prepare_grid / _prepare_points_for_prediction, gp_utils.parse_ls_limits) does
not touch those packages, so they are stubbed in ``sys.modules`` and the plumbing is executed as
is.  Only its *outputs* (arrays) are written, to ``tests/golden/plumbing_goldens.npz`` +
``plumbing_goldens.json``; no reference source enters this repository.

Usage:  python tests/golden/make_plumbing_goldens.py
"""
import json
import sys
import types
from pathlib import Path

import numpy as np
import pandas as pd

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")


def _stub(name, attrs=()):
    mod = types.ModuleType(name)
    for attr in attrs:
        setattr(mod, attr, type(attr, (), {}))
    sys.modules[name] = mod
    return mod


def import_reference():
    _stub("pymc")
    pt = _stub("pytensor")
    pt.tensor = _stub("pytensor.tensor")
    _stub("gpytorch")
    _stub("gpytorch.priors")
    _stub("gpytorch.priors.prior", ["Prior"])
    utils = _stub("gpytorch.priors.utils")
    utils._bufferize_attributes = lambda *a, **k: None
    utils._del_attributes = lambda *a, **k: None
    _stub("seaborn")
    unc = _stub("uncertainties")
    unc.unumpy = _stub("uncertainties.unumpy")
    sys.path.insert(0, str(REF))
    import gumbi  # noqa: E402

    return gumbi


EXAMPLE_STDZR = {
    "a": {"μ": -0.762, "σ2": 1.258**2}, "b": {"μ": -0.0368, "σ2": 0.351**2}, "c": {"μ": -5.30, "σ2": 0.582**2},
    "d": {"μ": -0.307, "σ2": 0.158**2}, "e": {"μ": -1.056, "σ2": 0.398**2}, "f": {"μ": 3.34, "σ2": 0.1501**2},
    "X": {"μ": -0.282, "σ2": 1**2}, "Y": {"μ": 4.48, "σ2": 0.75**2}, "lg10_Z": {"μ": 5, "σ2": 2**2},
}
LOG_VARS = ["d", "f", "b", "c", "Y"]
LOGIT_VARS = ["e", "X"]

# the five parametrisations of tests/test_regression.py:125-143 (+ the "fit_simple" one, :173-176)
CASES = {
    "multi_output": {"outputs": ["d", "c"], "continuous_dims": ["X", "Y"]},
    "categorical_code": {"continuous_dims": ["X", "Y"], "categorical_dims": "Code"},
    "name_as_continuous": {"continuous_dims": ["X", "Y", "Name"]},
    "three_continuous": {"continuous_dims": ["X", "Y", "lg10_Z"]},
    "single_name": {"continuous_dims": ["X", "Y", "Name"], "continuous_levels": {"Name": ["intense-opportunity"]}},
    "fit_simple": {"continuous_dims": ["X", "Y", "lg10_Z"], "continuous_levels": {"lg10_Z": [8]}},
}


def main():
    gumbi = import_reference()
    from gumbi.regression.base import Regressor
    from gumbi.utils.gp_utils import parse_ls_limits

    class Probe(Regressor):  # concrete shell around the reference's plumbing
        def fit(self, *a, **k):
            pass

        def build_model(self, *a, **k):
            pass

        def predict(self, points_array, with_noise=True, **k):
            raise RuntimeError

    es = pd.read_pickle(REF / "tests/test_data/test_dataset.pkl")
    arrays, meta = {}, {}
    for case, kwargs in CASES.items():
        stdzr = gumbi.Standardizer(**EXAMPLE_STDZR, log_vars=LOG_VARS, logit_vars=LOGIT_VARS)
        ds = gumbi.DataSet.from_tidy(es, names_column="Parameter", stdzr=stdzr)
        gp = Probe(ds, outputs="d")
        gp.specify_model(**{k: (dict(v) if isinstance(v, dict) else v) for k, v in kwargs.items()})
        X, y = gp.get_shaped_data("mean")
        arrays[f"{case}/X"] = X
        arrays[f"{case}/y"] = y
        meta[case] = {
            "kwargs": kwargs,
            "dims": gp.dims,
            "continuous_dims": gp.continuous_dims,
            "categorical_dims": gp.categorical_dims,
            "filter_dims": {k: [str(x) if isinstance(x, str) else float(x) for x in v]
                            for k, v in gp.filter_dims.items()},
            "categorical_coords": {d: {str(k): int(v) for k, v in c.items()} for d, c in gp.categorical_coords.items()},
            "n_levels": {d: len(v) for d, v in gp.levels.items()},
        }
        idx_s = [gp.dims.index(d) for d in gp.continuous_dims]
        for ard in (True, False):
            lo, up = parse_ls_limits(X[:, idx_s], ARD=ard)
            arrays[f"{case}/ls_lower_ard{int(ard)}"] = np.array(lo, dtype=float)
            arrays[f"{case}/ls_upper_ard{int(ard)}"] = np.array(up, dtype=float)
        # grid + points handed to the backend
        gp.prepare_grid(resolution=4)
        grid = gp.grid_points
        for name in grid.names:
            arrays[f"{case}/grid/{name}"] = np.asarray(grid[name].values(), dtype=float)
        meta[case]["grid_shape"] = list(gp.grid_parray.shape)
        meta[case]["grid_names"] = list(grid.names)
        cat_levels = None
        if [d for d in gp.categorical_dims if d != gp.out_col]:
            cat_levels = {d: gp.categorical_levels[d][0] for d in gp.categorical_dims if d != gp.out_col}
        points = gp.append_categorical_points(grid, categorical_levels=cat_levels) if gp.categorical_dims else grid
        output = gp._parse_prediction_output(None)
        pa, tall, coords = gp._prepare_points_for_prediction(points, output=output)
        arrays[f"{case}/points_array"] = np.asarray(pa, dtype=float)
        meta[case]["predict_output"] = list(output)
        meta[case]["cat_levels"] = cat_levels

    # un-standardisation of predictions: uparray(name, mu_z, var_z, stdzd=True)   (base.py:578-580)
    stdzr = gumbi.Standardizer(**EXAMPLE_STDZR, log_vars=LOG_VARS, logit_vars=LOGIT_VARS)
    mu_z = np.linspace(-2.0, 2.0, 9)
    var_z = np.linspace(0.05, 1.5, 9)
    arrays["unstdz/mu_z"], arrays["unstdz/var_z"] = mu_z, var_z
    for name in ["d", "e", "a", "not_in_stdzr"]:
        upa = gumbi.uparray(name, mu_z, var_z, stdzr=stdzr, stdzd=True)
        arrays[f"unstdz/{name}/mu"] = np.asarray(upa.μ)
        arrays[f"unstdz/{name}/var"] = np.asarray(upa.σ2)
    # standardisation of inputs: parray(...).z
    for name, vals in {"X": np.linspace(0.05, 0.95, 7), "Y": np.geomspace(10, 800, 7), "lg10_Z": np.linspace(2, 9, 7)}.items():
        arrays[f"stdz/{name}/nat"] = vals
        arrays[f"stdz/{name}/z"] = gumbi.parray(**{name: vals}, stdzr=stdzr).z.values()

    # Standardizer.from_DataFrame on the second fixture (tests/test_aggregation.py:71-80)
    df = pd.read_pickle(REF / "tests/test_data/estimates_test_data.pkl")
    ds2 = gumbi.DataSet.from_tidy(df, names_column="Parameter", log_vars=["Y", "c", "b"], logit_vars=["X", "e"])
    meta["estimates_stdzr"] = {k: {"μ": float(v["μ"]), "σ2": float(v["σ2"])} for k, v in ds2.stdzr.items()}
    meta["estimates_outputs"] = list(ds2.outputs)
    meta["estimates_wide_columns"] = list(ds2.wide.columns)
    tz = ds2.tidy.z
    meta["estimates_tidy_z_mean"] = {p: float(tz[tz.Parameter == p]["Value"].mean()) for p in tz.Parameter.unique()}

    np.savez_compressed(HERE / "plumbing_goldens.npz", **arrays)
    (HERE / "plumbing_goldens.json").write_text(json.dumps(meta, indent=1, ensure_ascii=False))
    print(f"wrote {len(arrays)} arrays for {len(CASES)} cases")


if __name__ == "__main__":
    main()
