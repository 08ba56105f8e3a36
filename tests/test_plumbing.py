"""DataSet / Standardizer / parray / Regressor plumbing against goldens captured from the
reference implementation (tests/golden/make_plumbing_goldens.py) and the reference's own
known-answer tests (tests/test_aggregation.py:32-63, tests/test_arrays.py:29-181,
tests/test_regression.py:49-112)."""
import json
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

import gumbi_amd as gmb
from gumbi_amd import DataSet, Standardizer, WideData, mvuparray, parray, uarray, uparray
from gumbi_amd.regression.base import Regressor
from gumbi_amd.utils.gp_utils import find_constrained_prior, parse_ls_limits

GOLD = Path(__file__).resolve().parent / "golden"
ARR = np.load(GOLD / "plumbing_goldens.npz")
META = json.loads((GOLD / "plumbing_goldens.json").read_text())

example_stdzr = {
    "a": {"μ": -0.762, "σ2": 1.258**2}, "b": {"μ": -0.0368, "σ2": 0.351**2}, "c": {"μ": -5.30, "σ2": 0.582**2},
    "d": {"μ": -0.307, "σ2": 0.158**2}, "e": {"μ": -1.056, "σ2": 0.398**2}, "f": {"μ": 3.34, "σ2": 0.1501**2},
    "X": {"μ": -0.282, "σ2": 1**2}, "Y": {"μ": 4.48, "σ2": 0.75**2}, "lg10_Z": {"μ": 5, "σ2": 2**2},
}
log_vars = ["d", "f", "b", "c", "Y"]
logit_vars = ["e", "X"]


def make_stdzr():
    return Standardizer(**{k: dict(v) for k, v in example_stdzr.items()}, log_vars=log_vars, logit_vars=logit_vars)


class Probe(Regressor):
    def fit(self, *a, **k):
        pass

    def build_model(self, *a, **k):
        pass

    def predict(self, points_array, with_noise=True, **k):
        raise RuntimeError


@pytest.fixture
def example_estimates():
    es = pd.read_pickle(GOLD / "test_dataset.pkl")
    return DataSet.from_tidy(es, names_column="Parameter", stdzr=make_stdzr())


@pytest.fixture
def example_gp(example_estimates):
    return Probe(example_estimates, outputs="d")


# ---------------------------------------------------------------------------- Standardizer
def test_stdz_known_answers():
    s = make_stdzr()
    nat = {p: s.untransform(p, v["μ"]) for p, v in example_stdzr.items()}
    assert np.allclose([s.stdz(p, nat[p]) for p in nat], 0)
    assert np.allclose([s.unstdz(p, s.stdz(p, nat[p])) for p in nat], list(nat.values()))

    stdzr = Standardizer(x={"μ": 1, "σ2": 0.1}, d={"μ": 0, "σ2": 0.1}, log_vars=["d"])
    assert stdzr.transform("x", μ=1) == 1
    assert stdzr.stdz("x", 1) == 0.0
    assert stdzr.unstdz("x", 0) == 1.0
    assert np.isclose(stdzr.stdz("x", 1 + 0.1**0.5), 1.0)
    assert np.isclose(stdzr.unstdz("x", 1), 1 + 0.1**0.5)
    assert stdzr.stdz("d", 1) == 0.0
    assert np.isclose(stdzr.stdz("d", np.exp(0.1**0.5)), 1.0)
    assert stdzr.transform("x", μ=1, σ2=0.1) == (1, 0.1)
    assert stdzr.stdz("x", 1, 0.1) == (0.0, 1.0)
    assert stdzr.stdz("d", 1, 0.1) == (0.0, 1.0)
    assert stdzr.transform("d", 1, 0.1) == (0.0, 0.1)
    assert np.allclose(stdzr.stdz(pd.Series(np.arange(1, 5), name="x")).values, [0.0, 3.162278, 6.324555, 9.486833])
    assert np.allclose(stdzr.stdz(pd.Series(np.arange(1, 5), name="d")).values,
                       [0.0, 2.19192384, 3.4741171, 4.38384769])


def test_stdzr_sigma_key_and_validation():
    s = Standardizer(d={"μ": -0.307, "σ": 0.158}, log_vars=["d"])
    assert np.isclose(s["d"]["σ2"], 0.158**2) and "σ" not in s["d"]
    with pytest.raises(AssertionError):
        Standardizer(x={"σ2": 1.0})
    with pytest.raises(TypeError):
        Standardizer(x={"μ": 0, "σ2": 1}, log_vars=3)
    with pytest.raises(ValueError):
        s.transform("d")
    merged = s | {"q": {"μ": 1.0, "σ2": 2.0}}
    assert isinstance(merged, Standardizer) and merged.log_vars == ["d"] and "q" in merged


def test_dataset_roundtrip_and_from_dataframe():
    df = pd.read_pickle(GOLD / "estimates_test_data.pkl")
    ds = DataSet.from_tidy(df, names_column="Parameter", log_vars=["Y", "c", "b"], logit_vars=["X", "e"])
    assert ds.outputs == META["estimates_outputs"]
    assert list(ds.wide.columns) == META["estimates_wide_columns"]
    for name, ref in META["estimates_stdzr"].items():
        assert np.isclose(ds.stdzr[name]["μ"], ref["μ"], rtol=1e-12, atol=0)
        assert np.isclose(ds.stdzr[name]["σ2"], ref["σ2"], rtol=1e-12, atol=0)
    tz = ds.tidy.z
    assert tz.shape == ds.tidy.shape
    for p, ref in META["estimates_tidy_z_mean"].items():
        assert np.isclose(tz[tz.Parameter == p]["Value"].mean(), ref, atol=1e-12)
        assert abs(tz[tz.Parameter == p]["Value"].mean()) < 1e-10

    wide_out = ds.wide
    specs = dict(outputs=ds.outputs, log_vars=["Y", "c", "b"], logit_vars=["X", "e"])
    wide_in_wd = WideData(wide_out, **specs)
    wide_in_ds = DataSet(wide_out, **specs)
    pd.testing.assert_frame_equal(pd.DataFrame(wide_in_wd), pd.DataFrame(wide_out))
    pd.testing.assert_frame_equal(pd.DataFrame(wide_in_ds.wide), pd.DataFrame(wide_out))
    ds.wide = wide_out.drop(0)
    pd.testing.assert_frame_equal(pd.DataFrame(ds.wide), pd.DataFrame(wide_out).drop(0))
    assert ds.tidy.shape[0] == 6 * (wide_out.shape[0] - 1)


# ---------------------------------------------------------------------------- arrays
def test_parray_known_answers():
    stdzr = make_stdzr()
    rpa = parray(d=np.arange(5, 10) / 10, stdzr=stdzr)
    assert np.allclose(rpa, np.arange(5, 10) / 10)
    assert np.allclose(rpa.values(), np.arange(5, 10) / 10)
    assert np.allclose(rpa.t, [-0.69314718, -0.51082562, -0.35667494, -0.22314355, -0.10536052])
    assert np.allclose(rpa.z, [-2.4439695, -1.29003559, -0.31439838, 0.53073702, 1.27619927])
    assert np.allclose(np.min(np.sqrt(np.mean(np.square(rpa - rpa[0] - 0.05)))).t, -1.5791256)
    assert np.argmax(rpa.values()) == 4
    pa1 = parray(param=np.arange(5), stdzr=stdzr)
    assert np.allclose(pa1, np.arange(5)) and np.allclose(pa1.t, np.arange(5)) and np.allclose(pa1.z, np.arange(5))
    pa2 = parray(param=np.arange(5), other=np.arange(5) * 10, stdzr=stdzr)
    assert np.allclose(pa2.get("param").values(), [0.0, 1.0, 2.0, 3.0, 4.0])
    assert np.allclose(pa2.get("other").values(), [0.0, 10.0, 20.0, 30.0, 40.0])
    assert pa2.values().shape == (2, 5)
    assert pa1[0].values() == 0
    assert np.allclose(pa1[::2].values(), [0, 2, 4])
    assert np.allclose(pa2[::2].get("param").values(), [0, 2, 4])


def test_parray_layers_and_stacking():
    stdzr = make_stdzr()
    pa = parray(X=np.linspace(0.1, 0.9, 4), stdzr=stdzr)
    two = pa.add_layers(Y=np.full(4, 88.0))
    assert two.names == ["X", "Y"] and two.shape == (4,)
    filled = pa.fill_with(Code=2)
    assert np.allclose(filled["Code"].values(), 2)
    col = pa[:, None]
    assert col.shape == (4, 1) and col.names == ["X"]
    tall = parray.vstack([col.add_layers(P=0.0), col.add_layers(P=1.0)])
    assert tall.shape == (8, 1) and np.allclose(tall["P"].values().squeeze(), [0] * 4 + [1] * 4)
    with pytest.raises(ValueError):
        parray.vstack([col, two[:, None]])
    zs = parray(X=np.array([-1.0, 0.0, 1.0]), stdzr=stdzr, stdzd=True)
    assert np.allclose(zs.z.values(), [-1.0, 0.0, 1.0])
    assert np.allclose(two.ravel().reshape(2, 2)["Y"].values(), 88.0)
    with pytest.raises(ValueError):
        parray(stdzr=stdzr)


def test_uarray_arithmetic():
    ua1, ua2 = uarray("A", μ=1, σ2=0.1), uarray("A", μ=2, σ2=0.2)
    ua3 = ua1 + 1
    assert np.isclose(ua3.μ, 2.0) and np.isclose(ua3.σ2, 0.1) and np.isclose(ua3.σ, 0.3162277660)
    assert np.isclose((ua2 + ua1).μ, 3.0) and np.isclose((ua2 + ua1).σ2, 0.3)
    assert np.isclose((ua2 - ua1).μ, 1.0) and np.isclose((ua2 - ua1).σ2, 0.3)
    ua6 = uarray.stack([ua1, ua2]).mean(axis=0)
    assert np.isclose(ua6.μ, 1.5) and np.isclose(ua6.σ2, 0.075)
    ua7 = uarray("B", np.arange(1, 5) / 10, np.arange(1, 5) / 100)
    assert np.isclose(ua7.mean().μ, 0.25) and np.isclose(ua7.mean().σ2, 0.00625)
    ua8 = ua1 + ua7.mean()
    assert ua8.name == "(A+B)" and np.isclose(ua8.μ, 1.25) and np.isclose(ua8.σ2, 0.10625)
    assert np.allclose(ua7.dist.ppf(0.95), [0.26448536, 0.43261743, 0.58489701, 0.72897073])


def test_uparray_roundtrip_and_goldens():
    stdzr = make_stdzr()
    upa = uparray("c", np.arange(1, 5) / 10, np.arange(1, 5) / 100, stdzr)
    assert np.allclose(upa.μ, np.arange(1, 5) / 10) and np.allclose(upa.σ2, np.arange(1, 5) / 100)
    rt_mu, rt_var = upa.stdzr.unstdz(upa.name, upa.z.μ, upa.z.σ2)
    assert np.allclose(upa.μ, rt_mu) and np.allclose(upa.σ2, rt_var)
    upa2 = uparray(upa.name, upa.z.μ, upa.z.σ2, stdzr, stdzd=True)
    assert np.allclose(upa.μ, upa2.μ) and np.allclose(upa.z.σ2, upa2.z.σ2)
    assert np.isclose(upa.mean().μ, 0.22133638) and np.isclose(upa.mean().σ2, 0.00625)
    assert np.allclose(upa.dist.ppf(0.025), [0.08220152, 0.1515835, 0.21364308, 0.27028359])
    # un-standardisation of engine output, as predict_points does (reference base.py:578-580)
    mu_z, var_z = ARR["unstdz/mu_z"], ARR["unstdz/var_z"]
    for name in ["d", "e", "a", "not_in_stdzr"]:
        got = uparray(name, mu_z, var_z, stdzr=stdzr, stdzd=True)
        assert np.array_equal(np.asarray(got.μ), ARR[f"unstdz/{name}/mu"])
        assert np.array_equal(np.asarray(got.σ2), ARR[f"unstdz/{name}/var"])
    for name in ["X", "Y", "lg10_Z"]:
        got = parray(**{name: ARR[f"stdz/{name}/nat"]}, stdzr=stdzr).z.values()
        assert np.array_equal(got, ARR[f"stdz/{name}/z"])
    grid = upa2.reshape(2, 2)
    assert grid.shape == (2, 2) and grid.name == "c" and np.allclose(grid.μ.ravel(), upa.μ)


def test_mvuparray():
    stdzr = Standardizer(c={"μ": -5.30, "σ": 0.582}, d={"μ": -0.307, "σ": 0.158}, log_vars=["d", "c"])
    m_upa = uparray("c", np.arange(1, 5) / 10, np.arange(1, 5) / 100, stdzr)
    r_upa = uparray("d", np.arange(1, 5) / 10 + 0.5, np.arange(1, 5) / 100 * 2, stdzr)
    cor = np.array([[1, -0.6], [-0.6, 1]])
    mv = mvuparray(m_upa, r_upa, cor=cor)
    assert mv.names == ["c", "d"] and mv.shape == (4,)
    assert np.allclose(mv.μ["d"].values(), [0.6, 0.7, 0.8, 0.9])
    assert np.allclose(mv.get("d").σ2, np.arange(1, 5) / 100 * 2)
    assert np.allclose(mv.z.get("c_z").μ, [5.15019743, 6.34117197, 7.03784742, 7.53214651])
    pa = mv.parray(c=0.09, d=0.61)
    assert np.isclose(mv[0].dist.cdf(pa.z.values()), 0.023900979112885523, rtol=2e-3)
    assert mv.reshape(2, 2).shape == (2, 2) and mv.reshape(2, 2).names == ["c", "d"]


# ---------------------------------------------------------------------------- Regressor parsing (reference tests)
def test_gp_default_fit_parsing(example_gp):
    gp = example_gp.specify_model(continuous_dims=["X", "Y"])
    assert gp.continuous_dims == ["X", "Y"] and gp.categorical_dims == []
    X, y = gp.get_structured_data()
    assert X.shape == (66,) and len(X.names) == 2 and y.shape == (66,)


def test_gp_numerical_and_categorical_continuous(example_gp):
    for third in ("lg10_Z", "Name"):
        gp = example_gp.specify_model(continuous_dims=["X", "Y", third])
        assert gp.continuous_dims == ["X", "Y", third] and gp.categorical_dims == []
        for dim in gp.continuous_dims:
            assert len(gp.continuous_levels[dim]) == len(gp.data.tidy[dim].unique())
            assert len(gp.continuous_coords[dim].values()) == len(gp.continuous_levels[dim])
        X, y = gp.get_structured_data()
        assert X.shape == (66,) and len(X.names) == 3 and y.shape == (66,)


def test_gp_params_fit_parsing(example_gp):
    gp = example_gp.specify_model(outputs=["d", "c"], continuous_dims=["X", "Y"])
    assert gp.continuous_dims == ["X", "Y"] and gp.categorical_dims == ["Parameter"]
    assert gp.categorical_levels == {"Parameter": ["d", "c"]}
    assert gp.categorical_coords == {"Parameter": {"d": 1, "c": 0}}
    X, y = gp.get_structured_data()
    assert X.shape == (66,) and len(X.names) == 2 and y.shape == (66,) and len(y.names) == 2


def test_gp_single_input_fit_parsing(example_gp):
    gp = example_gp.specify_model(continuous_dims=["X", "Y", "Name"],
                                  continuous_levels={"Name": ["intense-opportunity"]})
    assert gp.continuous_dims == ["X", "Y"]
    assert gp.filter_dims == {"Name": ["intense-opportunity"], "Parameter": ["d"]}
    X, y = gp.get_structured_data()
    assert X.shape == (7,) and len(X.names) == 2 and y.shape == (7,)


def test_specify_model_errors(example_estimates):
    with pytest.raises(TypeError):
        Probe("not a dataset")
    gp = Probe(example_estimates, outputs="d")
    with pytest.raises(ValueError):
        gp.specify_model(continuous_dims=["X"], categorical_dims=["X"])
    with pytest.raises(ValueError):
        gp.specify_model(continuous_dims=["X", "nope"])
    with pytest.raises(ValueError):
        gp.specify_model(outputs=["zzz"], continuous_dims=["X"])
    with pytest.raises(ValueError):
        gp.specify_model(continuous_dims=["X", "Y"], linear_dims=["lg10_Z"])
    with pytest.raises(KeyError):
        gp.specify_model(continuous_dims=["X"], continuous_levels={"Y": [1.0]})
    levels = {"Name": "intense-opportunity"}
    gp.specify_model(continuous_dims=["X", "Y", "Name"], continuous_levels=levels)
    assert levels == {"Name": "intense-opportunity"}  # caller's dict is not mutated (reference quirk fixed)


# ---------------------------------------------------------------------------- goldens from the reference
@pytest.mark.parametrize("case", [c for c in META if c not in ("estimates_stdzr", "estimates_outputs",
                                                                "estimates_wide_columns", "estimates_tidy_z_mean")])
def test_shaped_data_grid_and_points_match_reference(example_estimates, case):
    m = META[case]
    gp = Probe(example_estimates, outputs="d")
    gp.specify_model(**m["kwargs"])
    assert gp.dims == m["dims"]
    assert gp.continuous_dims == m["continuous_dims"] and gp.categorical_dims == m["categorical_dims"]
    assert {k: [x if isinstance(x, str) else float(x) for x in v] for k, v in gp.filter_dims.items()} == m["filter_dims"]
    assert {d: {str(k): int(v) for k, v in c.items()} for d, c in gp.categorical_coords.items()} == m["categorical_coords"]
    X, y = gp.get_shaped_data("mean")
    assert X.flags["C_CONTIGUOUS"] and X.dtype == np.float64
    assert np.array_equal(X, ARR[f"{case}/X"]) and np.array_equal(y, ARR[f"{case}/y"])
    # lengthscale-prior limits (sort-based here, pdist-based in the reference)
    idx_s = [gp.dims.index(d) for d in gp.continuous_dims]
    lo, up = parse_ls_limits(X[:, idx_s], ARD=True)
    assert np.allclose(lo, ARR[f"{case}/ls_lower_ard1"], rtol=1e-14, atol=0)
    assert np.allclose(up, ARR[f"{case}/ls_upper_ard1"], rtol=1e-14, atol=0)
    # grid and the points_array the backend receives
    gp.prepare_grid(resolution=4)
    assert list(gp.grid_parray.shape) == m["grid_shape"] and list(gp.grid_points.names) == m["grid_names"]
    for name in m["grid_names"]:
        assert np.allclose(gp.grid_points[name].values(), ARR[f"{case}/grid/{name}"], rtol=1e-13, atol=0)
    points = gp.grid_points
    if gp.categorical_dims:
        points = gp.append_categorical_points(points, categorical_levels=m["cat_levels"])
    output = gp._parse_prediction_output(None)
    assert list(output) == m["predict_output"]
    pa, _, _ = gp._prepare_points_for_prediction(points, output=output)
    assert pa.shape == ARR[f"{case}/points_array"].shape
    assert np.allclose(pa, ARR[f"{case}/points_array"], rtol=1e-12, atol=1e-13)


def test_prepare_grid_errors_and_at(example_gp):
    gp = example_gp.specify_model(continuous_dims=["X", "Y", "lg10_Z"])
    with pytest.raises(ValueError):
        gp.predict_grid()
    with pytest.raises(TypeError):
        gp.prepare_grid(at={"X": 0.5})
    with pytest.raises(TypeError):
        gp.prepare_grid(resolution="fine")
    at = gp.parray(lg10_Z=8.0)
    grid = gp.prepare_grid(at=at, resolution={"X": 5, "Y": 3, "lg10_Z": 2})
    assert grid.shape == (5, 3) and gp.prediction_dims == ["X", "Y"]
    assert np.allclose(grid["lg10_Z"].values(), 8.0) and gp.grid_points.shape == (15,)
    with pytest.raises(ValueError):
        gp.prepare_grid(at=gp.parray(X=0.5, Y=50.0, lg10_Z=8.0))
    xg, yg = gp.marginal_grids("X", "Y")
    assert xg.shape == (5, 3) and yg.shape == (5, 3)


def test_find_constrained_prior_mass():
    from scipy.special import gammaincc

    for lo, up, mass in [(0.05, 3.0, 0.98), (0.3, 0.9, 0.9), (0.01, 10.0, 0.98)]:
        p = find_constrained_prior(lo, up, mass)
        cdf = lambda x: gammaincc(p["alpha"], p["beta"] / x)  # noqa: E731
        assert np.isclose(cdf(lo), (1 - mass) / 2, rtol=1e-8)
        assert np.isclose(cdf(up), (1 + mass) / 2, rtol=1e-8)
    with pytest.raises(ValueError):
        find_constrained_prior(1.0, 0.5)


def test_parse_ls_limits_edge_cases():
    X = np.array([[0.0, 1.0], [0.0, 1.5], [0.0, 4.0]])
    lo, up = parse_ls_limits(X, ARD=True)
    assert lo == [0.01, 0.5] and up == [1, 3.0]  # degenerate column falls back to (0.01, 1)
    lo, up = parse_ls_limits(X, ARD=True, lower=[0.2, 0.7], upper=2.0)
    assert lo == [0.2, 0.7] and up == [2.0, 2.0]
    with pytest.raises(ValueError):
        parse_ls_limits(X, ARD=True, lower=[0.1, 0.2, 0.3])


# ---------------------------------------------------------------------------- ls_bounds helper
DELTA_CASES = {  # as in tests/golden/make_deltas_goldens.py (captured from the reference's make_deltas_parray)
    "standardized_XY": ("standardized", {"X": [0.5, None], "Y": [0.25, 6.0]}),
    "natural_log_logit": ("natural", {"Y": [20.0, None], "X": [0.05, 0.3], "d": [None, 0.5]}),
    "transformed_mix": ("transformed", {"X": [0.2, 1.5], "Y": [0.1, 2.0], "d": [0.5, 3.0]}),
    "standardized_plain": ("standardized", {"lg10_Z": [0.5, 3.0], "a": [None, 2.0]}),
}


@pytest.mark.parametrize("case", sorted(DELTA_CASES))
def test_make_deltas_parray_matches_reference(case):
    """``make_deltas_parray`` (reference gumbi/array_utils.py:8-33) and the bounds that reach the lengthscale
    prior through ``_prepare_lengthscales`` -> ``parse_ls_limits`` (pymc/GP.py:630-650, gp_utils.py:15-48)."""
    from gumbi_amd.array_utils import make_deltas_parray

    gold = np.load(GOLD / "deltas_goldens.npz")
    scale, deltas = DELTA_CASES[case]
    pa = make_deltas_parray(stdzr=make_stdzr(), scale=scale, **deltas)
    assert tuple(pa.shape) == tuple(gold[f"{case}/shape"])
    for dim in deltas:
        np.testing.assert_allclose(np.asarray(pa[dim].values(), float), gold[f"{case}/{dim}"], rtol=1e-12, equal_nan=True)
    zb = [[None if np.isnan(b) else float(b) for b in np.atleast_1d(pa[dim].z.values().squeeze())] for dim in deltas]
    lower, upper = (list(t) for t in zip(*zb))
    lo, up = parse_ls_limits(gold["points"][:, : len(deltas)], ARD=True, lower=lower, upper=upper)
    np.testing.assert_allclose(lo, gold[f"{case}/ls_lower"], rtol=1e-12)
    np.testing.assert_allclose(up, gold[f"{case}/ls_upper"], rtol=1e-12)


def test_make_deltas_parray_natural_scale_of_plain_variables():
    """The reference raises a TypeError for natural- / transformed-scale deltas of UNtransformed variables
    (a list minus a float in aggregation.py:393); here they mean what the docstring promises: delta / sigma."""
    from gumbi_amd.array_utils import make_deltas_parray

    pa = make_deltas_parray(stdzr=make_stdzr(), scale="natural", a=[0.3, 4.0], lg10_Z=[None, 5.0])
    np.testing.assert_allclose(pa["a"].z.values().squeeze(), [0.3 / 1.258, 4.0 / 1.258], rtol=1e-12)
    z = pa["lg10_Z"].z.values().squeeze()
    assert np.isnan(z[0]) and abs(z[1] - 2.5) < 1e-12
    with pytest.raises(ValueError):
        make_deltas_parray(stdzr=make_stdzr(), scale="zscore", a=[0.3, 4.0])
