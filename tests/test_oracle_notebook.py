"""Pins the ORACLE (and the host-side model logic: plumbing, priors, MAP objective) against PyMC
itself, on CPU: the reference's Multioutput_Regression notebook
(docs/source/notebooks/examples/Multioutput_Regression.ipynb) was run by its author on PyMC 5 -- its
first cell prints a pytensor warning -- and prints the predictive means / variances of a MAP fit on
the package's own example data set (gumbi/data/Example_DataSet.pkl, kept as the data fixture
tests/golden/example_dataset.pkl).  Here the same calls run through ``gumbi_amd.GP`` with the numeric
engine replaced by a numpy stand-in built on the oracle, so no GPU is needed; the GPU test of the same
name in tests/test_gpu_frontend.py does it through libgumbi_hip.so.
"""
from pathlib import Path

import numpy as np
import pandas as pd

from oracle import gp_oracle as O  # noqa: F401
from oracle_engine import OracleEngine

GOLD = Path(__file__).resolve().parent / "golden"

NB_MU = np.array([[-9.59442479, 0.65605058, 0.00646403, 0.81416271, 0.15214448],
                  [-8.05656298, 0.66609041, 0.00635764, 0.81267686, 0.16440518],
                  [-6.40414117, 0.67787309, 0.00620618, 0.8105507, 0.17662809],
                  [-4.75033515, 0.68729924, 0.00617787, 0.81008143, 0.19510875],
                  [-2.94787273, 0.69658766, 0.00619329, 0.81021742, 0.21940875]])
NB_S2 = np.array([[0.01676639, 1.22016392e-04, 0.00134064, 1.22105406e-05, 2.80013388e-04],
                  [0.00462947, 1.03825281e-04, 0.00114775, 1.03319592e-05, 8.16338238e-05],
                  [0.00455407, 9.92785623e-05, 0.00110227, 9.91092000e-06, 8.03754540e-05],
                  [0.00462973, 1.04073081e-04, 0.00115023, 1.03548470e-05, 8.16411508e-05],
                  [0.01685376, 1.22594426e-04, 0.00134647, 1.22658105e-05, 2.81471085e-04]])


def _fit_and_predict(monkeypatch, jacobian, kronecker=False):
    import gumbi_amd as gmb
    from gumbi_amd.regression import hip_gp, icm

    monkeypatch.setattr(hip_gp, "Engine", OracleEngine)
    monkeypatch.setattr(icm, "Engine", OracleEngine)
    df = pd.read_pickle(GOLD / "example_dataset.pkl")
    df = df[(df.Name == "binary-pollen") & (df.Color == "cyan") & (df.Metric == "mean")]
    ds = gmb.DataSet(df, outputs=["a", "b", "c", "d", "e", "f"], log_vars=["Y", "b", "c", "d", "f"],
                     logit_vars=["X", "e"])
    fit_params = ["a", "b", "c", "d", "e"]
    gp = gmb.GP(ds, outputs=fit_params, kronecker=kronecker)
    gp.map_includes_jacobian = jacobian
    gp.fit(continuous_dims="lg10_Z", linear_dims="lg10_Z")
    assert len(gp.model.y) == 70
    gp.prepare_grid(limits=gp.parray(lg10_Z=[1, 9]), resolution=5)
    gp.predict_grid()
    mv = gp.predictions
    mu = np.stack([np.asarray(mv.get(p).μ).ravel() for p in fit_params], axis=1)
    s2 = np.stack([np.asarray(mv.get(p).σ2).ravel() for p in fit_params], axis=1)
    return mu, s2


def test_oracle_reproduces_the_pymc5_notebook(monkeypatch):
    mu, s2 = _fit_and_predict(monkeypatch, jacobian=False)
    assert np.max(np.abs(mu - NB_MU) / np.abs(NB_MU)) < 8e-3
    assert np.max(np.abs(mu[1:4] - NB_MU[1:4]) / np.abs(NB_MU[1:4])) < 1e-3   # interior of the grid: 1e-4 .. 1e-3
    assert np.max(np.abs(s2 - NB_S2) / NB_S2) < 0.06


def test_kronecker_algebra_on_the_oracle_gives_the_same_fit(monkeypatch):
    """The host-side Kronecker algebra of gumbi_amd/regression/icm.py (P systems of size N), driven by the
    oracle instead of the HIP engine: same MAP fit and predictions as the stacked system above."""
    mu_s, s2_s = _fit_and_predict(monkeypatch, jacobian=False, kronecker=False)
    mu_k, s2_k = _fit_and_predict(monkeypatch, jacobian=False, kronecker="auto")
    assert np.max(np.abs(mu_k - mu_s) / np.abs(mu_s)) < 1e-5
    assert np.max(np.abs(s2_k - s2_s) / s2_s) < 1e-4
    assert np.max(np.abs(mu_k[1:4] - NB_MU[1:4]) / np.abs(NB_MU[1:4])) < 1e-3


def test_the_jacobian_reading_of_find_map_does_not(monkeypatch):
    """The PyMC3 objective (log-Jacobians included) misses the printed variances by ~35 %: the notebook
    discriminates between the two readings of ``pm.find_MAP``, which is why ``jacobian=False`` is the
    restated behaviour (pymc/tuning/starting.py: ``model.compile_logp(jacobian=False)``)."""
    mu, s2 = _fit_and_predict(monkeypatch, jacobian=True)
    assert np.max(np.abs(s2 - NB_S2) / NB_S2) > 0.2


def test_additive_front_end_blocks_and_prior_on_cpu(monkeypatch):
    """Host logic of ``specify_model(additive=True)`` (reference ``pymc/GP.py:732-754``) with the oracle standing
    in for the engine: parameter blocks (names and order of include/gumbi_hip.h), prior sum, MAP dictionary."""
    import gumbi_amd as gmb
    from gumbi_amd.regression import hip_gp

    monkeypatch.setattr(hip_gp, "Engine", OracleEngine)
    df = pd.read_pickle(GOLD / "example_dataset.pkl")
    df = df[(df.Color == "cyan") & (df.Metric == "mean") & df.Name.isin(sorted(df.Name.unique())[:3])]
    ds = gmb.DataSet(df, outputs=["a", "b", "c", "d", "e", "f"], log_vars=["Y", "b", "c", "d", "f"],
                     logit_vars=["X", "e"])
    gp = gmb.GP(ds, outputs=["d"], kronecker=False)
    gp.specify_model(continuous_dims=["lg10_Z"], linear_dims=["lg10_Z"], categorical_dims="Name", additive=True)
    gp.build_model()
    spec = gp.model.spec
    assert spec.additive and spec.theta_size() == O.theta_size(spec.as_dict())
    names = [b[0] for b in gp.model.blocks]
    assert names == ["ls_total", "η_total", "σ", "c_total", "τ_total", "W_Name", "κ_Name",
                     "ls_Name", "η_Name", "c_Name", "τ_Name"]
    theta = gp._initial_theta()
    assert theta[gp.model.blocks[names.index("η_Name")][3]] == 2.0
    lp, glp = gp._log_prior(theta)
    a, b = gp.model.ls_params["alpha"], gp.model.ls_params["beta"]
    assert np.isclose(lp, O.log_prior_and_jacobian(spec.as_dict(), theta, a, b), rtol=1e-12)
    h = 1e-6
    for i in range(theta.size):
        tp, tm = theta.copy(), theta.copy()
        tp[i] += h
        tm[i] -= h
        assert abs((gp._log_prior(tp)[0] - gp._log_prior(tm)[0]) / (2 * h) - glp[i]) <= 1e-5 * max(1, abs(glp[i]))
    gp.find_MAP(maxeval=25)
    assert gp.nlml_trace[-1] < gp.nlml_trace[0]
    assert {"ls_Name", "ls_Name_log__", "η_Name", "τ_Name", "c_Name", "W_Name", "κ_Name"} <= set(gp.MAP)
    X = gp.prepare_grid(resolution=7)
    pred = gp.predict_grid(categorical_levels={"Name": gp.categorical_levels["Name"][1]})
    assert pred.shape == gp.grid_parray.shape and np.all(np.isfinite(pred.μ)) and np.all(pred.σ2 > 0)
