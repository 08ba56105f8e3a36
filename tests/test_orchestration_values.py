"""Value-level checks of the orchestration rows of SURVEY.md section 8(f): ``cross_validate``
(reference gumbi/regression/base.py:844-1105), ``propose`` (:816-838) and ``get_conditional_prediction``
(:1111-1178).

CPU half (``-m "not gpu"``): the three calls run with the numeric engine replaced by the oracle stand-in
(tests/oracle_engine.py) and are compared with the same quantities obtained another way -- the manual
fit / predict pipeline on the returned splits, numerical quadrature of the acquisition, the grid's own
nodes.  GPU half (``-m gpu``): the same calls through libgumbi_hip.so on the same splits / grids give the
same VALUES as the oracle-driven run.
"""
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

from oracle_engine import OracleEngine

GOLD = Path(__file__).resolve().parent / "golden"

STDZR = {"d": {"μ": -0.307, "σ2": 0.158**2}, "X": {"μ": -0.282, "σ2": 1.0}, "Y": {"μ": 4.48, "σ2": 0.75**2}}


def _dataset():
    import gumbi_amd as gmb

    stdzr = gmb.Standardizer(**{k: dict(v) for k, v in STDZR.items()}, log_vars=["d", "Y"], logit_vars=["X"])
    es = pd.read_pickle(GOLD / "test_dataset.pkl")
    return gmb.DataSet.from_tidy(es, names_column="Parameter", stdzr=stdzr)


def _use_oracle(monkeypatch):
    from gumbi_amd.regression import hip_gp, icm

    monkeypatch.setattr(hip_gp, "Engine", OracleEngine)
    monkeypatch.setattr(icm, "Engine", OracleEngine)


def _model(**gp_kwargs):
    import gumbi_amd as gmb

    gp = gmb.GP(_dataset(), outputs="d", **gp_kwargs)
    gp.specify_model(continuous_dims=["X", "Y"])
    gp.build_model()
    return gp


def _cross_validate(gp):
    return gp.cross_validate(n_train=50, seed=3, errors="standardized")


def _grid_prediction(gp, resolution=9):
    gp.find_MAP()
    gp.prepare_grid(resolution=resolution)
    gp.predict_grid()
    return gp


# ---------------------------------------------------------------------------------------------------
# CPU: orchestration against independently obtained values (numerics by the oracle)
# ---------------------------------------------------------------------------------------------------
def test_cross_validate_equals_the_manual_pipeline_on_its_splits(monkeypatch):
    import gumbi_amd as gmb

    _use_oracle(monkeypatch)
    gp = _model()
    res = _cross_validate(gp)
    assert res["train"]["errors"].shape == (50,) and res["test"]["errors"].shape == (16,)
    # the same split fitted by hand: a fresh GP on the returned training DataSet, MAP, predictions at the
    # training and held-out inputs, errors and NLPDs from the returned predictions
    train = gmb.GP(res["train"]["data"], outputs="d", seed=3)
    train.specify_model(continuous_dims=["X", "Y"])
    train.build_model()
    train.find_MAP()
    for part in ("train", "test"):
        holder = gmb.GP(res[part]["data"], outputs="d", seed=3)
        holder.specify_model(continuous_dims=["X", "Y"])
        Xs, ys = holder.get_structured_data()
        pred = train.predict_points(Xs)
        assert np.allclose(res[part]["errors"], ys.z.values() - pred.z.μ, rtol=1e-9, atol=1e-12)
        assert np.allclose(res[part]["NLPDs"], pred.nlpd(ys.values()), rtol=1e-9, atol=1e-12)
    # the training errors of a GP with fitted noise are small next to the held-out ones' scale (1 = one
    # standard deviation of the output), and the split is a partition of the 66 observations
    assert np.sqrt(np.mean(res["train"]["errors"] ** 2)) < 0.5
    both = pd.concat([pd.DataFrame(res[p]["data"].wide) for p in ("train", "test")])
    assert len(both) == 66 and not both.duplicated(subset=["X", "Y", "d"]).any()


def test_propose_surface_is_the_target_value_expected_improvement(monkeypatch):
    from scipy import integrate, stats

    _use_oracle(monkeypatch)
    gp = _grid_prediction(_model(), resolution=7)
    pz = gp.predictions.z
    mu, s2 = np.asarray(pz.μ).ravel(), np.asarray(pz.σ2).ravel()
    target = float(np.asarray(gp.predictions.μ).ravel()[11])
    tz = float(np.asarray(gp.parray(d=target, stdzd=False).z.values()).ravel()[0])
    obs = gp.get_filtered_data(standardized=True)
    obs = obs[obs[gp.out_col] == "d"]["Value"].to_numpy()
    best = float(np.min((obs - tz) ** 2))
    prop = gp.propose(target, acquisition="EI")
    surf = np.asarray(gp.proposal_surface).ravel()
    for i in (0, 11, 24, 48):  # E[max(0, best - (t - y)^2)], y ~ N(mu_i, s2_i), by quadrature
        sd = np.sqrt(s2[i])
        lo, hi = tz - np.sqrt(best), tz + np.sqrt(best)
        val, _ = integrate.quad(lambda y: (best - (tz - y) ** 2) * stats.norm.pdf(y, mu[i], sd), lo, hi,
                                epsabs=1e-13, epsrel=1e-11)
        assert np.isclose(surf[i], val, rtol=1e-7, atol=1e-12), (i, surf[i], val)
    assert gp.proposal_idx == int(np.argmax(surf))
    assert np.asarray(prop) == np.asarray(gp.predictions_X.ravel()[gp.proposal_idx])
    # posterior-density acquisition: log N(t; mu, s2) up to the sign convention
    gp.propose(target, acquisition="PD")
    assert np.allclose(np.asarray(gp.proposal_surface).ravel(), stats.norm.logpdf(tz, mu, np.sqrt(s2)), rtol=1e-10)


def test_conditional_prediction_is_the_grid_slice_at_nodes_and_linear_between(monkeypatch):
    _use_oracle(monkeypatch)
    gp = _grid_prediction(_model(), resolution=9)
    mu, s2 = np.asarray(gp.predictions.μ), np.asarray(gp.predictions.σ2)
    ynodes = gp.grid_vectors["Y"].squeeze()
    k = 3
    at_node = float(ynodes.values()[k])
    grid, cond = gp.get_conditional_prediction(Y=at_node)
    assert grid.names == ["X"] and cond.shape == (9,)
    assert np.allclose(cond.μ, mu[:, k], rtol=1e-10) and np.allclose(cond.σ2, s2[:, k], rtol=1e-10)
    assert np.allclose(grid.values(), gp.grid_vectors["X"].squeeze().values())
    # half way between two nodes IN STANDARDIZED units (the space the interpolation works in)
    zn = ynodes.z.values()
    mid = gp.parray(Y=0.5 * (zn[k] + zn[k + 1]), stdzd=True)
    _, cond_mid = gp.get_conditional_prediction(Y=float(mid.values()))
    assert np.allclose(cond_mid.μ, 0.5 * (mu[:, k] + mu[:, k + 1]), rtol=1e-9)
    assert np.allclose(cond_mid.σ2, 0.5 * (s2[:, k] + s2[:, k + 1]), rtol=1e-9)
    # ... and the other way round
    _, cond_x = gp.get_conditional_prediction(X=float(gp.grid_vectors["X"].squeeze().values()[5]))
    assert np.allclose(cond_x.μ, mu[5, :], rtol=1e-10)


def test_specify_model_and_shaping_stay_linear_at_c3_and_c5_sizes():
    """DataSet -> specify_model -> get_shaped_data -> prepare_grid at the
    BASELINE sizes (N = 50k and 100k, d = 8) in seconds, not minutes (the reference needs 54 s at 50k for
    get_shaped_data alone; a list-scan subset test in specify_model used to take 215 s here)."""
    import time

    import gumbi_amd as gmb

    for N, budget in ((50_000, 5.0), (100_000, 10.0)):
        rng = np.random.default_rng(2021)
        cols = [f"x{k}" for k in range(8)]
        df = pd.DataFrame(rng.standard_normal((N, 8)), columns=cols)
        df["y"] = rng.standard_normal(N)
        t0 = time.perf_counter()
        gp = gmb.GP(gmb.DataSet(df, outputs=["y"]), outputs=["y"])
        gp.specify_model(continuous_dims=cols)
        X, y = gp.get_shaped_data("mean")
        gp.prepare_grid(at=gp.parray(**{c: 0.0 for c in cols[2:]}), resolution=100)
        dt = time.perf_counter() - t0
        assert X.shape == (N, 8) and y.shape == (N,) and gp.grid_points.shape == (10_000,)
        # standardised columns: exactly what the backend receives
        assert np.allclose(X.mean(0), 0, atol=1e-9) and np.allclose(X.std(0, ddof=1), 1, rtol=1e-9)
        assert dt < budget, f"front end took {dt:.1f} s at N={N}"


# ---------------------------------------------------------------------------------------------------
# GPU: libgumbi_hip.so gives the same values as the oracle-driven run on the same splits / grids
# ---------------------------------------------------------------------------------------------------
def _oracle_twin(fn):
    """fn() with the oracle stand-in patched in (own MonkeyPatch so the HIP run stays untouched)."""
    mp = pytest.MonkeyPatch()
    try:
        _use_oracle(mp)
        return fn()
    finally:
        mp.undo()


@pytest.mark.gpu
def test_cross_validate_values_hip_vs_oracle(gpu):
    hip = _cross_validate(_model())
    ref = _oracle_twin(lambda: _cross_validate(_model()))
    for part in ("train", "test"):
        # identical splits (same seed, same host code) ...
        assert pd.DataFrame(hip[part]["data"].wide).equals(pd.DataFrame(ref[part]["data"].wide))
        # ... and the same numbers up to where two L-BFGS runs on objectives equal to 1e-10 end up
        assert np.allclose(hip[part]["errors"], ref[part]["errors"], rtol=1e-4, atol=1e-6)
        assert np.allclose(hip[part]["NLPDs"], ref[part]["NLPDs"], rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_propose_and_conditional_prediction_values_hip_vs_oracle(gpu):
    def run():
        gp = _grid_prediction(_model(), resolution=9)
        target = 0.75
        out = {"mu": np.asarray(gp.predictions.μ).copy(), "s2": np.asarray(gp.predictions.σ2).copy()}
        for acq in ("EI", "PD"):
            gp.propose(target, acquisition=acq)
            out[acq] = (np.asarray(gp.proposal_surface).copy(), gp.proposal_idx)
        _, cond = gp.get_conditional_prediction(Y=90.0)
        out["cond"] = (np.asarray(cond.μ).copy(), np.asarray(cond.σ2).copy())
        out["theta"] = gp._theta_fitted.copy()
        return out

    hip, ref = run(), _oracle_twin(run)
    assert np.allclose(hip["theta"], ref["theta"], rtol=1e-4)
    assert np.allclose(hip["mu"], ref["mu"], rtol=1e-5) and np.allclose(hip["s2"], ref["s2"], rtol=1e-4)
    for acq in ("EI", "PD"):
        assert np.allclose(hip[acq][0], ref[acq][0], rtol=1e-3, atol=1e-9)
        assert hip[acq][1] == ref[acq][1]
    assert np.allclose(hip["cond"][0], ref["cond"][0], rtol=1e-5) and np.allclose(hip["cond"][1], ref["cond"][1], rtol=1e-4)
    # the same comparison with the optimiser taken out: both engines at ONE theta agree to solver precision
    def at_theta(theta):
        gp = _model()
        gp.find_MAP(theta=theta)
        gp.prepare_grid(resolution=9)
        gp.predict_grid()
        gp.propose(0.75)
        _, cond = gp.get_conditional_prediction(Y=90.0)
        return np.asarray(gp.predictions.μ), np.asarray(gp.proposal_surface), np.asarray(cond.μ)

    a, b = at_theta(ref["theta"]), _oracle_twin(lambda: at_theta(ref["theta"]))
    for x, y in zip(a, b):
        assert np.allclose(x, y, rtol=1e-8, atol=1e-12)


def test_optimiser_loop_keeps_the_host_blas_to_one_thread_only_for_gpu_engines():
    """HipGP.find_MAP runs scipy's L-BFGS-B -- whose LAPACK calls on m x m matrices (m <= 10) wake a many-threaded OpenBLAS for
    0.1 ms and more per iteration -- with the host BLAS limited to one thread, but ONLY when the evaluations do not run on that
    BLAS (engines flagged ``host_blas_free``): the oracle engine of the CPU baseline keeps all its threads."""
    import contextlib

    from gumbi_amd.regression import hip_gp
    from gumbi_amd import engine, distributed
    from gumbi_amd.regression import icm

    assert engine.Engine.host_blas_free and icm.IcmEngine.host_blas_free and distributed.DistributedEngine.host_blas_free
    assert not getattr(OracleEngine, "host_blas_free", False)
    assert isinstance(hip_gp._single_threaded_host_blas(OracleEngine.__new__(OracleEngine)), contextlib.nullcontext)

    class Flagged:
        host_blas_free = True

    threadpoolctl = pytest.importorskip("threadpoolctl")
    before = [c.num_threads for c in threadpoolctl.ThreadpoolController().select(user_api="blas").lib_controllers]
    with hip_gp._single_threaded_host_blas(Flagged()):
        inside = [c.num_threads for c in threadpoolctl.ThreadpoolController().select(user_api="blas").lib_controllers]
    after = [c.num_threads for c in threadpoolctl.ThreadpoolController().select(user_api="blas").lib_controllers]
    assert inside and all(n == 1 for n in inside) and after == before
