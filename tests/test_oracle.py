"""The CPU oracle against its golden vectors (cross-checked against scikit-learn when they were
generated, tests/golden/make_gp_goldens.py) and against finite differences."""
from pathlib import Path

import numpy as np
import pytest

from oracle import gp_oracle as O

GOLD = np.load(Path(__file__).resolve().parent / "golden" / "gp_goldens.npz")
CASES = sorted({k.split("/")[0] for k in GOLD.files})


def spec_for(case):
    X = GOLD[f"{case}/X"]
    if case.startswith("composite"):
        return O.make_spec(4, [0, 1], idx_lin=[1], coreg=[(2, 3)], out_col=3, n_out=2, hetero_noise=True)
    d = X.shape[1]
    return O.make_spec(d, range(d), kind=int(GOLD[f"{case}/kind"]), ard=bool(GOLD[f"{case}/ard"]))


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_goldens(case):
    spec = spec_for(case)
    X, y, Xs, theta = (GOLD[f"{case}/{k}"] for k in ("X", "y", "Xs", "theta"))
    mu, var = O.predict(spec, theta, X, y, Xs, with_noise=True)
    assert np.allclose(mu, GOLD[f"{case}/mu"], rtol=1e-10, atol=1e-12)
    assert np.allclose(var, GOLD[f"{case}/var"], rtol=1e-10, atol=1e-12)
    _, var0 = O.predict(spec, theta, X, y, Xs, with_noise=False)
    assert np.allclose(var0, GOLD[f"{case}/var_noiseless"], rtol=1e-10, atol=1e-12)
    assert np.isclose(O.nlml(spec, theta, X, y), float(GOLD[f"{case}/nlml"]), rtol=1e-12)
    # the two ways of forming r^2 (PyMC's GEMM expansion vs direct differences) agree far inside 1e-8
    mu_d, var_d = O.predict(spec, theta, X, y, Xs, with_noise=True, dist_mode="direct")
    assert np.max(np.abs(mu_d - mu)) <= 1e-8 * np.max(np.abs(mu))
    assert np.max(np.abs(var_d - var)) <= 1e-8


@pytest.mark.parametrize("kind", list(O.KINDS))
@pytest.mark.parametrize("ard", [True, False])
def test_gradient_matches_finite_differences(kind, ard):
    X, y, ls = O.synthetic_table(90, 3, seed=11)
    spec = O.make_spec(3, range(3), kind=kind, ard=ard)
    theta = O.pack_theta(spec, ls if ard else [1.1], 1.3, 0.3)
    _, g = O.nlml_and_grad(spec, theta, X, y)
    h = 1e-6
    for i in range(theta.size):
        tp, tm = theta.copy(), theta.copy()
        tp[i] += h
        tm[i] -= h
        fd = (O.nlml(spec, tp, X, y, dist_mode="direct") - O.nlml(spec, tm, X, y, dist_mode="direct")) / (2 * h)
        assert abs(fd - g[i]) <= 2e-6 * max(1.0, abs(g[i])), (i, fd, g[i])


def test_composite_gradient_golden_and_fd():
    case = "composite_N140"
    spec = spec_for(case)
    X, y, theta = GOLD[f"{case}/X"], GOLD[f"{case}/y"], GOLD[f"{case}/theta"]
    val, g = O.nlml_and_grad(spec, theta, X, y, dist_mode="gemm")
    assert np.isclose(val, float(GOLD[f"{case}/nlml"]), rtol=1e-12)
    assert np.allclose(g, GOLD[f"{case}/grad"], rtol=1e-9, atol=1e-10)
    h = 1e-5
    for i in range(theta.size):
        tp, tm = theta.copy(), theta.copy()
        tp[i] += h
        tm[i] -= h
        fd = (O.nlml(spec, tp, X, y) - O.nlml(spec, tm, X, y)) / (2 * h)
        assert abs(fd - g[i]) <= 1e-5 * max(1.0, abs(g[i])), (i, fd, g[i])


def test_theta_packing_roundtrip():
    spec = O.make_spec(5, [0, 1, 2], idx_lin=[0, 2], coreg=[(3, 4)], out_col=4, n_out=3)
    assert O.theta_size(spec) == 3 + 2 + 3 + 12 + 9 + 9
    rng = np.random.default_rng(0)
    theta = O.pack_theta(spec, [1, 2, 3], 1.5, 0.2, c=[0.1, 0.2], tau=0.7,
                         coreg=[(rng.standard_normal((4, 2)), np.ones(4))], W_out=rng.standard_normal((3, 2)),
                         kappa_out=np.ones(3), W_noise=rng.standard_normal((3, 2)), kappa_noise=np.ones(3))
    p = O.unpack_theta(spec, theta)
    assert np.allclose(p["ls"], [1, 2, 3]) and p["eta"] == 1.5 and p["tau"] == 0.7 and p["W_noise"].shape == (3, 2)


def test_parse_ls_limits_matches_scipy_pdist():
    from scipy.spatial.distance import pdist

    rng = np.random.default_rng(3)
    X = rng.standard_normal((60, 3))
    X[5] = X[4]  # a duplicate pair: zero distances are ignored
    lo, up = O.parse_ls_limits(X, ARD=False)
    d = pdist(X)
    assert np.isclose(lo[0], d[d != 0].min()) and np.isclose(up[0], d.max())
    lo, up = O.parse_ls_limits(X, ARD=True)
    for j in range(3):
        d = pdist(X[:, [j]])
        assert np.isclose(lo[j], max(d[d != 0].min(), 0.01)) and np.isclose(up[j], d.max())


def test_prior_terms():
    from scipy import stats

    x = np.array([0.3, 1.7])
    assert np.allclose(O.logp_inverse_gamma(x, 2.5, 1.2), stats.invgamma(2.5, scale=1.2).logpdf(x))
    assert np.allclose(O.logp_gamma(x, 1.5, 1.0), stats.gamma(1.5, scale=1.0).logpdf(x))
    assert np.allclose(O.logp_exponential(x, 1.0), stats.expon().logpdf(x))
    assert np.allclose(O.logp_halfnormal(x, 10.0), stats.halfnorm(scale=10).logpdf(x))
    assert np.allclose(O.logp_normal(x, 0.0, 3.0), stats.norm(0, 3).logpdf(x))
