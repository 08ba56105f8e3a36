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


def additive_problem(n=60, seed=4, hetero=True, lin=True, two_outputs=True):
    """Small additive model: 2 continuous dims (1 linear), one categorical dim with 3 levels, optionally 2
    outputs: K = (k0 [+lin0]) o B_out + (k1 [+lin1]) o B_cat o B_out."""
    rng = np.random.default_rng(seed)
    P = 2 if two_outputs else 1
    Xc = rng.standard_normal((n, 2))
    cat = rng.integers(0, 3, n).astype(float)
    cols = [Xc, cat[:, None]]
    X1 = np.column_stack(cols)
    if two_outputs:
        X = np.vstack([np.column_stack([X1, np.full(n, p)]) for p in range(P)])
    else:
        X = X1
    D = X.shape[1]
    spec = O.make_spec(D, [0, 1], idx_lin=[1] if lin else [], coreg=[(2, 3)], out_col=3 if two_outputs else -1,
                       n_out=P if two_outputs else 0, hetero_noise=hetero, additive=True)
    nt = O.theta_size(spec)
    theta = np.abs(rng.normal(1.0, 0.3, nt)) + 0.2
    y = rng.standard_normal(len(X))
    return spec, theta, X, y


@pytest.mark.parametrize("two_outputs,lin,hetero", [(True, True, True), (False, True, False), (True, False, False)])
def test_additive_model_covariance_and_gradient(two_outputs, lin, hetero):
    """Additive GP (pymc/GP.py:732-754): the covariance is the sum of the per-term covariances, and the
    analytic gradient matches central differences for EVERY parameter, including the per-dim kernels."""
    spec, theta, X, y = additive_problem(two_outputs=two_outputs, lin=lin, hetero=hetero)
    terms = O.additive_terms(spec, theta)
    assert len(terms) == 2
    K = O.cov_full(spec, theta, X, dist_mode="direct")
    K_manual = sum(O.cov_full(sp, th, X, dist_mode="direct") for sp, th, _ in terms)
    assert np.allclose(K, K_manual, rtol=0, atol=1e-14)
    # every kernel parameter of theta is owned by exactly one term, except the shared output table
    owners = np.zeros(theta.size, int)
    for _, _, mp in terms:
        for dst in mp:
            if dst >= 0:
                owners[dst] += 1
    n_ls = 2
    assert owners[n_ls + 1] == 0  # sigma belongs to the noise
    val, g = O.nlml_and_grad(spec, theta, X, y, dist_mode="direct")
    assert np.isclose(val, O.nlml(spec, theta, X, y, dist_mode="direct"), rtol=1e-12)
    h = 1e-6
    for i in range(theta.size):
        tp, tm = theta.copy(), theta.copy()
        tp[i] += h
        tm[i] -= h
        fd = (O.nlml(spec, tp, X, y, dist_mode="direct") - O.nlml(spec, tm, X, y, dist_mode="direct")) / (2 * h)
        assert abs(fd - g[i]) <= 2e-5 * max(1.0, abs(g[i])), (i, fd, g[i])
    # predictions go through the same summed covariance
    Xs = X[:7].copy()
    Xs[:, 0] += 0.1
    mu, var = O.predict(spec, theta, X, y, Xs, dist_mode="direct")
    assert np.all(np.isfinite(mu)) and np.all(var > 0)


def test_host_posterior_blockwise_is_the_oracle():
    """tests/host_reference.py (the full-size host check of tests/test_gpu_configs.py) against ``O.predict`` / ``O.nlml`` at a size where both run in a second."""
    from host_reference import host_posterior_blockwise

    def rel(a, b):
        return np.max(np.abs(a - b)) / np.max(np.abs(b))

    N, d = 1500, 8
    X, y, ls = O.synthetic_table(N, d)
    spec = O.make_spec(d, range(d), kind="Matern52")
    theta = O.pack_theta(spec, ls, 1.0, 0.2)
    Xs = O.synthetic_grid(d, res=6)
    mu, var, nl = host_posterior_blockwise(spec, theta, X, y, Xs, block=200, threads=3)
    mu_r, var_r = O.predict(spec, theta, X, y, Xs, with_noise=True)
    assert rel(mu, mu_r) < 1e-12 and np.max(np.abs(var - var_r)) < 1e-12 and np.isclose(nl, O.nlml(spec, theta, X, y), rtol=1e-13)


@pytest.mark.parametrize("kind", ["ExpQuad", "Matern52", "Matern12"])
def test_host_baseline_variants_of_the_gradient_agree_with_the_pinned_form(kind):
    """What bench.py's host baseline times (PyMC's GEMM distance expansion with its GEMM-form ARD gradient, Sigma^-1 by LAPACK
    dpotri) against the form the parity tests pin (direct differences, L^-1 by a triangular solve)."""
    N, d = 500, 5
    X, y, ls = O.synthetic_table(N, d, seed=13)
    spec = O.make_spec(d, range(d), kind=kind)
    theta = O.pack_theta(spec, ls, 1.2, 0.25)
    v0, g0 = O.nlml_and_grad(spec, theta, X, y, dist_mode="direct")
    v1, g1 = O.nlml_and_grad(spec, theta, X, y, dist_mode="gemm", inverse="potri")
    tol = 1e-11 if kind != "Matern12" else 1e-7  # (the kinked kernel sees the expansion's rounding near r = 0)
    assert abs(v1 - v0) <= tol * abs(v0) and np.max(np.abs(g1 - g0)) <= tol * np.max(np.abs(g0))
