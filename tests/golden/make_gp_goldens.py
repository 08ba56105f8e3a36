"""Golden (X, y, theta, X*) -> (mu, var, nlml) vectors for the GP arithmetic.

PyMC (the third-party module that holds the reference's arithmetic) cannot be installed here, so
these vectors come from the numpy restatement in ``oracle/gp_oracle.py`` and every stationary case
is cross-checked in this script against an independent implementation, scikit-learn's
``GaussianProcessRegressor(optimizer=None)`` -- the script aborts if they disagree by more than
1e-9 (mean, relative) / 1e-10 (variance, absolute) / 1e-8 (NLML, absolute).
Matern12 / Exponential differ from scikit-learn by PyMC's ``sqrt(r^2 + 1e-12)`` and are checked
at 2e-6 instead; the composite (linear x coregion x multi-output) case has no scikit-learn
counterpart and is pinned by the finite-difference gradient test in tests/test_oracle.py.

Usage:  python tests/golden/make_gp_goldens.py
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))

from oracle import gp_oracle as O  # noqa: E402


def sklearn_check(kind, X, y, Xs, ls, eta, sigma, mu, var, nlml):
    from sklearn.gaussian_process import GaussianProcessRegressor as GPR
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern, WhiteKernel

    nu = {"ExpQuad": None, "Matern52": 2.5, "Matern32": 1.5, "Matern12": 0.5}.get(kind, "skip")
    if nu == "skip":
        return None
    base = RBF(ls) if nu is None else Matern(ls, nu=nu)
    kern = ConstantKernel(eta**2, "fixed") * base + WhiteKernel(sigma**2 + O.JITTER, "fixed")
    gpr = GPR(kernel=kern, alpha=0.0, optimizer=None).fit(X, y)
    m, s = gpr.predict(Xs, return_std=True)
    tol = (2e-6, 2e-6, 1e-2) if kind == "Matern12" else (1e-9, 1e-10, 1e-8)
    err = (np.max(np.abs(mu - m)) / np.max(np.abs(m)), np.max(np.abs(var + O.JITTER - s**2)),
           abs(nlml + gpr.log_marginal_likelihood_value_))
    assert err[0] < tol[0] and err[1] < tol[1] and err[2] < tol[2], (kind, err)
    return err


def main():
    out = {}
    rng = np.random.default_rng(2021)
    report = []
    # stationary ARD cases, single output
    for kind, N, d, M in [("ExpQuad", 200, 1, 50), ("ExpQuad", 384, 4, 100), ("Matern52", 300, 8, 80),
                          ("Matern32", 257, 3, 64), ("Matern12", 130, 2, 40), ("Exponential", 128, 2, 40),
                          ("ExpQuad", 512, 5, 128)]:
        X, y, ls = O.synthetic_table(N, d, seed=int(rng.integers(1 << 30)))
        Xs = rng.standard_normal((M, d)) * 1.3
        eta, sigma = 1.0 + 0.3 * rng.random(), 0.15 + 0.2 * rng.random()
        spec = O.make_spec(d, range(d), kind=kind)
        theta = O.pack_theta(spec, ls, eta, sigma)
        mu, var = O.predict(spec, theta, X, y, Xs, with_noise=True)
        mu0, var0 = O.predict(spec, theta, X, y, Xs, with_noise=False)
        val = O.nlml(spec, theta, X, y)
        err = sklearn_check(kind, X, y, Xs, ls, eta, sigma, mu, var, val)
        key = f"{kind}_N{N}_d{d}"
        report.append((key, err))
        out.update({f"{key}/X": X, f"{key}/y": y, f"{key}/Xs": Xs, f"{key}/theta": theta, f"{key}/mu": mu,
                    f"{key}/var": var, f"{key}/var_noiseless": var0, f"{key}/nlml": np.array(val),
                    f"{key}/kind": np.array(O.KINDS[kind]), f"{key}/ard": np.array(1)})
    # shared lengthscale (ARD=False)
    N, d, M = 220, 3, 60
    X, y, _ = O.synthetic_table(N, d, seed=7)
    Xs = rng.standard_normal((M, d))
    spec = O.make_spec(d, range(d), kind="Matern52", ard=False)
    theta = O.pack_theta(spec, [1.3], 0.9, 0.25)
    mu, var = O.predict(spec, theta, X, y, Xs, with_noise=True)
    err = sklearn_check("Matern52", X, y, Xs, 1.3, 0.9, 0.25, mu, var, O.nlml(spec, theta, X, y))
    report.append(("Matern52_iso", err))
    key = "Matern52_iso_N220_d3"
    out.update({f"{key}/X": X, f"{key}/y": y, f"{key}/Xs": Xs, f"{key}/theta": theta, f"{key}/mu": mu,
                f"{key}/var": var, f"{key}/var_noiseless": O.predict(spec, theta, X, y, Xs, with_noise=False)[1],
                f"{key}/nlml": np.array(O.nlml(spec, theta, X, y)), f"{key}/kind": np.array(1), f"{key}/ard": np.array(0)})

    # composite: 2 continuous (1 linear) x 3-level categorical x 2 outputs, heteroskedastic noise
    n = 70
    Xc = rng.standard_normal((n, 2))
    cat = rng.integers(0, 3, n).astype(float)
    X1 = np.column_stack([Xc, cat])
    X = np.vstack([np.column_stack([X1, np.full(n, p)]) for p in range(2)])
    y = rng.standard_normal(2 * n)
    spec = O.make_spec(4, [0, 1], idx_lin=[1], coreg=[(2, 3)], out_col=3, n_out=2, hetero_noise=True)
    theta = O.pack_theta(spec, [0.9, 1.4], 1.2, 0.3, c=[0.2], tau=0.5,
                         coreg=[(rng.standard_normal((3, 2)), [0.4, 0.7, 1.1])],
                         W_out=rng.standard_normal((2, 2)), kappa_out=[0.3, 0.6],
                         W_noise=rng.standard_normal((2, 2)), kappa_noise=[0.5, 0.9])
    Ms = 30
    Xs1 = np.column_stack([rng.standard_normal((Ms, 2)), rng.integers(0, 3, Ms).astype(float)])
    Xs = np.vstack([np.column_stack([Xs1, np.full(Ms, p)]) for p in range(2)])
    mu, var = O.predict(spec, theta, X, y, Xs, with_noise=True)
    val, grad = O.nlml_and_grad(spec, theta, X, y, dist_mode="gemm")
    key = "composite_N140"
    out.update({f"{key}/X": X, f"{key}/y": y, f"{key}/Xs": Xs, f"{key}/theta": theta, f"{key}/mu": mu,
                f"{key}/var": var, f"{key}/var_noiseless": O.predict(spec, theta, X, y, Xs, with_noise=False)[1],
                f"{key}/nlml": np.array(val), f"{key}/grad": grad})
    np.savez_compressed(HERE / "gp_goldens.npz", **out)
    for k, e in report:
        print(k, "sklearn (mean rel, var abs, nlml abs):", e)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
