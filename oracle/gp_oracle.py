"""CPU oracle for the Gumbi GP hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker / reported baseline.  The product path (``gumbi_amd``)
never imports it and has no CPU fallback.

What it restates
----------------
Gumbi (``/root/reference``, v0.4.1) holds none of the GP arithmetic itself: it *declares* a
PyMC model (``gumbi/regression/pymc/GP.py:389-464, 468-583, 652-757``) and asks PyMC to
optimise it (``:799-813``) and to predict (``:837-849``).  The arithmetic lives in the
un-vendored third-party dependency **pymc** (``requirements.txt:6`` ``pymc>=5.3.1``,
``environment.yml:10`` ``pymc>=5.10``; no lock file), which is not installed in the build
container and cannot be.  This file therefore restates PyMC >= 5's *published* algorithm
(``pymc.gp.cov``, ``pymc.gp.Marginal``) in numpy, anchored on Gumbi's call sites.

PARITY STATUS: pinned against PyMC's own output at the precision the reference publishes, not
at 1e-8.  The reference's tests assert no posterior value for this path
(``tests/test_regression.py:146-191`` are smoke tests) and PyMC cannot be run here, but the
reference's notebook ``docs/source/notebooks/examples/Multioutput_Regression.ipynb`` -- run by its
author on PyMC 5 (its first cell prints a pytensor warning) on the package's own example data set
-- prints the predictive means and variances of a MAP fit of the full composite model (RBF +
linear kernel x output coregion x heteroskedastic output noise).  ``tests/test_oracle_notebook.py``
replays those calls with this oracle as the numeric engine and reproduces the printed means to
1e-4 .. 1e-3 in the interior of the grid (<= 6e-3 at its extrapolating ends) and the variances to
<= 5 %; the same replay discriminates the two readings of ``pm.find_MAP``'s objective (see
``log_prior_and_jacobian``).  Everything finer than that precision is pinned by (i) an independent
implementation, scikit-learn's ``GaussianProcessRegressor(optimizer=None)``, on every golden case
(``tests/golden/make_gp_goldens.py``; agreement <= 1e-10), and (ii) finite-difference checks of
the analytic NLML gradient (``tests/test_oracle.py``).

Formulas (PyMC >= 5)
--------------------
* every covariance first slices ``X[:, active_dims]``;
* stationary: ``Xs = X / ls``; ``r2 = clip(-2 Xs Xs'^T + |Xs|^2 + |Xs'|^2, 0, inf)``;
  ``r = sqrt(r2 + 1e-12)``;
* ExpQuad ``exp(-r2/2)``; Matern52 ``(1 + sqrt5 r + 5/3 r^2) exp(-sqrt5 r)``;
  Matern32 ``(1 + sqrt3 r) exp(-sqrt3 r)``; Matern12 ``exp(-r)``; Exponential ``exp(-r/2)``;
  ``diag=True`` of any stationary kernel is exactly 1;
* Linear ``(X-c)(X'-c)^T``; Coregion ``B = W W^T + diag(kappa)``, ``B[int(x_i), int(x'_j)]``;
  WhiteNoise ``sigma^2 I`` (self-covariance only);
* Marginal: ``Sigma = K(X,X) + Noise(X) + 1e-6 I``; ``L = chol(Sigma)``;
  ``log p = -N/2 log 2pi - sum log L_ii - |L^-1 y|^2 / 2``;
* conditional (``diag=True``): ``A = L^-1 K(X,X*)``, ``v = L^-1 y``, ``mu = A^T v``,
  ``var = diag K(X*,X*) - colsum(A^2) (+ diag Noise(X*) if pred_noise)``.

Composition as Gumbi declares it (``pymc/GP.py:711-729``, non-additive model)::

    K = (eta^2 k_cont [+ tau * Linear(c)]) * prod_dims Coregion_dim * Coregion_outputs
    noise = WhiteNoise(sigma) [* Coregion("Output_noise") when multi-output, :565-569]

Parameter vector ``theta`` (natural scale), shared with ``include/gumbi_hip.h``::

    [ ls (n_cont if ard else 1) | eta | sigma |
      c (n_lin) , tau                      -- only if n_lin > 0
      for each coregion dim: W (L x 2 row-major), kappa (L)
      W_out (P x 2), kappa_out (P)         -- only if out_col >= 0
      W_noise (P x 2), kappa_noise (P)     -- only if out_col >= 0 and hetero_noise ]
"""

from __future__ import annotations

import numpy as np

try:  # scipy is present in the image; keep a numpy-only fallback for the solves
    from scipy.linalg import cholesky as _sp_cholesky
    from scipy.linalg import solve_triangular as _sp_solve_triangular
except Exception:  # pragma: no cover
    _sp_cholesky = None
    _sp_solve_triangular = None

KINDS = {"ExpQuad": 0, "Matern52": 1, "Matern32": 2, "Matern12": 3, "Exponential": 4}
JITTER = 1e-6  # pymc.gp.util.JITTER_DEFAULT, used by Marginal


# ----------------------------------------------------------------------------------------
# spec / theta handling
# ----------------------------------------------------------------------------------------
def make_spec(
    D,
    idx_cont,
    kind="ExpQuad",
    ard=True,
    idx_lin=(),
    coreg=(),
    out_col=-1,
    n_out=0,
    hetero_noise=True,
    jitter=JITTER,
    additive=False,
):
    """Plain-dict kernel spec (same field names as ``gumbi_amd.engine.KernelSpec``).

    ``coreg`` is a sequence of ``(column, n_levels)`` for categorical dims other than the
    output column (``pymc/GP.py:716-721``); ``out_col``/``n_out`` describe the output
    coregion (``:724-727``).
    """
    kind_id = KINDS[kind] if isinstance(kind, str) else int(kind)
    return dict(
        D=int(D),
        kind=kind_id,
        ard=bool(ard),
        idx_cont=[int(i) for i in idx_cont],
        idx_lin=[int(i) for i in idx_lin],
        coreg=[(int(c), int(n)) for c, n in coreg],
        out_col=int(out_col),
        n_out=int(n_out),
        hetero_noise=bool(hetero_noise),
        jitter=float(jitter),
        additive=bool(additive),
    )


def _term_block_size(spec):
    """Parameters of ONE continuous(+linear) kernel: ls | eta | [c, tau]."""
    n = (len(spec["idx_cont"]) if spec["ard"] else 1) + 1
    if spec["idx_lin"]:
        n += len(spec["idx_lin"]) + 1
    return n


def theta_size(spec):
    if spec.get("additive"):
        base = dict(spec, additive=False)
        return theta_size(base) + len(spec["coreg"]) * _term_block_size(spec)
    n = (len(spec["idx_cont"]) if spec["ard"] else 1) + 2
    if spec["idx_lin"]:
        n += len(spec["idx_lin"]) + 1
    for _, L in spec["coreg"]:
        n += 3 * L
    if spec["out_col"] >= 0:
        n += 3 * spec["n_out"]
        if spec["hetero_noise"]:
            n += 3 * spec["n_out"]
    return n


def unpack_theta(spec, theta):
    """Split the flat natural-scale vector into named pieces (see module docstring)."""
    theta = np.asarray(theta, dtype=np.float64)
    assert theta.shape == (theta_size(spec),), (theta.shape, theta_size(spec))
    if spec.get("additive"):  # global term, tables and noise; the per-dim kernels follow (additive_terms)
        base = dict(spec, additive=False)
        return unpack_theta(base, theta[: theta_size(base)])
    p = {}
    k = 0
    n_ls = len(spec["idx_cont"]) if spec["ard"] else 1
    p["ls"] = theta[k : k + n_ls]
    k += n_ls
    p["eta"] = theta[k]
    p["sigma"] = theta[k + 1]
    k += 2
    if spec["idx_lin"]:
        nl = len(spec["idx_lin"])
        p["c"] = theta[k : k + nl]
        p["tau"] = theta[k + nl]
        k += nl + 1
    p["coreg"] = []
    for _, L in spec["coreg"]:
        W = theta[k : k + 2 * L].reshape(L, 2)
        kap = theta[k + 2 * L : k + 3 * L]
        p["coreg"].append((W, kap))
        k += 3 * L
    if spec["out_col"] >= 0:
        P = spec["n_out"]
        p["W_out"] = theta[k : k + 2 * P].reshape(P, 2)
        p["kappa_out"] = theta[k + 2 * P : k + 3 * P]
        k += 3 * P
        if spec["hetero_noise"]:
            p["W_noise"] = theta[k : k + 2 * P].reshape(P, 2)
            p["kappa_noise"] = theta[k + 2 * P : k + 3 * P]
            k += 3 * P
    assert k == theta.size
    return p


def pack_theta(spec, ls, eta, sigma, c=None, tau=None, coreg=(), W_out=None, kappa_out=None,
               W_noise=None, kappa_noise=None):
    parts = [np.atleast_1d(np.asarray(ls, float)), [float(eta)], [float(sigma)]]
    if spec["idx_lin"]:
        parts += [np.atleast_1d(np.asarray(c, float)), [float(tau)]]
    for W, kap in coreg:
        parts += [np.asarray(W, float).ravel(), np.asarray(kap, float).ravel()]
    if spec["out_col"] >= 0:
        parts += [np.asarray(W_out, float).ravel(), np.asarray(kappa_out, float).ravel()]
        if spec["hetero_noise"]:
            parts += [np.asarray(W_noise, float).ravel(), np.asarray(kappa_noise, float).ravel()]
    theta = np.concatenate([np.asarray(p, float).ravel() for p in parts])
    assert theta.size == theta_size(spec)
    return theta


def coregion_B(W, kappa):
    """pm.gp.cov.Coregion: ``B = W W^T + diag(kappa)`` (call site ``pymc/GP.py:460-462``)."""
    W = np.asarray(W, float)
    return W @ W.T + np.diag(np.asarray(kappa, float))


# ----------------------------------------------------------------------------------------
# covariance pieces
# ----------------------------------------------------------------------------------------
def square_dist(X, Xs, ls, mode="gemm"):
    """pm.gp.cov.Stationary.square_dist.  ``mode='gemm'`` is PyMC's own expansion
    (``-2 X X'^T + |X|^2 + |X'|^2`` then clip); ``mode='direct'`` sums squared scaled
    differences (what the HIP kernel does) -- they differ by O(eps * |x|^2)."""
    A = np.asarray(X, float) / ls
    B = A if Xs is None else np.asarray(Xs, float) / ls
    if mode == "direct":
        d = A[:, None, :] - B[None, :, :]
        return np.einsum("ijk,ijk->ij", d, d)
    A2 = np.sum(A * A, axis=1)
    B2 = np.sum(B * B, axis=1)
    sqd = -2.0 * (A @ B.T) + (A2[:, None] + B2[None, :])
    return np.clip(sqd, 0.0, np.inf)


def stationary_from_r2(kind, r2):
    """Kernel value from squared scaled distance (pymc.gp.cov.{ExpQuad,Matern52,...}.full)."""
    if kind == 0:
        return np.exp(-0.5 * r2)
    r = np.sqrt(r2 + 1e-12)  # pm.gp.cov.Stationary.euclidean_dist
    if kind == 1:
        s5 = np.sqrt(5.0)
        return (1.0 + s5 * r + 5.0 / 3.0 * np.square(r)) * np.exp(-s5 * r)
    if kind == 2:
        s3 = np.sqrt(3.0)
        return (1.0 + s3 * r) * np.exp(-s3 * r)
    if kind == 3:
        return np.exp(-r)
    if kind == 4:
        return np.exp(-0.5 * r)
    raise ValueError(f"unknown kernel kind {kind}")


def _coreg_factor(spec, p, X, Xs):
    """Elementwise product of all coregion kernels (``pymc/GP.py:716-727``)."""
    F = None
    tables = [(col, coregion_B(W, kap)) for (col, _), (W, kap) in zip(spec["coreg"], p["coreg"])]
    if spec["out_col"] >= 0:
        tables.append((spec["out_col"], coregion_B(p["W_out"], p["kappa_out"])))
    for col, B in tables:
        ci = X[:, col].astype(np.int32)
        cj = ci if Xs is None else Xs[:, col].astype(np.int32)
        G = B[np.ix_(ci, cj)]
        F = G if F is None else F * G
    return F


def additive_terms(spec, theta):
    """Additive model (``specify_model(additive=True)``, ``pymc/GP.py:732-754``): a "global" GP with the
    continuous (+linear) kernel, plus one GP per categorical dim with its OWN continuous (+linear)
    kernel times that dim's Coregion, every term times the output Coregion; ``gp_total`` is their sum,
    i.e. the covariances add.  theta = [theta of the non-additive spec | per categorical dim:
    ls | eta | (c, tau)].  Returns ``[(sub_spec, sub_theta, index_map)]``: each term as a non-additive
    model of this module; ``index_map[i]`` is the position of sub_theta[i] in theta (-1: noise
    parameters, which belong to no term)."""
    theta = np.asarray(theta, float)
    base = dict(spec, additive=False)
    nb = theta_size(base)
    n_ls = len(spec["idx_cont"]) if spec["ard"] else 1
    nk = _term_block_size(spec)
    n_core = n_ls + 2 + ((len(spec["idx_lin"]) + 1) if spec["idx_lin"] else 0)  # ls|eta|sigma|c,tau
    core0 = list(range(n_core))
    coreg_pos, k = [], n_core
    for _, L in spec["coreg"]:
        coreg_pos.append(list(range(k, k + 3 * L)))
        k += 3 * L
    tail = list(range(k, nb))  # W_out, kappa_out [, W_noise, kappa_noise]
    n_out_par = 3 * spec["n_out"] if spec["out_col"] >= 0 else 0
    tail_map = tail[:n_out_par] + [-1] * (len(tail) - n_out_par)
    terms = []
    sub0 = dict(base, coreg=[])
    idx0 = core0 + tail
    m0 = core0[:n_ls + 1] + [-1] + core0[n_ls + 2:] + tail_map
    terms.append((sub0, theta[idx0], m0))
    for j, cd in enumerate(spec["coreg"]):
        off = nb + j * nk
        blk = list(range(off, off + nk))
        core = blk[:n_ls + 1] + [n_ls + 1] + blk[n_ls + 1:]  # splice sigma's slot in
        sub = dict(base, coreg=[cd])
        idx = core + coreg_pos[j] + tail
        mp = blk[:n_ls + 1] + [-1] + blk[n_ls + 1:] + coreg_pos[j] + tail_map
        terms.append((sub, theta[idx], mp))
    return terms


def cov_full(spec, theta, X, Xs=None, dist_mode="gemm"):
    """K(X, X') for the total covariance Gumbi declares (no noise, no jitter)."""
    if spec.get("additive"):
        return sum(cov_full(sp, th, X, Xs, dist_mode) for sp, th, _ in additive_terms(spec, theta))
    p = unpack_theta(spec, theta)
    X = np.asarray(X, float)
    Xs_ = None if Xs is None else np.asarray(Xs, float)
    ic = spec["idx_cont"]
    r2 = square_dist(X[:, ic], None if Xs_ is None else Xs_[:, ic], p["ls"], mode=dist_mode)
    K = p["eta"] ** 2 * stationary_from_r2(spec["kind"], r2)
    if spec["idx_lin"]:
        il = spec["idx_lin"]
        A = X[:, il] - p["c"]
        B = A if Xs_ is None else Xs_[:, il] - p["c"]
        K = K + p["tau"] * (A @ B.T)  # tau * pm.gp.cov.Linear  (pymc/GP.py:453)
    F = _coreg_factor(spec, p, X, Xs_)
    if F is not None:
        K = K * F
    return K


def cov_diag(spec, theta, Xs):
    """diag K(X*, X*): stationary diag is exactly 1 (PyMC ``Stationary.diag``)."""
    if spec.get("additive"):
        return sum(cov_diag(sp, th, Xs) for sp, th, _ in additive_terms(spec, theta))
    p = unpack_theta(spec, theta)
    Xs = np.asarray(Xs, float)
    d = np.full(Xs.shape[0], p["eta"] ** 2)
    if spec["idx_lin"]:
        A = Xs[:, spec["idx_lin"]] - p["c"]
        d = d + p["tau"] * np.sum(A * A, axis=1)
    tables = [(col, coregion_B(W, kap)) for (col, _), (W, kap) in zip(spec["coreg"], p["coreg"])]
    if spec["out_col"] >= 0:
        tables.append((spec["out_col"], coregion_B(p["W_out"], p["kappa_out"])))
    for col, B in tables:
        ci = Xs[:, col].astype(np.int32)
        d = d * np.diag(B)[ci]
    return d


def noise_diag(spec, theta, X):
    """diag Noise(X): ``sigma^2`` or ``sigma^2 * B_noise[p,p]`` (``pymc/GP.py:560-569``)."""
    p = unpack_theta(spec, theta)
    X = np.asarray(X, float)
    d = np.full(X.shape[0], p["sigma"] ** 2)
    if spec["out_col"] >= 0 and spec["hetero_noise"]:
        Bn = coregion_B(p["W_noise"], p["kappa_noise"])
        d = d * np.diag(Bn)[X[:, spec["out_col"]].astype(np.int32)]
    return d


def sigma_matrix(spec, theta, X, dist_mode="gemm"):
    """Sigma = K(X,X) + Noise(X) + jitter*I  (what Marginal factorises)."""
    S = cov_full(spec, theta, X, None, dist_mode=dist_mode)
    S[np.diag_indices_from(S)] += noise_diag(spec, theta, X) + spec["jitter"]
    return S


# ----------------------------------------------------------------------------------------
# factorisation, likelihood, prediction
# ----------------------------------------------------------------------------------------
def cholesky_lower(S):
    if _sp_cholesky is not None:
        return _sp_cholesky(S, lower=True, check_finite=False)
    return np.linalg.cholesky(S)


def solve_lower(L, B):
    if _sp_solve_triangular is not None:
        return _sp_solve_triangular(L, B, lower=True, check_finite=False)
    return np.linalg.solve(L, B)  # pragma: no cover


def factorize(spec, theta, X, y, dist_mode="gemm"):
    S = sigma_matrix(spec, theta, X, dist_mode=dist_mode)
    L = cholesky_lower(S)
    v = solve_lower(L, np.asarray(y, float))
    return L, v


def nlml(spec, theta, X, y, dist_mode="gemm"):
    """Negative log marginal likelihood ``-log N(y; 0, Sigma)``."""
    L, v = factorize(spec, theta, X, y, dist_mode=dist_mode)
    N = len(v)
    return 0.5 * N * np.log(2.0 * np.pi) + np.sum(np.log(np.diag(L))) + 0.5 * float(v @ v)


def predict(spec, theta, X, y, Xs, with_noise=True, dist_mode="gemm"):
    """``pm.gp.Marginal.predict(Xnew, diag=True, pred_noise=with_noise)`` (``pymc/GP.py:845-847``)."""
    L, v = factorize(spec, theta, X, y, dist_mode=dist_mode)
    Kxs = cov_full(spec, theta, X, Xs, dist_mode=dist_mode)
    A = solve_lower(L, Kxs)
    mu = A.T @ v
    var = cov_diag(spec, theta, Xs) - np.sum(A * A, axis=0)
    if with_noise:
        var = var + noise_diag(spec, theta, Xs)
    return mu, var


# ----------------------------------------------------------------------------------------
# analytic gradient of the NLML w.r.t. natural-scale theta
# ----------------------------------------------------------------------------------------
def _stationary_dr2(kind, r2):
    """d k / d r2 for each stationary kernel."""
    if kind == 0:
        return -0.5 * np.exp(-0.5 * r2)
    r = np.sqrt(r2 + 1e-12)
    if kind == 1:
        s5 = np.sqrt(5.0)
        # dk/dr = -5/3 r (1 + s5 r) exp(-s5 r);  dr/dr2 = 1/(2r)
        return -(5.0 / 6.0) * (1.0 + s5 * r) * np.exp(-s5 * r)
    if kind == 2:
        s3 = np.sqrt(3.0)
        return -1.5 * np.exp(-s3 * r)
    if kind == 3:
        return -np.exp(-r) / (2.0 * r)
    if kind == 4:
        return -0.25 * np.exp(-0.5 * r) / r
    raise ValueError(kind)


def _kernel_grad_given_M(spec, theta, X, M, dist_mode="direct"):
    """sum_ij M_ij dK_ij/dtheta for every KERNEL parameter of a non-additive spec (entries of sigma
    and of the noise table stay 0), ``M = dNLML/dSigma``."""
    p = unpack_theta(spec, theta)
    X = np.asarray(X, float)
    N = X.shape[0]
    ic = spec["idx_cont"]
    Xc = X[:, ic]
    r2 = square_dist(Xc, None, p["ls"], mode=dist_mode)
    kst = stationary_from_r2(spec["kind"], r2)
    base = p["eta"] ** 2 * kst
    lin = None
    if spec["idx_lin"]:
        Al = X[:, spec["idx_lin"]] - p["c"]
        lin = Al @ Al.T
        base = base + p["tau"] * lin
    F = _coreg_factor(spec, p, X, None)
    K = base if F is None else base * F
    g = np.zeros_like(np.asarray(theta, float))
    k = 0
    Fm = 1.0 if F is None else F
    MF = M * Fm
    # lengthscales: d r2/d ls_k = -2 (x_k-x'_k)^2 / ls_k^3
    dk = p["eta"] ** 2 * _stationary_dr2(spec["kind"], r2)
    if spec["ard"] and dist_mode == "gemm":
        # The expanded form PyMC's graph differentiates (square_dist = |x|^2 + |x'|^2 - 2 x.x'; reverse mode sends the
        # cotangent G = M * dk through the dot product): sum_ij G_ij (x_ik - x_jk)^2 = sum_i x_ik^2 (R_i + C_i) - 2 x_k^T G x_k
        # with R, C the row and column sums of G -- one N x N x d product instead of d passes over N x N temporaries.
        G = MF * dk
        rc = G.sum(axis=1) + G.sum(axis=0)
        GX = G @ Xc
        for j in range(len(ic)):
            xj = Xc[:, j]
            g[k + j] = -2.0 / p["ls"][j] ** 3 * (float(np.dot(xj * xj, rc)) - 2.0 * float(np.dot(xj, GX[:, j])))
        k += len(ic)
    elif spec["ard"]:
        for j in range(len(ic)):
            dj = Xc[:, j][:, None] - Xc[:, j][None, :]
            g[k + j] = np.sum(MF * dk * (-2.0) * dj * dj / p["ls"][j] ** 3)
        k += len(ic)
    else:
        raw = square_dist(Xc, None, 1.0, mode=dist_mode)
        g[k] = np.sum(MF * dk * (-2.0) * raw / p["ls"][0] ** 3)
        k += 1
    g[k] = np.sum(MF * 2.0 * p["eta"] * kst)  # eta
    k += 1
    k += 1  # sigma: not a kernel parameter
    if spec["idx_lin"]:
        nl = len(spec["idx_lin"])
        for j in range(nl):
            # d/dc_j (x_j - c_j)(x'_j - c_j) = -(x_j - c_j) - (x'_j - c_j)
            aj = Al[:, j]
            g[k + j] = np.sum(MF * p["tau"] * (-(aj[:, None] + aj[None, :])))
        k += nl
        g[k] = np.sum(MF * lin)
        k += 1

    def _coreg_grads(col, W, kap, base_wo):
        """gradients w.r.t. W (L,2) and kappa (L) of a coregion factor on column ``col``."""
        Lc = W.shape[0]
        ci = X[:, col].astype(np.int32)
        G = np.zeros((Lc, Lc))
        np.add.at(G, (ci[:, None].repeat(N, 1), ci[None, :].repeat(N, 0)), M * base_wo)
        # dB = dW W^T + W dW^T + diag(dkappa)  ->  dNLML/dW = (G + G^T) W ; dNLML/dkappa = diag(G)
        return ((G + G.T) @ W).ravel(), np.diag(G).copy()

    tables = list(zip([c for c, _ in spec["coreg"]], p["coreg"]))
    if spec["out_col"] >= 0:
        tables.append((spec["out_col"], (p["W_out"], p["kappa_out"])))
    for col, (W, kap) in tables:
        B = coregion_B(W, kap)
        ci = X[:, col].astype(np.int32)
        this = B[np.ix_(ci, ci)]
        with np.errstate(divide="ignore", invalid="ignore"):
            base_wo = np.where(this != 0.0, K / this, 0.0)
        if np.any(this == 0.0):  # recompute without this factor when a table entry is zero
            others = None
            for col2, (W2, kap2) in tables:
                if col2 == col:
                    continue
                c2 = X[:, col2].astype(np.int32)
                t2 = coregion_B(W2, kap2)[np.ix_(c2, c2)]
                others = t2 if others is None else others * t2
            base_wo = base if others is None else base * others
        gW, gk = _coreg_grads(col, W, kap, base_wo)
        Lc = W.shape[0]
        g[k : k + 2 * Lc] = gW
        g[k + 2 * Lc : k + 3 * Lc] = gk
        k += 3 * Lc
    return g


def nlml_and_grad(spec, theta, X, y, dist_mode="direct", inverse="trsm"):
    """NLML and d NLML / d theta (natural scale), ``dNLML = 1/2 tr((Sigma^-1 - a a^T) dSigma)``.

    PyMC obtains the same derivative by reverse-mode autodiff through its Cholesky op
    (``pm.find_MAP``, call site ``pymc/GP.py:811``); it is restated analytically here.
    ``inverse``: how Sigma^-1 is formed from L -- "trsm" (L^-1 by a triangular solve of the identity, then L^-T L^-1: the form
    the parity tests pin) or "potri" (LAPACK dpotri on the factor: a third of the flops; what bench.py's host baseline times).
    """
    theta = np.asarray(theta, float)
    p = unpack_theta(spec, theta)
    X = np.asarray(X, float)
    y = np.asarray(y, float)
    N = X.shape[0]
    S = sigma_matrix(spec, theta, X, dist_mode=dist_mode)
    L = cholesky_lower(S)
    v = solve_lower(L, y)
    val = 0.5 * N * np.log(2.0 * np.pi) + np.sum(np.log(np.diag(L))) + 0.5 * float(v @ v)
    if inverse == "potri":
        from scipy.linalg import lapack, solve_triangular

        Sinv, info = lapack.dpotri(L, lower=1)
        if info != 0:
            raise np.linalg.LinAlgError(f"dpotri: info = {info}")
        Sinv = np.tril(Sinv)  # (dpotri fills the lower triangle only)
        Sinv = Sinv + np.tril(Sinv, -1).T
        alpha = solve_triangular(L, v, lower=True, trans="T", check_finite=False)
    else:
        Linv = solve_lower(L, np.eye(N))
        Sinv = Linv.T @ Linv
        alpha = Linv.T @ v
    M = 0.5 * (Sinv - np.outer(alpha, alpha))  # dNLML/dSigma_ij

    g = np.zeros_like(theta)
    if spec.get("additive"):
        for sub, th, mp in additive_terms(spec, theta):
            gt = _kernel_grad_given_M(sub, th, X, M, dist_mode)
            for i, dst in enumerate(mp):
                if dst >= 0:
                    g[dst] += gt[i]
        base = dict(spec, additive=False)
        k = theta_size(base)
    else:
        g += _kernel_grad_given_M(spec, theta, X, M, dist_mode)
        k = theta.size
    n_ls = len(spec["idx_cont"]) if spec["ard"] else 1
    i_sigma = n_ls + 1
    # noise
    Md = np.diag(M)
    if spec["out_col"] >= 0 and spec["hetero_noise"]:
        P = spec["n_out"]
        k -= 3 * P
        Wn, kn = p["W_noise"], p["kappa_noise"]
        Bn = coregion_B(Wn, kn)
        pi = X[:, spec["out_col"]].astype(np.int32)
        g[i_sigma] = np.sum(Md * 2.0 * p["sigma"] * np.diag(Bn)[pi])
        Gd = np.zeros(P)
        np.add.at(Gd, pi, Md * p["sigma"] ** 2)
        # only the diagonal of B_noise enters: d diag(B)_a = 2 W_a . dW_a + dkappa_a
        g[k : k + 2 * P] = (2.0 * Gd[:, None] * Wn).ravel()
        g[k + 2 * P : k + 3 * P] = Gd
    else:
        g[i_sigma] = np.sum(Md) * 2.0 * p["sigma"]
    return val, g


# ----------------------------------------------------------------------------------------
# lengthscale-prior limits  (gumbi/utils/gp_utils.py:15-48)
# ----------------------------------------------------------------------------------------
def parse_ls_limits(X, ARD, lower=None, upper=None):
    """Restatement of ``gp_utils.parse_ls_limits`` with an explicit O(N^2) pairwise pass
    (small N only).  ARD: one set of 1-D distances per column; otherwise one joint set."""
    X = np.asarray(X, float)
    groups = [X[:, [j]] for j in range(X.shape[1])] if ARD else [X]

    def _bcast(v):
        if v is None:
            return [None] * len(groups)
        v = list(v) if isinstance(v, (list, tuple, np.ndarray)) else [v]
        if len(v) == 1:
            v = v * len(groups)
        if len(v) != len(groups):
            raise ValueError("Number of bounds must match number of dimensions")
        return v

    lowers, uppers = _bcast(lower), _bcast(upper)
    for i, pts in enumerate(groups):
        n = pts.shape[0]
        iu = np.triu_indices(n, 1)
        d = np.sqrt(np.sum((pts[iu[0]] - pts[iu[1]]) ** 2, axis=1))
        nz = d[d != 0]
        default_lower = nz.min() if nz.size else 0.01
        lo = default_lower if lowers[i] is None else lowers[i]
        lo = max(lo, default_lower, 0.01)
        up = uppers[i]
        if up is None:
            up = nz.max() if nz.size else 1
        lowers[i], uppers[i] = lo, up
    return lowers, uppers


# ----------------------------------------------------------------------------------------
# MAP objective pieces: log-priors (+ optionally the log-Jacobians of the log transforms)
# (priors at pymc/GP.py:407,409,451-452,460-461,560; PyMC optimises log-transformed
# positive variables, so each contributes log|d theta/d u| = u = log theta)
# ----------------------------------------------------------------------------------------
def _lgamma(x):
    from math import lgamma

    return np.vectorize(lgamma)(x)


def logp_inverse_gamma(x, alpha, beta):
    return alpha * np.log(beta) - _lgamma(alpha) - (alpha + 1.0) * np.log(x) - beta / x


def logp_gamma(x, alpha, beta):
    return alpha * np.log(beta) - _lgamma(alpha) + (alpha - 1.0) * np.log(x) - beta * x


def logp_exponential(x, lam):
    return np.log(lam) - lam * x


def logp_normal(x, mu, sd):
    return -0.5 * np.log(2.0 * np.pi) - np.log(sd) - 0.5 * ((x - mu) / sd) ** 2


def logp_halfnormal(x, sd):
    return 0.5 * np.log(2.0 / np.pi) - np.log(sd) - 0.5 * (x / sd) ** 2


def log_prior_and_jacobian(spec, theta, ls_alpha, ls_beta, jacobian=False):
    """Sum of the log-prior densities that ``pm.find_MAP`` adds to the likelihood.

    PyMC >= 4 (the reference requires pymc >= 5.3.1) builds the MAP objective with
    ``model.compile_logp(jacobian=False)`` (``pymc/tuning/starting.py: find_MAP``): the optimiser
    works on the log-transformed variables but the log-Jacobians of the transforms are NOT part of
    the objective, i.e. the optimum is the mode of the density over the natural parameters.
    ``jacobian=True`` restates the PyMC3 behaviour (mode of the density over the transformed ones).
    Evidence beyond the source reading: the reference's Multioutput_Regression notebook (run by its
    author on PyMC 5 -- its first cell shows a pytensor warning) prints predictive means / variances
    that the ``jacobian=False`` objective reproduces to 1e-4 / 4 %, the other one to 5e-3 / 36 %
    (``tests/test_gpu_frontend.py``, ``tools/gpu_notebook_check2.py``)."""
    p = unpack_theta(spec, theta)
    a = np.asarray(ls_alpha, float)
    b = np.asarray(ls_beta, float)
    J = 1.0 if jacobian else 0.0
    tot = np.sum(logp_inverse_gamma(p["ls"], a, b) + J * np.log(p["ls"]))
    tot += logp_gamma(p["eta"], 2.0, 1.0) + J * np.log(p["eta"])
    tot += logp_exponential(p["sigma"], 1.0) + J * np.log(p["sigma"])
    if spec["idx_lin"]:
        tot += np.sum(logp_normal(p["c"], 0.0, 10.0))
        tot += logp_halfnormal(p["tau"], 10.0) + J * np.log(p["tau"])
    pairs = list(p["coreg"])
    if spec["out_col"] >= 0:
        pairs.append((p["W_out"], p["kappa_out"]))
        if spec["hetero_noise"]:
            pairs.append((p["W_noise"], p["kappa_noise"]))
    for W, kap in pairs:
        tot += np.sum(logp_normal(W, 0.0, 3.0))
        tot += np.sum(logp_gamma(kap, 1.5, 1.0) + J * np.log(kap))
    if spec.get("additive"):
        # the per-dimension kernels are built by the same factories (``pymc/GP.py:738-741``), so their
        # ls / eta / c / tau carry the same priors as the global ones
        n_ls = len(spec["idx_cont"]) if spec["ard"] else 1
        theta = np.asarray(theta, float)
        nb, nk = theta_size(dict(spec, additive=False)), _term_block_size(spec)
        for j in range(len(spec["coreg"])):
            blk = theta[nb + j * nk: nb + (j + 1) * nk]
            ls, eta = blk[:n_ls], blk[n_ls]
            tot += np.sum(logp_inverse_gamma(ls, a, b) + J * np.log(ls))
            tot += logp_gamma(eta, 2.0, 1.0) + J * np.log(eta)
            if spec["idx_lin"]:
                tot += np.sum(logp_normal(blk[n_ls + 1:-1], 0.0, 10.0))
                tot += logp_halfnormal(blk[-1], 10.0) + J * np.log(blk[-1])
    return float(tot)


# ----------------------------------------------------------------------------------------
# synthetic benchmark tables -- shared by tests and bench.py
# ----------------------------------------------------------------------------------------
def synthetic_table(N, d, seed=2021, sigma=0.2):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, d))
    ls = np.geomspace(0.7, 2.0, d) if d > 1 else np.array([1.0])
    f = np.sum(np.sin(X / ls), axis=1) / np.sqrt(d)
    y = f + sigma * rng.standard_normal(N)
    y = (y - y.mean()) / y.std(ddof=1)
    return X, y, ls


def synthetic_grid(d, res=100, lim=2.4):
    g = np.linspace(-lim, lim, res)
    if d == 1:
        return g[:, None].copy()
    G0, G1 = np.meshgrid(g, g, indexing="ij")
    Xs = np.zeros((res * res, d))
    Xs[:, 0] = G0.ravel()
    Xs[:, 1] = G1.ravel()
    return Xs
