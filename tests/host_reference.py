"""TEST INFRASTRUCTURE: the oracle's arithmetic at full config size (used by tests/test_gpu_configs.py; checked against the
oracle itself in tests/test_oracle.py).  Nothing under ``gumbi_amd/`` imports this."""
import numpy as np

from oracle import gp_oracle as O


def host_posterior_blockwise(spec, theta, X, y, Xs, with_noise=True, block=512, threads=None):
    """The oracle's posterior at FULL config size without the oracle's N x N temporaries: the lower triangle of Sigma filled
    block row by block row (``O.cov_full`` on row blocks, PyMC's distance expansion, a thread pool -- numpy's ufuncs release the
    GIL), ONE in-place LAPACK ``dpotrf``, then v = L^-1 y, A = L^-1 K(X, X*), mean, variance and the NLML (the formulas of
    ``O.predict`` / ``O.nlml``, reference pymc/GP.py:837-849).  Peak host memory ~ 8 N^2 bytes + the workers' blocks."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    import scipy.linalg

    N = len(X)
    S = np.empty((N, N), order="F")
    nz = O.noise_diag(spec, theta, X) + spec["jitter"]

    def fill(r0):
        r1 = min(r0 + block, N)
        S[r0:r1, :r1] = O.cov_full(spec, theta, X[r0:r1], X[:r1], dist_mode="gemm")
        S[np.arange(r0, r1), np.arange(r0, r1)] += nz[r0:r1]

    threads = threads or max(1, min(16, (os.cpu_count() or 2) // 2))
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(fill, range(0, N, block)))
    L = scipy.linalg.cholesky(S, lower=True, overwrite_a=True, check_finite=False)  # (reads and writes the lower triangle only)
    v = scipy.linalg.solve_triangular(L, y, lower=True, check_finite=False)
    A = scipy.linalg.solve_triangular(L, O.cov_full(spec, theta, X, Xs, dist_mode="gemm"), lower=True, check_finite=False)
    mu = A.T @ v
    var = O.cov_diag(spec, theta, Xs) - np.sum(A * A, axis=0)
    if with_noise:
        var = var + O.noise_diag(spec, theta, Xs)
    nlml = 0.5 * N * np.log(2.0 * np.pi) + float(np.sum(np.log(np.diag(L)))) + 0.5 * float(v @ v)
    return mu, var, nlml
