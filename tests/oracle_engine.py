"""numpy stand-in for ``gumbi_amd.engine.Engine`` built on the CPU oracle.  TEST INFRASTRUCTURE ONLY: the
tests monkeypatch it into ``gumbi_amd.regression.hip_gp`` / ``icm`` to run the host-side logic (plumbing,
priors, MAP loop, result wrapping) without a GPU, and to produce reference values the HIP path is compared
with.  Nothing under ``gumbi_amd/`` imports this."""
import numpy as np

from oracle import gp_oracle as O


class OracleEngine:
    """numpy stand-in for ``gumbi_amd.engine.Engine`` (test infrastructure only)."""

    def __init__(self, device=0, stream=None, sibling_of=None):
        self.X = self.y = self.spec = self.theta = None
        self._factored = False

    def set_data(self, X, y):
        self.X, self.y = np.asarray(X, float), np.asarray(y, float)
        self._factored = False

    def set_y(self, y):
        self.y = np.asarray(y, float)
        self._factored = False

    def factor_is_current(self):
        return self._factored

    def set_kernel(self, spec):
        self.spec = spec.as_dict()

    def set_theta(self, theta):
        self.theta = np.asarray(theta, float).copy()
        self._factored = False

    def factorize(self):
        O.factorize(self.spec, self.theta, self.X, self.y, dist_mode="direct")  # raises LinAlgError if not PD
        self._factored = True

    def nlml(self, grad=False):
        if not grad:
            return O.nlml(self.spec, self.theta, self.X, self.y, dist_mode="direct")
        return O.nlml_and_grad(self.spec, self.theta, self.X, self.y, dist_mode="direct")

    def predict(self, Xs, with_noise=True):
        return O.predict(self.spec, self.theta, self.X, self.y, np.asarray(Xs, float), with_noise=with_noise,
                         dist_mode="direct")

    def copy_alpha(self):
        from scipy.linalg import solve_triangular

        L, v = O.factorize(self.spec, self.theta, self.X, self.y, dist_mode="direct")
        return solve_triangular(L, v, lower=True, trans="T")

    def close(self):
        pass


class HostBaselineEngine(OracleEngine):
    """The stand-in ``bench.py --cpu-fit`` times as the host baseline: PyMC's own distance expansion (``dist_mode="gemm"``, the
    form pm.gp.cov.Stationary.square_dist computes and differentiates) and ONE factorisation per objective evaluation
    (``evaluate`` -- what a compiled PyTensor value-and-gradient function does; Sigma^-1 from the factor by LAPACK dpotri),
    instead of the parity tests' direct differences and their separate factorize / nlml calls."""

    host_blas_free = False  # its arithmetic IS the host's BLAS: find_MAP must not throttle it to one thread

    def evaluate(self, theta, grad=True):
        self.set_theta(theta)
        if not grad:
            return O.nlml(self.spec, self.theta, self.X, self.y, dist_mode="gemm")
        out = O.nlml_and_grad(self.spec, self.theta, self.X, self.y, dist_mode="gemm", inverse="potri")
        self._factored = True
        return out

    def factorize(self):
        O.factorize(self.spec, self.theta, self.X, self.y, dist_mode="gemm")
        self._factored = True

    def predict(self, Xs, with_noise=True):
        return O.predict(self.spec, self.theta, self.X, self.y, np.asarray(Xs, float), with_noise=with_noise, dist_mode="gemm")
