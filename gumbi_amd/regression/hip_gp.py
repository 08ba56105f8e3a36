"""``GP``: the MI355X-native exact-GP backend behind Gumbi's ``Regressor`` seam.

Same class surface as the reference's default backend ``PymcGP`` (``gumbi/regression/pymc/GP.py``:
``__init__`` :227-249, ``fit`` :255-387, ``build_model`` :468-583, ``find_MAP`` :799-813,
``predict`` :837-849; attributes ``model``, ``gp_dict``, ``MAP``, ``model_specs``), but every
covariance build, Cholesky factorisation, triangular solve and gradient runs in
``libgumbi_hip.so`` (HIP, gfx950) through :mod:`gumbi_amd.engine`.  This module holds only what
PyMC's *model declaration* held: which column feeds which kernel, the priors, the
unconstrained parametrisation and the optimiser loop.  There is no CPU implementation of the
numerics here and no fallback.

Model (non-additive, reference docstring :61-71 and code :711-729, :560-569)::

    K     = (eta^2 k(r; ls) [+ tau Linear(c)]) * prod_dims Coregion(W_dim, kappa_dim) * Coregion_out
    noise = WhiteNoise(sigma) [* Coregion("Output_noise")]
    y ~ N(0, K + noise + 1e-6 I)

Priors (:407, :409, :451-452, :460-461, :560): ls ~ InverseGamma(alpha, beta) from the pairwise
distance rule; eta ~ Gamma(2,1); sigma ~ Exponential(1); c ~ N(0,10); tau ~ HalfNormal(10);
W ~ N(0,3) with seeded initial value; kappa ~ Gamma(1.5,1).  ``find_MAP`` minimises, like
``pm.find_MAP`` of PyMC >= 4, ``-(log-lik + log-priors)`` over the log of the positive parameters
with L-BFGS-B (scipy -- the optimiser PyMC itself calls).  The log-Jacobians of the log transforms
are NOT part of that objective (``model.compile_logp(jacobian=False)`` in pymc/tuning/starting.py;
confirmed against the predictions printed in the reference's Multioutput_Regression notebook, see
``HipGP.map_includes_jacobian``).
"""

from __future__ import annotations

import warnings
from math import lgamma

import numpy as np

from ..aggregation import DataSet
from ..engine import KERNEL_KINDS, Engine, KernelSpec
from . import icm
from ..utils.gp_utils import get_ls_prior
from ..utils.misc import assert_in
from .base import Regressor

__all__ = ["HipGP"]

_REJECTED = 1e100  # objective value of a rejected step (non-PD covariance); far above any real NLML


class HipModel:
    """What ``gp.model`` exposes: the declared structure (kernel spec, priors, parameter
    layout).  Plays the descriptive role of the ``pm.Model`` the reference stores."""

    def __init__(self, spec: KernelSpec, ls_params: dict, blocks: list, X, y):
        self.spec = spec
        self.ls_params = ls_params
        self.blocks = blocks  # [(name, kind, shape, slice)], kind in {"pos", "real"}
        self.X = X
        self.y = y

    @property
    def named_vars(self):
        return [b[0] for b in self.blocks]

    def __repr__(self):
        names = ", ".join(f"{n}{shape}" for n, _, shape, _ in self.blocks)
        return f"HipModel(kernel={self.spec.kind}, N={len(self.y)}, D={self.X.shape[1]}, params=[{names}])"


_BLAS_CONTROLLER = None
_BLAS_LIBS_SEEN = -1


def _single_threaded_host_blas(engine):
    """Context for the optimiser's loop when the evaluations run on the GPU: scipy's L-BFGS-B factors its m x m (m <= 10)
    matrices with LAPACK, and a many-threaded OpenBLAS spends 0.1 - 20 ms per iteration waking its pool for them -- more than a
    whole MAP evaluation at Gumbi's usual sizes (N = 392: 0.27 ms).  One thread inside ``minimize`` only, and only for engines
    whose arithmetic is not the host's BLAS (``host_blas_free``).

    The limit is PROCESS-GLOBAL while the context is open (that is how threadpoolctl works): another thread doing host BLAS work
    during ``find_MAP`` is throttled too.  ``GUMBI_HOST_BLAS_LIMIT=0`` switches it off.  The library scan is cached and redone
    when the number of shared objects in the process changed -- a BLAS loaded after the first fit (torch's bundled copy, a
    second numpy / scipy one) is then covered as well; counting the objects costs ~20 us, a scan ~2 ms."""
    import contextlib
    import os

    if not getattr(engine, "host_blas_free", False) or os.environ.get("GUMBI_HOST_BLAS_LIMIT", "1") == "0":
        return contextlib.nullcontext()
    global _BLAS_CONTROLLER, _BLAS_LIBS_SEEN
    try:
        n_libs = _loaded_shared_objects()
        if _BLAS_CONTROLLER is None or n_libs != _BLAS_LIBS_SEEN:
            from threadpoolctl import ThreadpoolController

            _BLAS_CONTROLLER = ThreadpoolController()
            _BLAS_LIBS_SEEN = n_libs
        return _BLAS_CONTROLLER.limit(limits=1, user_api="blas")
    except Exception:  # noqa: BLE001 -- no threadpoolctl, or a BLAS it cannot steer: the fit is merely slower
        return contextlib.nullcontext()


def _loaded_shared_objects() -> int:
    """How many shared objects the process has mapped (dl_iterate_phdr): changes when a library -- e.g. another BLAS -- is loaded."""
    import ctypes

    global _PHDR_COUNTER
    try:
        if _PHDR_COUNTER is None:
            cb_t = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)
            count = [0]

            def _cb(_info, _size, _data):
                count[0] += 1
                return 0

            libc = ctypes.CDLL(None)
            libc.dl_iterate_phdr.argtypes = [cb_t, ctypes.c_void_p]
            _PHDR_COUNTER = (libc, cb_t(_cb), count)
        libc, cb, count = _PHDR_COUNTER
        count[0] = 0
        libc.dl_iterate_phdr(cb, None)
        return count[0]
    except Exception:  # noqa: BLE001 -- no dl_iterate_phdr: behave as before (one scan per process)
        return 0


_PHDR_COUNTER = None


class HipGP(Regressor):
    """Gaussian-process regression on an MI355X.  Drop-in for ``gumbi.GP`` on the
    ``fit() / prepare_grid() / predict_grid()`` path; see the module docstring."""

    def __init__(self, dataset: DataSet, outputs=None, seed=2021, device=0, distributed=None, kronecker="auto"):
        """``distributed``: ``True`` (default process group) or a ``torch.distributed`` group -- ONE GP
        spread over the group's GPUs (:class:`gumbi_amd.distributed.DistributedEngine`: block-cyclic
        Cholesky, sharded gradient and prediction).  Every rank constructs the same GP on the same data
        and makes the same calls; all ranks obtain identical MAP estimates and predictions."""
        super().__init__(dataset, outputs, seed)
        self.device = device
        self.distributed = distributed
        #: "auto": multi-output tables that repeat the same inputs for every output are evaluated in
        #: Kronecker form (:mod:`gumbi_amd.regression.icm`: P systems of size N instead of one of size
        #: PN); False: always the stacked system, as the reference builds it
        self.kronecker = kronecker
        self.model = None
        self.gp_dict = None
        self.MAP = None
        self.trace = None
        self.engine = None

        self.continuous_kernel = "ExpQuad"
        self.heteroskedastic_inputs = False
        self.heteroskedastic_outputs = True
        self.sparse = False
        self.latent = False
        self.n_u = 100
        self.model_specs = {
            "seed": self.seed,
            "continuous_kernel": self.continuous_kernel,
            "heteroskedastic_inputs": self.heteroskedastic_inputs,
            "heteroskedastic_outputs": self.heteroskedastic_outputs,
            "sparse": self.sparse,
            "n_u": self.n_u,
        }
        self.nlml_trace = []
        self._rejected = 0
        self._last_eval_theta = None
        self.n_refactor = 0
        self._theta_fitted = None

    # ------------------------------------------------------------------------------------------
    def fit(self, outputs=None, linear_dims=None, continuous_dims=None, continuous_levels=None,
            continuous_coords=None, categorical_dims=None, categorical_levels=None, additive=False,
            seed=None, continuous_kernel="ExpQuad", period=None, heteroskedastic_inputs=False,
            heteroskedastic_outputs=True, sparse=False, n_u=100, ARD=True, ls_bounds=None, mass=0.98,
            spec_kwargs=None, build_kwargs=None, MAP_kwargs=None):
        """``specify_model`` -> ``build_model`` -> ``find_MAP`` (reference :255-387)."""
        self.specify_model(outputs=outputs, linear_dims=linear_dims, continuous_dims=continuous_dims,
                           continuous_levels=continuous_levels, continuous_coords=continuous_coords,
                           categorical_dims=categorical_dims, categorical_levels=categorical_levels,
                           additive=additive, **(spec_kwargs or {}))
        self.build_model(seed=seed, continuous_kernel=continuous_kernel, period=period,
                         heteroskedastic_inputs=heteroskedastic_inputs,
                         heteroskedastic_outputs=heteroskedastic_outputs, sparse=sparse, n_u=n_u, ARD=ARD,
                         ls_bounds=ls_bounds, mass=mass, **(build_kwargs or {}))
        self.find_MAP(**(MAP_kwargs or {}))
        return self

    # -- helpers ---------------------------------------------------------------------------------------
    def _get_dim_indexes(self):
        dims = self.dims
        return {
            "l": [dims.index(d) for d in self.linear_dims],
            "s": [dims.index(d) for d in self.continuous_dims],
            "c": [dims.index(d) for d in self.categorical_dims],
            "p": dims.index(self.out_col) if self.out_col in dims else None,
        }

    def _prepare_lengthscales(self, X, *, ARD, ls_bounds=None, mass=0.98):
        X_s = X[:, self._get_dim_indexes()["s"]]
        lower = upper = None
        if ls_bounds is not None:
            zb = []
            for dim in self.continuous_dims:
                if dim in ls_bounds.names:
                    vals = np.atleast_1d(ls_bounds[dim].z.values().squeeze())
                    zb.append([None if np.isnan(b) else float(b) for b in vals])
            lower, upper = (list(t) for t in zip(*zb))
            if not ARD and (len(lower) != 1 or len(upper) != 1):
                raise ValueError("Bounds must be specified for only a single dimension if ARD is False")
        return get_ls_prior(X_s, ARD=ARD, lower=lower, upper=upper, mass=mass, device=self.device)

    # -- model declaration ---------------------------------------------------------------------------------
    def build_model(self, seed=None, continuous_kernel="ExpQuad", period=None, heteroskedastic_inputs=False,
                    heteroskedastic_outputs=True, sparse=False, n_u=100, ARD=True, ls_bounds=None, mass=0.98):
        """Declare the marginal GP (reference :468-583) and move the data into HBM."""
        if heteroskedastic_inputs:
            raise NotImplementedError("Heteroskedasticity over inputs is not yet implemented.")
        if sparse:
            raise NotImplementedError("The sparse (FITC) approximation is not part of the HIP backend.")
        periodic = [k + "+Periodic" for k in KERNEL_KINDS] + ["Periodic"]
        assert_in("Continuous kernel", continuous_kernel, list(KERNEL_KINDS) + periodic)
        if continuous_kernel in periodic:
            if period is None and continuous_kernel != "Periodic":
                raise ValueError("Period must be specified for periodic kernel")
            raise NotImplementedError("Periodic kernels are not part of the HIP backend.")

        X, y = self.get_shaped_data("mean")
        D_in = len(self.dims)
        assert X.shape[1] == D_in
        if not self.continuous_dims:
            raise ValueError("At least one continuous dimension is required")

        seed = self.seed if seed is None else seed
        self.seed = seed
        self.continuous_kernel = continuous_kernel
        self.heteroskedastic_inputs = heteroskedastic_inputs
        self.heteroskedastic_outputs = heteroskedastic_outputs
        self.sparse, self.n_u, self.latent = sparse, n_u, False
        self.model_specs = {"seed": seed, "continuous_kernel": continuous_kernel,
                            "heteroskedastic_inputs": heteroskedastic_inputs,
                            "heteroskedastic_outputs": heteroskedastic_outputs, "sparse": sparse, "n_u": n_u}
        self._ARD = ARD

        idx = self._get_dim_indexes()
        multi = self.out_col in self.categorical_dims
        coreg = [(i, len(self.categorical_levels[d])) for d, i in zip(self.categorical_dims, idx["c"])
                 if d != self.out_col]
        spec = KernelSpec(D=D_in, idx_cont=idx["s"], kind=continuous_kernel, ard=ARD, idx_lin=idx["l"],
                          coreg=coreg, out_col=idx["p"] if multi else -1,
                          n_out=len(self.categorical_levels[self.out_col]) if multi else 0,
                          hetero_noise=bool(heteroskedastic_outputs), jitter=1e-6,
                          additive=bool(self.additive and coreg))
        ls_params = self._prepare_lengthscales(X, ARD=ARD, ls_bounds=ls_bounds, mass=mass)

        # parameter blocks in the packing order of include/gumbi_hip.h
        blocks, k = [], 0

        def add(name, kind, shape):
            nonlocal k
            n = int(np.prod(shape)) if shape else 1
            blocks.append((name, kind, shape, slice(k, k + n)))
            k += n

        n_ls = len(idx["s"]) if ARD else 1
        add("ls_total", "pos", (n_ls,))
        add("η_total", "pos", ())
        add("σ", "pos", ())
        if idx["l"]:
            add("c_total", "real", (len(idx["l"]),))
            add("τ_total", "pos", ())
        for d, (_, L) in zip([d for d in self.categorical_dims if d != self.out_col], coreg):
            add(f"W_{d}", "real", (L, 2))
            add(f"κ_{d}", "pos", (L,))
        if multi:
            P = spec.n_out
            add(f"W_{self.out_col}", "real", (P, 2))
            add(f"κ_{self.out_col}", "pos", (P,))
            if spec.hetero_noise:
                add("W_Output_noise", "real", (P, 2))
                add("κ_Output_noise", "pos", (P,))
        if spec.additive:
            # one more continuous (+ linear) kernel per categorical dimension (reference :732-754),
            # appended in the order of include/gumbi_hip.h
            for d in [d for d in self.categorical_dims if d != self.out_col]:
                add(f"ls_{d}", "pos", (n_ls,))
                add(f"η_{d}", "pos", ())
                if idx["l"]:
                    add(f"c_{d}", "real", (len(idx["l"]),))
                    add(f"τ_{d}", "pos", ())
        assert k == spec.theta_size()

        self.model = HipModel(spec, ls_params, blocks, X, y)
        if self.engine is not None:
            self.engine.close()
        if self.distributed:
            from ..distributed import DistributedEngine

            self.engine = DistributedEngine(self.device, None if self.distributed is True else self.distributed)
        elif self.kronecker and spec.out_col >= 0 and not spec.additive and icm.aligned_outputs(X, spec):
            self.engine = icm.IcmEngine(device=self.device)
        else:
            self.engine = Engine(device=self.device)
        self.engine.set_data(X, y)
        self.engine.set_kernel(spec)
        self.gp_dict = {"total": self.engine}
        self.MAP = None
        self._theta_fitted = None
        return self

    # -- priors, transforms ------------------------------------------------------------------------------------------
    def _initial_theta(self):
        """PyMC's default starting point: the moment of each prior, and the seeded draw for W
        (``initval`` at reference :459-460)."""
        m = self.model
        theta = np.zeros(m.spec.theta_size())
        for name, _kind, shape, sl in m.blocks:
            if name.startswith("ls_"):
                a, b = np.asarray(m.ls_params["alpha"]), np.asarray(m.ls_params["beta"])
                theta[sl] = np.where(a > 1, b / np.maximum(a - 1, 1e-300), b / (a + 1))
            elif name.startswith("η_"):
                theta[sl] = 2.0
            elif name == "σ":
                theta[sl] = 1.0
            elif name.startswith("c_"):
                theta[sl] = 0.0
            elif name.startswith("τ_"):
                theta[sl] = 10.0
            elif name.startswith("W_"):
                theta[sl] = np.random.default_rng(self.seed).standard_normal(size=shape).ravel()
            elif name.startswith("κ_"):
                theta[sl] = 1.5
        return theta

    def _log_prior(self, theta):
        """(value, d/dtheta) of the sum of log-prior densities on the natural scale."""
        m = self.model
        val, grad = 0.0, np.zeros_like(theta)
        for name, _kind, _shape, sl in m.blocks:
            x = theta[sl]
            if name.startswith("ls_"):
                a, b = np.asarray(m.ls_params["alpha"], float), np.asarray(m.ls_params["beta"], float)
                val += np.sum(a * np.log(b) - np.array([lgamma(v) for v in a]) - (a + 1) * np.log(x) - b / x)
                grad[sl] = -(a + 1) / x + b / x**2
            elif name.startswith("η_"):  # Gamma(2, 1)
                val += np.sum(np.log(x) - x)
                grad[sl] = 1.0 / x - 1.0
            elif name == "σ":  # Exponential(1)
                val += np.sum(-x)
                grad[sl] = -1.0
            elif name.startswith("c_"):  # Normal(0, 10)
                val += np.sum(-0.5 * np.log(2 * np.pi) - np.log(10.0) - 0.5 * (x / 10.0) ** 2)
                grad[sl] = -x / 100.0
            elif name.startswith("τ_"):  # HalfNormal(10)
                val += np.sum(0.5 * np.log(2 / np.pi) - np.log(10.0) - 0.5 * (x / 10.0) ** 2)
                grad[sl] = -x / 100.0
            elif name.startswith("W_"):  # Normal(0, 3)
                val += np.sum(-0.5 * np.log(2 * np.pi) - np.log(3.0) - 0.5 * (x / 3.0) ** 2)
                grad[sl] = -x / 9.0
            elif name.startswith("κ_"):  # Gamma(1.5, 1)
                val += np.sum(-lgamma(1.5) + 0.5 * np.log(x) - x)
                grad[sl] = 0.5 / x - 1.0
        return float(val), grad

    def _positive_mask(self):
        mask = np.zeros(self.model.spec.theta_size(), dtype=bool)
        for _name, kind, _shape, sl in self.model.blocks:
            if kind == "pos":
                mask[sl] = True
        return mask

    #: PyMC >= 4's ``find_MAP`` compiles ``model.logp(jacobian=False)`` (pymc/tuning/starting.py): the
    #: log-Jacobians of the log transforms are NOT in the objective.  True restates PyMC3.
    map_includes_jacobian = False

    def _objective(self, u, pos):
        """-(log-lik + log-prior [+ log|d theta/du|, see ``map_includes_jacobian``]) and its gradient
        w.r.t. the unconstrained u (u = log theta for the positive parameters)."""
        theta = np.where(pos, np.exp(np.clip(u, -700, 700)), u)
        eng = self.engine
        # from here on the engine's factor belongs to THIS theta (or to none): an evaluation that is rejected further
        # down must not leave `_last_eval_theta` pointing at an earlier point whose factor is gone
        self._last_eval_theta = None
        try:
            fused = getattr(eng, "evaluate", None)  # the single-GPU engine: one C call, no host round trip in between
            if fused is not None:
                nlml, g_nlml = fused(theta)
            else:
                eng.set_theta(theta)
                eng.factorize()
                nlml, g_nlml = eng.nlml(grad=True)
        except np.linalg.LinAlgError:  # covariance not positive definite at this theta: reject the step
            self._rejected += 1
            return _REJECTED, np.zeros_like(u)
        except ValueError as err:
            # parameters the engine refuses on their VALUE (a length scale or kappa that underflowed to zero,
            # an overflow to inf) are rejected steps as well; anything else -- call order, packing, shapes --
            # is a bug and must surface
            if "must be positive" in str(err) or "not finite" in str(err):
                self._rejected += 1
                return _REJECTED, np.zeros_like(u)
            raise
        lp, g_lp = self._log_prior(theta)
        J = 1.0 if self.map_includes_jacobian else 0.0
        f = nlml - lp - J * np.sum(u[pos])
        g = g_nlml - g_lp
        g = np.where(pos, g * theta - J, g)
        if not np.isfinite(f):
            self._rejected += 1
            return _REJECTED, np.zeros_like(u)
        self.nlml_trace.append(float(nlml))
        self._last_eval_theta = theta.copy()
        return float(f), g

    # -- fitting -----------------------------------------------------------------------------------------------------------
    def find_MAP(self, *args, start=None, theta=None, maxeval=5000, method="L-BFGS-B", progressbar=False,
                 **kwargs):
        """Maximum a posteriori hyper-parameters (reference :799-813, ``pm.find_MAP``).

        ``theta`` (natural-scale vector in ``include/gumbi_hip.h`` order, or a dict keyed by the
        MAP names) skips the optimisation and just installs the given hyper-parameters.
        ``start`` is a dict of natural-scale starting values keyed by MAP names."""
        assert self.model is not None, "build_model must be called before find_MAP"
        from scipy.optimize import minimize

        pos = self._positive_mask()
        theta0 = self._initial_theta()
        if start is not None:
            theta0 = self._theta_from_dict(start, theta0)
        self.nlml_trace = []
        self._rejected = 0
        self._last_eval_theta = None
        if theta is not None:
            th = self._theta_from_dict(theta, theta0) if isinstance(theta, dict) else np.asarray(theta, float)
            self.n_eval = 0
        else:
            u0 = theta0.copy()
            u0[pos] = np.log(theta0[pos])
            with _single_threaded_host_blas(self.engine):
                res = minimize(self._objective, u0, args=(pos,), jac=True, method=method,
                               options={"maxfun": int(maxeval), **kwargs.pop("options", {})}, **kwargs)
            th = np.where(pos, np.exp(res.x), res.x)
            self.n_eval = int(res.nfev)
            self.opt_result = res
            # pm.find_MAP reports what the optimiser did; a fit that never left the rejection plateau (every
            # evaluation non-PD: L-BFGS-B sees a zero gradient and "converges" at the start) or that scipy
            # flags as failed must not pass silently
            if res.fun >= _REJECTED or not self.nlml_trace:
                warnings.warn("find_MAP: no admissible evaluation -- the covariance was not positive definite at "
                              f"every one of the {self.n_eval} points tried; MAP is the starting point",
                              RuntimeWarning, stacklevel=2)
            elif not res.success and res.status != 1:  # status 1 = evaluation budget (maxeval) reached
                warnings.warn(f"find_MAP: L-BFGS-B stopped without converging ({res.message})", RuntimeWarning,
                              stacklevel=2)
        # leave the engine factorised at the MAP so predict() reuses the resident factor.  When the optimiser's
        # last evaluation was AT the MAP (the usual L-BFGS-B ending) that factorisation is still resident -- the
        # gradient puts the factor back together -- and nothing is recomputed (the reference factorises again
        # in every predict call, pymc/GP.py:845-847)
        current = getattr(self.engine, "factor_is_current", lambda: False)
        if self._last_eval_theta is not None and np.array_equal(th, self._last_eval_theta) and current():
            self.n_refactor = 0
        else:
            self.engine.set_theta(th)
            self.engine.factorize()
            self.n_refactor = 1
        self._theta_fitted = th
        self.MAP = self._theta_to_dict(th)
        return self.MAP

    def _theta_to_dict(self, theta):
        out = {}
        for name, kind, shape, sl in self.model.blocks:
            val = np.array(theta[sl]).reshape(shape) if shape else np.array(theta[sl][0])
            out[name] = val
            if kind == "pos":
                out[name + "_log__"] = np.log(val)
        return out

    def _theta_from_dict(self, dct, base):
        theta = np.array(base, dtype=float)
        for name, _kind, _shape, sl in self.model.blocks:
            if name in dct:
                theta[sl] = np.asarray(dct[name], dtype=float).ravel()
            elif name + "_log__" in dct:
                theta[sl] = np.exp(np.asarray(dct[name + "_log__"], dtype=float).ravel())
        return theta

    def sample(self, *args, **kwargs):
        raise NotImplementedError("NUTS sampling of hyper-parameters is not part of the HIP backend.")

    # -- prediction ------------------------------------------------------------------------------------------------------------
    def predict(self, points_array, with_noise=True, additive_level="total", **kwargs):
        """Posterior mean / variance at standardized points (reference :837-849): one call into
        ``gmb_predict`` against the factor left resident by :meth:`find_MAP`."""
        if additive_level != "total":
            raise NotImplementedError("Prediction for additive sublevels is not yet supported.")
        if self.MAP is None or self.engine is None:
            raise ValueError("The model must be fit (find_MAP) before predicting")
        return self.engine.predict(np.asarray(points_array, dtype=np.float64), with_noise=with_noise)
