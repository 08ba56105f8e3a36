"""Kronecker (intrinsic coregionalisation) evaluation of the multi-output GP.

When every output is observed at the same N inputs, the stacked covariance the reference hands to
PyMC (``gumbi/regression/pymc/GP.py:711-729`` with the output Coregion of ``:724-727`` and the
"Output_noise" Coregion of ``:565-569``) is

    Sigma = B (x) K + D (x) I,      B = W W^T + diag(kappa)            (P x P, outputs)
                                    K = eta^2 k(X, X) [+ tau lin] ...   (N x N, inputs)
                                    D = diag(sigma^2 Bn[p,p] + jitter)  (P x P, noise)

and with ``D^-1/2 B D^-1/2 = Q Lambda Q^T`` it factors as

    Sigma = (D^1/2 Q (x) I) (Lambda (x) K + I) (Q^T D^1/2 (x) I):

P independent N x N systems ``lambda_p K + I`` instead of one PN x PN system -- P^2 fewer flops and
bytes.  Each of them is exactly what the single-output HIP engine evaluates (amplitude
``eta sqrt(lambda_p)``, linear variance ``tau lambda_p``, unit noise, rotated observations), so this
module is host-side algebra around ``gumbi_amd.engine.Engine``: no new kernels.

    NLML      = sum_p f_p + N/2 sum_a log D_aa,      f_p = engine NLML of system p
    d/d ls, c, other tables = sum_p d f_p
    d/d eta   = sum_p sqrt(lambda_p) d f_p / d eta_p ;   d/d tau = sum_p lambda_p d f_p / d tau_p
    d/d B_ab  = 1/2 (T_ab - alpha_a^T K alpha_b),  T = c diag(t) c^T,  t_p = tr((lambda_p K + I)^-1 K)
    d/d D_aa  = 1/2 (sum_p c_ap^2 s_p - |alpha_a|^2),  s_p = tr((lambda_p K + I)^-1),  c = D^-1/2 Q

(``alpha_a = sum_p c_ap alpha~_p``; ``K alpha~_p = (y~_p - alpha~_p) / lambda_p``; ``s_p`` comes out of
the engine's sigma gradient, ``t_p = (N - s_p) / lambda_p``) -- no eigenvector derivatives are needed
because only traces of the ORIGINAL expression are evaluated in the rotated basis.

Prediction at (x*, output a):  mean = sqrt(D_aa) sum_p Q_ap mean_p(x*),
                               var  = D_aa sum_p Q_ap^2 var_p(x*)  [+ sigma^2 Bn_aa with noise].

``tests/test_gpu_frontend.py`` checks value, gradient and predictions against the stacked engine.
"""

from __future__ import annotations

import numpy as np

from ..engine import Engine, KernelSpec


def aligned_outputs(X, spec) -> bool:
    """True when the stacked table (output-major, task index in the last column) repeats the same
    inputs for every output -- the precondition of the Kronecker form."""
    X = np.asarray(X)
    P = int(spec.n_out)
    if spec.out_col != X.shape[1] - 1 or P < 2 or len(X) % P:
        return False
    N = len(X) // P
    blocks = X.reshape(P, N, X.shape[1])
    if not np.array_equal(blocks[:, :, -1], np.repeat(np.arange(P, dtype=X.dtype)[:, None], N, axis=1)):
        return False
    return bool(np.all(blocks[:, :, :-1] == blocks[0:1, :, :-1]))


class IcmEngine:
    """Same calls as :class:`gumbi_amd.engine.Engine` (``set_data / set_kernel / set_theta / factorize /
    nlml / predict / close``) for an aligned multi-output table; ``theta`` keeps the stacked model's layout."""

    #: the evaluations do not run on the host's BLAS (HipGP.find_MAP keeps OpenBLAS to one thread inside the optimiser's loop)
    host_blas_free = True

    #: device memory the P resident systems may take together (factor + gradient workspace each); above it ONE
    #: inner engine serves the systems in turn and every switch re-factorises
    RESIDENT_BYTES = 160e9

    def __init__(self, device=0, stream=None):
        self.eng = Engine(device=device, stream=stream)
        self._engs = [self.eng]  # one inner engine per system when they fit (siblings: they share eng's streams)
        self._slots = {}         # engine index -> what it holds: dict(p, y, theta)
        self.X = self.y = self.spec = self.theta = None
        self._state = None       # decomposition of the current theta
        self._per_system = False

    # -- declaration ------------------------------------------------------------------------------------------
    def set_data(self, X, y):
        self.X = np.ascontiguousarray(X, dtype=np.float64)
        self.y = np.ascontiguousarray(y, dtype=np.float64)
        self._state = None
        self._slots = {}

    def set_kernel(self, spec: KernelSpec):
        if not aligned_outputs(self.X, spec):
            raise ValueError("the Kronecker path needs the same inputs for every output")
        self.spec = spec
        self.P = int(spec.n_out)
        self.N = len(self.X) // self.P
        self.Xn = np.ascontiguousarray(self.X[: self.N, :-1])
        self.Y = self.y.reshape(self.P, self.N)
        self.spec_k = KernelSpec(D=spec.D - 1, idx_cont=list(spec.idx_cont), kind=spec.kind, ard=spec.ard,
                                 idx_lin=list(spec.idx_lin), coreg=list(spec.coreg), out_col=-1, n_out=0,
                                 hetero_noise=False, jitter=0.0)
        self.nk = self.spec_k.theta_size()
        n_ls = len(spec.idx_cont) if spec.ard else 1
        self.i_eta, self.i_sigma = n_ls, n_ls + 1
        self.i_tau = n_ls + 2 + len(spec.idx_lin) if spec.idx_lin else -1
        self._state = None
        self._slots = {}
        npad = -(-self.N // 128) * 128
        self._per_system = self.P * 2.0 * 8.0 * npad * (npad + 128) <= self.RESIDENT_BYTES

    def set_theta(self, theta):
        theta = np.asarray(theta, dtype=np.float64)
        if theta.shape != (self.spec.theta_size(),):
            raise ValueError(f"theta must have {self.spec.theta_size()} entries")
        pos = [self.i_eta, self.i_sigma] + ([self.i_tau] if self.i_tau >= 0 else [])
        if np.any(theta[pos] <= 0) or np.any(theta[: self.i_eta] <= 0):
            raise ValueError("lengthscales, eta, sigma and tau must be positive")
        self.theta = theta.copy()
        self._state = None

    # -- decomposition ----------------------------------------------------------------------------------------
    def _decompose(self):
        if self._state is not None:
            return self._state
        P, th, k = self.P, self.theta, self.nk
        W = th[k: k + 2 * P].reshape(P, 2)
        kap = th[k + 2 * P: k + 3 * P]
        if np.any(kap <= 0):
            raise ValueError("kappa must be positive")
        sigma = th[self.i_sigma]
        if self.spec.hetero_noise:
            Wn = th[k + 3 * P: k + 5 * P].reshape(P, 2)
            kn = th[k + 5 * P: k + 6 * P]
            if np.any(kn <= 0):
                raise ValueError("kappa must be positive")
            bn = np.sum(Wn * Wn, axis=1) + kn
        else:
            Wn = kn = None
            bn = np.ones(P)
        B = W @ W.T + np.diag(kap)
        Dd = sigma**2 * bn + self.spec.jitter
        lam, Q = np.linalg.eigh(B / np.sqrt(np.outer(Dd, Dd)))
        c = Q / np.sqrt(Dd)[:, None]
        Yt = c.T @ self.Y
        self._state = dict(W=W, Wn=Wn, bn=bn, B=B, Dd=Dd, lam=lam, Q=Q, c=c, Yt=Yt, sigma=sigma)
        return self._state

    def _theta_p(self, lam_p):
        tp = self.theta[: self.nk].copy()
        tp[self.i_eta] *= np.sqrt(lam_p)
        tp[self.i_sigma] = 1.0
        if self.i_tau >= 0:
            tp[self.i_tau] *= lam_p
        return tp

    def _prepare(self, p, st):
        """The inner engine that holds system p with its inputs, kernel and rotated y in place; also its slot, the system's
        theta and whether it still has to be factorised at it."""
        k = p if self._per_system else 0
        while len(self._engs) <= k:
            self._engs.append(Engine(sibling_of=self.eng))
        eng = self._engs[k]
        slot = self._slots.get(k)
        tp = self._theta_p(st["lam"][p])
        if slot is None:   # inputs and kernel once; the systems differ in y and theta only (gmb_set_y keeps the
            eng.set_data(self.Xn, st["Yt"][p])   # workspaces -- a gmb_set_data per system re-allocated them)
            eng.set_kernel(self.spec_k)
            slot = self._slots[k] = dict(p=p, y=st["Yt"], theta=None)
        elif slot["p"] != p or slot["y"] is not st["Yt"]:
            eng.set_y(st["Yt"][p])
            slot.update(p=p, y=st["Yt"], theta=None)
        stale = slot["theta"] is None or not np.array_equal(slot["theta"], tp) or not eng.factor_is_current()
        return eng, slot, tp, stale

    def _load(self, p, st):
        """The inner engine that holds system p, factorised at the current theta.  With one engine per system
        (they fit: P x (factor + gradient workspace) <= RESIDENT_BYTES) nothing is recomputed while theta stands --
        the gradient leaves the factor intact -- so ``predict`` after ``nlml`` and repeated ``predict`` calls reuse
        the resident factors; otherwise engine 0 serves the systems in turn."""
        eng, slot, tp, stale = self._prepare(p, st)
        if stale:
            eng.set_theta(tp)
            eng.factorize()
            slot["theta"] = tp
        return eng

    def _evaluate(self, p, st):
        """System p's objective and gradient: ONE engine call (gmb_evaluate: factorisation and gradient enqueued back to
        back) when it has to be factorised anyway -- every evaluation of a MAP fit --, else the gradient alone."""
        eng, slot, tp, stale = self._prepare(p, st)
        if stale:
            slot["theta"] = None
            fused = getattr(eng, "evaluate", None)
            if fused is not None:
                f, g = fused(tp)
            else:  # (a stand-in engine of the CPU tests)
                eng.set_theta(tp)
                eng.factorize()
                f, g = eng.nlml(grad=True)
            slot["theta"] = tp
        else:
            f, g = eng.nlml(grad=True)
        return eng, f, g

    # -- evaluation -------------------------------------------------------------------------------------------
    def factorize(self):
        """Validates theta (the P factorisations themselves happen inside ``nlml`` / ``predict``: one
        inner engine serves the P systems in turn)."""
        st = self._decompose()
        if np.any(st["lam"] <= 0) or not np.all(np.isfinite(st["lam"])):
            raise np.linalg.LinAlgError("output covariance is not positive definite")

    def nlml(self, grad: bool = False):
        st = self._decompose()
        P, N, lam, c = self.P, self.N, st["lam"], st["c"]
        const = 0.5 * N * np.sum(np.log(st["Dd"]))
        if not grad:
            val = const
            for p in range(P):
                val += self._load(p, st).nlml()
            return val
        val, gk = const, np.zeros(self.nk)
        s = np.zeros(P)
        At = np.zeros((P, N))
        for p in range(P):
            eng, f, g = self._evaluate(p, st)
            a = eng.copy_alpha()
            val += f
            At[p] = a
            s[p] = g[self.i_sigma] + a @ a
            g = g.copy()
            g[self.i_eta] *= np.sqrt(lam[p])
            if self.i_tau >= 0:
                g[self.i_tau] *= lam[p]
            g[self.i_sigma] = 0.0
            gk += g
        t = (N - s) / lam
        alpha = c @ At                                         # alpha_a = sum_p c_ap alpha~_p
        Kalpha = c @ ((st["Yt"] - At) / lam[:, None])          # K alpha_a
        T = (c * t) @ c.T
        A = alpha @ Kalpha.T
        GB = 0.5 * (T - 0.5 * (A + A.T))
        GD = 0.5 * ((c * c) @ s - np.sum(alpha * alpha, axis=1))
        grad_out = np.zeros_like(self.theta)
        grad_out[: self.nk] = gk
        k = self.nk
        grad_out[k: k + 2 * P] = (2.0 * GB @ st["W"]).ravel()
        grad_out[k + 2 * P: k + 3 * P] = np.diag(GB)
        sigma = st["sigma"]
        grad_out[self.i_sigma] = np.sum(GD * 2.0 * sigma * st["bn"])
        if self.spec.hetero_noise:
            gbn = GD * sigma**2
            grad_out[k + 3 * P: k + 5 * P] = (2.0 * gbn[:, None] * st["Wn"]).ravel()
            grad_out[k + 5 * P: k + 6 * P] = gbn
        return val, grad_out

    def predict(self, Xs, with_noise=True):
        Xs = np.ascontiguousarray(Xs, dtype=np.float64)
        if Xs.ndim != 2 or Xs.shape[1] != self.spec.D:
            raise ValueError(f"points_array must be (M, {self.spec.D}), got {Xs.shape}")
        st = self._decompose()
        M = len(Xs)
        if M == 0:
            return np.empty(0), np.empty(0)
        task = Xs[:, -1].astype(np.int64)
        x = np.ascontiguousarray(Xs[:, :-1])
        mean, var = np.zeros(M), np.zeros(M)
        Q, Dd = st["Q"], st["Dd"]
        for p in range(self.P):
            m_p, v_p = self._load(p, st).predict(x, with_noise=False)
            mean += Q[task, p] * m_p
            var += Q[task, p] ** 2 * v_p
        mean *= np.sqrt(Dd[task])
        var *= Dd[task]
        if with_noise:
            var += st["sigma"] ** 2 * st["bn"][task]
        return mean, var

    def factor_is_current(self):
        """True when every system's factor for the current theta is resident (one engine per system)."""
        if not self._per_system or self._state is None:
            return False
        st = self._state
        for p in range(self.P):
            slot = self._slots.get(p)
            if (slot is None or slot["p"] != p or slot["y"] is not st["Yt"] or slot["theta"] is None
                    or not np.array_equal(slot["theta"], self._theta_p(st["lam"][p])) or not self._engs[p].factor_is_current()):
                return False
        return True

    # -- measurement (bench.py --config c4) -----------------------------------------------------------------------
    def set_profiling(self, on):
        """Per-launch HIP events on every inner engine (resets their totals)."""
        for eng in self._engs:
            eng.set_profiling(on)

    def timings(self):
        """The inner engines' ``gmb_timings`` with the cumulative ``total_*`` counters summed over the P systems (the
        per-call fields are engine 0's)."""
        tms = [eng.timings() for eng in self._engs]
        out = dict(tms[0])
        for key in out:
            if key.startswith("total_"):
                out[key] = type(out[key])(sum(t[key] for t in tms))
        return out

    def close(self):
        for eng in reversed(self._engs):  # siblings first: they borrow engine 0's streams
            eng.close()
        self._engs = [self.eng]
        self._slots = {}
