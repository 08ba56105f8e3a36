"""Named, standardizer-aware structured arrays: the containers ``predict_points`` /
``predict_grid`` take and return.

API-compatible re-implementation of the path subset of ``gumbi/arrays.py``:
``LayeredArray`` (:174-307), ``ParameterArray`` (:310-483), ``UncertainArray`` (:486-858, without
the third-party ``uncertainties`` dependency: first-order propagation for independent normals is
written out here), ``UncertainParameterArray`` (:861-1188) and ``MVUncertainParameterArray``
(:1191-1460).  All of them are numpy structured-array views, so reshape / ravel / indexing and
``np.meshgrid`` / ``np.vstack`` keep working on them exactly as in the reference.
"""

from __future__ import annotations

import warnings

import numpy as np
from scipy.special import expit, logit
from scipy.stats import chi2, lognorm, multivariate_normal, norm

from .aggregation import Standardizer
from .utils.misc import assert_in, identity

__all__ = [
    "LayeredArray",
    "ParameterArray",
    "UncertainArray",
    "UncertainParameterArray",
    "MVUncertainParameterArray",
]

_INT_TYPES = (int, np.integer)


def _is_point_index(item):
    return isinstance(item, _INT_TYPES) or (
        isinstance(item, tuple) and len(item) > 0 and all(isinstance(v, _INT_TYPES) for v in item)
    )


def _structured(fields: dict, shape=None):
    """Pack equally-shaped arrays into one structured ndarray (one field per key)."""
    arrays = {k: np.asarray(v) for k, v in fields.items() if v is not None}
    dtype = np.dtype([(k, a.dtype) for k, a in arrays.items()])
    base = np.empty(next(iter(arrays.values())).shape if shape is None else shape, dtype=dtype)
    for k, a in arrays.items():
        base[k] = a
    return base


# ------------------------------------------------------------------------------------------------
class LayeredArray(np.ndarray):
    """An ndarray holding one or more *named* values ("layers") at every index."""

    def __new__(cls, stdzr=None, **arrays):
        if not arrays:
            raise ValueError("Must supply at least one array")
        obj = _structured(arrays).view(cls)
        obj.names = list(obj.dtype.names)
        obj.stdzr = stdzr
        return obj

    def __array_finalize__(self, src):
        if src is None:
            return
        self.names = getattr(src, "names", None)
        self.stdzr = getattr(src, "stdzr", None)

    # arithmetic is defined for single-layer arrays only and returns the same kind of array
    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        plain = []
        for x in inputs:
            if isinstance(x, LayeredArray):
                if len(x.names) > 1:
                    raise ValueError("Cannot operate on array with multiple layer names")
                plain.append(x.astype(float).view(np.ndarray))
            else:
                plain.append(x)
        # `out=` cannot alias a structured buffer, so the result is always returned fresh (numpy's
        # own reductions, e.g. np.mean, use the returned value)
        result = getattr(ufunc, method)(*plain, **kwargs)
        if result is NotImplemented:
            return NotImplemented
        name = self.names[0]

        def wrap(r):  # comparisons / predicates give plain boolean arrays, arithmetic re-wraps
            return r if np.asarray(r).dtype == bool else self._rewrap(name, r)

        if isinstance(result, tuple):
            return tuple(wrap(r) for r in result)
        return wrap(result)

    # ndarray's rich comparisons special-case structured dtypes before __array_ufunc__ is
    # consulted; route them through the ufuncs so single-layer arrays compare like numbers
    def __eq__(self, other):
        return np.equal(self, other)

    def __ne__(self, other):
        return np.not_equal(self, other)

    def __lt__(self, other):
        return np.less(self, other)

    def __le__(self, other):
        return np.less_equal(self, other)

    def __gt__(self, other):
        return np.greater(self, other)

    def __ge__(self, other):
        return np.greater_equal(self, other)

    __hash__ = None

    def _rewrap(self, name, values):
        return LayeredArray(**{name: values})

    def _from_layers(self, layers: dict):
        return LayeredArray(**layers)

    def __getitem__(self, item):
        picked = np.ndarray.__getitem__(self, item)
        if isinstance(item, str):
            return self._from_layers({item: np.asarray(picked)})
        if _is_point_index(item) or isinstance(item, slice):
            plain = np.asarray(picked)
            return self._from_layers({n: plain[n] for n in plain.dtype.names})
        return picked

    def __repr__(self):
        return f"{tuple(self.names)}: {np.asarray(self)}"

    __str__ = __repr__

    def get(self, name, default=None):
        if name in self.names:
            return self[name]
        if default is None:
            return None
        return self._from_layers({name: default})

    def drop(self, name, missing_ok=True):
        if name in self.names:
            return self._from_layers({n: a for n, a in self.as_dict().items() if n != name})
        if missing_ok:
            return self
        raise KeyError(f"Name {name} not found in array.")

    def values(self):
        """Plain float ndarray; layers stacked along a new leading axis if there are several."""
        plain = np.asarray(self)
        layers = [plain[n].astype(float) for n in self.names]
        return layers[0] if len(layers) == 1 else np.stack(layers)

    def dstack(self):
        plain = np.asarray(self)
        return np.dstack([plain[n].astype(float) for n in self.names])

    def as_list(self, order=None):
        order = self.names if order is None else order
        assert all(n in order for n in self.names)
        return [self[n] for n in order]

    def as_dict(self):
        plain = np.asarray(self)
        return {n: plain[n].astype(float) for n in self.names}

    def add_layers(self, **arrays):
        return self._from_layers({**self.as_dict(), **arrays})


# ------------------------------------------------------------------------------------------------
class ParameterArray(LayeredArray):
    """LayeredArray of natural-scale parameter values that knows how to standardize itself
    (``.z``) or transform itself (``.t``) through its :class:`Standardizer`."""

    def __new__(cls, stdzr: Standardizer = None, stdzd=False, **arrays):
        if not arrays:
            raise ValueError("Must supply at least one array")
        if stdzd:
            arrays = {n: stdzr.unstdz(n, np.array(a)) for n, a in arrays.items()}
        obj = _structured(arrays).view(cls)
        obj.names = list(obj.dtype.names)
        obj.stdzr = stdzr
        return obj

    def _rewrap(self, name, values):
        return ParameterArray(**{name: values}, stdzr=self.stdzr)

    def _from_layers(self, layers: dict):
        return ParameterArray(**layers, stdzr=self.stdzr)

    def parray(self, *args, **kwargs):
        return ParameterArray(*args, **kwargs, stdzr=self.stdzr)

    def get(self, name, default=None):
        if isinstance(name, (list, tuple, set)):
            return self._from_layers({n: a for n, a in self.as_dict().items() if n in name})
        return super().get(name, default)

    def __getitem__(self, item):
        if isinstance(item, (list, set)) and all(isinstance(i, str) for i in item):
            return self.get(item)
        return super().__getitem__(item)

    @property
    def z(self) -> LayeredArray:
        plain = np.asarray(self)
        return LayeredArray(
            stdzr=self.stdzr,
            **{n + "_z": self.stdzr.stdz(n, plain[n].astype(float)) for n in self.names},
        )

    @property
    def t(self) -> LayeredArray:
        plain = np.asarray(self)
        return LayeredArray(
            stdzr=self.stdzr,
            **{n + "_t": self.stdzr.transform(n, plain[n].astype(float)) for n in self.names},
        )

    def add_layers(self, stdzd=False, **arrays):
        if isinstance(stdzd, (np.ndarray, list)):  # a layer that happens to be called "stdzd"
            arrays["stdzd"], stdzd = stdzd, False
        if stdzd:
            arrays = {n: self.stdzr.unstdz(n, np.asarray(a)) for n, a in arrays.items()}
        return self._from_layers({**self.as_dict(), **{n: np.asarray(a) for n, a in arrays.items()}})

    def fill_with(self, **params):
        """One constant new layer per keyword, broadcast to this array's shape."""
        assert all(isinstance(v, (float, int, np.floating, np.integer)) for v in params.values())
        return self.add_layers(**{n: np.full(self.shape, v) for n, v in params.items()})

    @classmethod
    def _combine(cls, fn, parrays, **kwargs):
        names = [pa.names for pa in parrays]
        if any(n != names[0] for n in names):
            raise ValueError("Arrays do not have the same names!")
        joined = fn([np.asarray(pa) for pa in parrays], **kwargs)
        return cls(**{n: joined[n] for n in joined.dtype.names}, stdzr=parrays[0].stdzr)

    @classmethod
    def stack(cls, parray_list, axis=0, **kwargs):
        return cls._combine(np.stack, parray_list, axis=axis, **kwargs)

    @classmethod
    def vstack(cls, parray_list, **kwargs):
        return cls._combine(np.vstack, parray_list, **kwargs)

    @classmethod
    def hstack(cls, parray_list, **kwargs):
        return cls._combine(np.hstack, parray_list, **kwargs)


# ------------------------------------------------------------------------------------------------
class UncertainArray(np.ndarray):
    """Mean ``μ`` and variance ``σ2`` of an independent normal at every index."""

    def __new__(cls, name: str, μ, σ2, stdzr=None, **extra):
        mu, var = np.asarray(μ), np.asarray(σ2)
        assert mu.shape == var.shape
        obj = _structured({"μ": mu, "σ2": var, **extra}, shape=mu.shape).view(cls)
        obj.name = name
        obj.stdzr = stdzr
        obj.fields = list(obj.dtype.names)
        return obj

    def __array_finalize__(self, src):
        if src is None:
            return
        self.name = getattr(src, "name", None)
        self.stdzr = getattr(src, "stdzr", None)
        self.fields = getattr(src, "fields", None)

    def _field(self, key):
        return np.asarray(self)[key]

    @property
    def μ(self):
        return self._field("μ")

    @μ.setter
    def μ(self, val):
        np.ndarray.__setitem__(self, "μ", val)

    @property
    def σ2(self):
        return self._field("σ2")

    @σ2.setter
    def σ2(self, val):
        np.ndarray.__setitem__(self, "σ2", val)

    @property
    def σ(self):
        return np.sqrt(self.σ2)

    @σ.setter
    def σ(self, val):
        np.ndarray.__setitem__(self, "σ2", np.asarray(val) ** 2)

    @property
    def dist(self):
        return norm(loc=self.μ, scale=self.σ)

    def nlpd(self, target):
        """Negative log posterior density of ``target``."""
        return -np.log(self.dist.pdf(target))

    def vEI(self, target, best_yet, k=1):
        """Expected improvement towards a TARGET value (Uhrenholt & Jensen 2019, "Efficient Bayesian
        optimization for target vector estimation"): ``E[max(0, best_yet - |target - y|^2)]`` for
        ``y ~ N(mu, sigma^2 I_k)``.  ``|target - y|^2 / sigma^2`` is noncentral chi-square with k degrees
        of freedom and noncentrality ``|target - mu|^2 / sigma^2``, which gives the closed form below;
        ``best_yet`` is the smallest SQUARED distance observed so far (reference ``arrays.py:672-697``)."""
        from scipy.stats import ncx2

        nc = (target - self.μ) ** 2 / self.σ2
        x = best_yet / self.σ2
        return best_yet * ncx2.cdf(x, k, nc) - self.σ2 * (k * ncx2.cdf(x, k + 2, nc) + nc * ncx2.cdf(x, k + 4, nc))

    @staticmethod
    def stack(uarray_list, axis=0):
        names = {ua.name for ua in uarray_list}
        if len(names) != 1:
            raise ValueError("Arrays do not have the same name!")
        joined = np.stack([np.asarray(ua) for ua in uarray_list], axis=axis)
        return UncertainArray(uarray_list[0].name, **{f: joined[f] for f in joined.dtype.names})

    def __repr__(self):
        return f"{self.name}{self.fields}: {np.asarray(self)}"

    __str__ = __repr__

    def _make(self, name, mu, var):
        return UncertainArray(name, mu, var)

    def __getitem__(self, item):
        picked = np.ndarray.__getitem__(self, item)
        if _is_point_index(item):
            plain = np.asarray(picked)
            return self._make(self.name, plain["μ"], plain["σ2"])
        if isinstance(item, slice):
            return picked
        return np.asarray(picked)

    # --- first-order propagation for independent normals --------------------------------------
    def _moments(self):
        return self.μ.astype(float), self.σ2.astype(float)

    def sum(self, axis=None, dtype=None, out=None, keepdims=False, **kwargs):
        mu, var = self._moments()
        return self._make(self.name, mu.sum(axis=axis, keepdims=keepdims), var.sum(axis=axis, keepdims=keepdims))

    def mean(self, axis=None, dtype=None, out=None, keepdims=False, **kwargs):
        mu, var = self._moments()
        n = mu.size if axis is None else np.prod([mu.shape[a] for a in np.atleast_1d(axis)])
        return self._make(
            self.name, mu.mean(axis=axis, keepdims=keepdims), var.sum(axis=axis, keepdims=keepdims) / n**2
        )

    def _binary(self, other, symbol, fn):
        mu, var = self._moments()
        if isinstance(other, UncertainArray):
            omu, ovar = other._moments()
            name = self.name if self.name == other.name else f"({self.name}{symbol}{other.name})"
        else:
            omu, ovar = np.asarray(other, dtype=float), 0.0
            name = self.name
        return self._make(name, *fn(mu, var, omu, ovar))

    def __add__(self, other):
        return self._binary(other, "+", lambda m, v, om, ov: (m + om, v + ov))

    __radd__ = __add__

    def __sub__(self, other):
        return self._binary(other, "-", lambda m, v, om, ov: (m - om, v + ov))

    def __rsub__(self, other):
        if isinstance(other, UncertainArray):
            return other.__sub__(self)
        mu, var = self._moments()
        return self._make(self.name, np.asarray(other, dtype=float) - mu, var)

    def __mul__(self, other):
        return self._binary(other, "*", lambda m, v, om, ov: (m * om, om**2 * v + m**2 * ov))

    __rmul__ = __mul__

    def __truediv__(self, other):
        return self._binary(
            other, "/", lambda m, v, om, ov: (m / om, v / om**2 + (m**2 / om**4) * ov)
        )


# ------------------------------------------------------------------------------------------------
class UncertainParameterArray(UncertainArray):
    """Mean / variance of a parameter with a :class:`Standardizer` attached.

    Follows the reference convention (:883-913): ``μ`` is the natural-scale location (the
    transform's inverse applied to the transformed-space mean) while ``σ2`` stays the variance in
    transformed space.  ``stdzd=True`` takes standardized moments and un-standardizes them
    (``Standardizer._unstdz_dist``), which is how ``predict_points`` wraps the engine's output
    (``gumbi/regression/base.py:578-580``)."""

    def __new__(cls, name: str, μ, σ2, stdzr: Standardizer = None, stdzd=False):
        mu, var = np.asarray(μ), np.asarray(σ2)
        assert mu.shape == var.shape
        if stdzd:
            mu, var = stdzr.unstdz(name, mu, var)
        obj = _structured({"μ": mu, "σ2": var}, shape=np.shape(mu)).view(cls)
        obj.name = name
        obj.stdzr = stdzr
        obj.fields = list(obj.dtype.names)
        return obj

    def _make(self, name, mu, var):
        return UncertainParameterArray(name, mu, var, stdzr=self.stdzr)

    @property
    def z(self) -> UncertainArray:
        zm, zv = self.stdzr.stdz(self.name, self.μ, self.σ2)
        return UncertainArray(f"{self.name}_z", zm, zv, stdzr=self.stdzr)

    @property
    def t(self) -> UncertainArray:
        tm, tv = self.stdzr.transform(self.name, self.μ, self.σ2)
        return UncertainArray(f"{self.name}_t", tm, tv, stdzr=self.stdzr)

    @property
    def _kind(self):
        return self.stdzr.kind(self.name) if self.stdzr is not None else "identity"

    @property
    def dist(self):
        kind = self._kind
        if kind == "log":
            return lognorm(scale=self.μ, s=self.σ)
        if kind == "logit":
            return _LogitNormal(self.μ, self.σ)
        return norm(loc=self.μ, scale=self.σ)

    def _from_z(self, z: UncertainArray):
        return UncertainParameterArray(z.name.replace("_z", ""), z.μ, z.σ2, stdzr=self.stdzr, stdzd=True)

    def _from_t(self, t: UncertainArray):
        name = t.name.replace("_t", "")
        mu, var = self.stdzr.untransform(name, t.μ, t.σ2)
        return UncertainParameterArray(name, mu, var, stdzr=self.stdzr)

    def _warn_if_poorly_defined(self):
        if self._kind != "identity":
            warnings.warn(f"Transform is poorly defined for {self._kind}; results may be unexpected.")

    def sum(self, axis=None, dtype=None, out=None, keepdims=False, **kwargs):
        self._warn_if_poorly_defined()
        return self._from_z(self.z.sum(axis=axis, keepdims=keepdims))

    def mean(self, axis=None, dtype=None, out=None, keepdims=False, **kwargs):
        """Natural-space parameters of the mean of the standardized-space normals."""
        return self._from_z(self.z.mean(axis=axis, keepdims=keepdims))

    def extract(self, field):
        assert_in("field", field, self.fields)
        return ParameterArray(**{self.name: getattr(self, field)}, stdzr=self.stdzr)

    def _combine_t(self, other, op):
        self._warn_if_poorly_defined()
        if isinstance(other, UncertainParameterArray):
            if self.stdzr != other.stdzr:
                warnings.warn("uparrays have dissimilar Standardizers")
            new = self._from_t(op(self.t, other.t))
            new.stdzr = Standardizer(**{**self.stdzr, **other.stdzr})
            return new
        return self._from_z(op(self.z, other))

    def __add__(self, other):
        return self._combine_t(other, lambda a, b: a + b)

    def __sub__(self, other):
        return self._combine_t(other, lambda a, b: a - b)

    def __rsub__(self, other):
        return self._combine_t(other, lambda a, b: b - a)


class _LogitNormal:
    """Logit-normal helper: ``expit`` of a normal with mean ``logit(loc)`` and sd ``scale``."""

    def __init__(self, loc, scale):
        self._n = norm(loc=logit(loc), scale=scale)

    def pdf(self, x):
        return self._n.pdf(logit(x)) / (x * (1 - x))

    def cdf(self, x):
        return self._n.cdf(logit(x))

    def ppf(self, q):
        return expit(self._n.ppf(q))

    def rvs(self, size=None, random_state=None):
        return expit(self._n.rvs(size=size, random_state=random_state))


# ------------------------------------------------------------------------------------------------
class MVUncertainParameterArray(np.ndarray):
    """Several :class:`UncertainParameterArray` of one shape plus their correlation matrix --
    what ``predict_points`` returns for a multi-output GP (``gumbi/regression/base.py:582-599``)."""

    def __new__(cls, *uparrays, cor, stdzr=None):
        shape = uparrays[0].shape
        assert all(u.shape == shape for u in uparrays)
        assert np.shape(cor)[0] == len(uparrays)
        stdzr = uparrays[0].stdzr if stdzr is None else stdzr
        mu = _structured({u.name: u.μ for u in uparrays}, shape=shape)
        var = _structured({u.name: u.σ2 for u in uparrays}, shape=shape)
        base = np.empty(shape, dtype=np.dtype([("μ", mu.dtype), ("σ2", var.dtype)]))
        base["μ"], base["σ2"] = mu, var
        obj = base.view(cls)
        obj.names = [u.name for u in uparrays]
        obj.stdzr = stdzr
        obj.fields = ["μ", "σ2"]
        obj.cor = np.asarray(cor)
        return obj

    def __array_finalize__(self, src):
        if src is None:
            return
        for attr in ("names", "fields", "stdzr", "cor"):
            setattr(self, attr, getattr(src, attr, None))

    def __repr__(self):
        return f"{tuple(self.names)}{self.fields}: {np.asarray(self)}"

    def _layer(self, field) -> ParameterArray:
        plain = np.asarray(self)[field]
        return ParameterArray(**{n: plain[n] for n in self.names}, stdzr=self.stdzr)

    @property
    def μ(self) -> ParameterArray:
        return self._layer("μ")

    @property
    def σ2(self) -> ParameterArray:
        return self._layer("σ2")

    @property
    def σ(self) -> ParameterArray:
        plain = np.asarray(self)["σ2"]
        return ParameterArray(**{n: np.sqrt(plain[n]) for n in self.names}, stdzr=self.stdzr)

    def get(self, name, default=None):
        if isinstance(name, str):
            if name not in self.names:
                return default
            plain = np.asarray(self)
            return UncertainParameterArray(name, plain["μ"][name], plain["σ2"][name], stdzr=self.stdzr)
        idx = [self.names.index(n) for n in name]
        return MVUncertainParameterArray(*[self.get(n) for n in name], cor=self.cor[np.ix_(idx, idx)],
                                         stdzr=self.stdzr)

    def __getitem__(self, item):
        if isinstance(item, str) and item in ("μ", "σ2"):
            return self._layer(item)
        if _is_point_index(item):
            return MVUncertainParameterArray(*[self.get(n)[item] for n in self.names], cor=self.cor,
                                             stdzr=self.stdzr)
        return np.ndarray.__getitem__(self, item)

    def parray(self, *args, **kwargs):
        kwargs.setdefault("stdzr", self.stdzr)
        return ParameterArray(*args, **kwargs)

    def uparray(self, *args, **kwargs):
        kwargs.setdefault("stdzr", self.stdzr)
        return UncertainParameterArray(*args, **kwargs)

    def mvuparray(self, *args, **kwargs):
        kwargs.setdefault("stdzr", self.stdzr)
        kwargs.setdefault("cor", self.cor)
        return MVUncertainParameterArray(*args, **kwargs)

    @property
    def t(self):
        stdzr = Standardizer(**{k + "_t": v for k, v in self.stdzr.items()})
        return MVUncertainParameterArray(*[self.get(n).t for n in self.names], cor=self.cor, stdzr=stdzr)

    @property
    def z(self):
        stdzr = Standardizer(**{n + "_z": {"μ": 0, "σ2": 1} for n in self.names})
        return MVUncertainParameterArray(*[self.get(n).z for n in self.names], cor=self.cor, stdzr=stdzr)

    def cov(self, stdzd=True, whiten=1e-10):
        """Covariance matrix of a single (0-d) point."""
        if self.ndim != 0:
            raise NotImplementedError("Multidimensional multivariate covariance calculations are not yet supported.")
        sd = np.array([float((self.get(n).z if stdzd else self.get(n).t).σ) for n in self.names])
        cov = np.diag(sd) @ self.cor @ np.diag(sd)
        if whiten:
            cov = cov + whiten * np.eye(len(sd))
        return cov

    @property
    def dist(self):
        """Frozen multivariate normal of a single point in standardized space."""
        if self.ndim != 0:
            raise NotImplementedError("Multidimensional multivariate distributions are not yet supported.")
        mean = np.array([float(self.get(n).z.μ) for n in self.names])
        return multivariate_normal(mean=mean, cov=self.cov(stdzd=True))

    def mahalanobis(self, parray: ParameterArray) -> float:
        cov_inv = np.linalg.inv(self.cov(stdzd=True))
        pz = parray.z
        diff = np.array([float(pz[n + "_z"].values()) - float(self.get(n).z.μ) for n in self.names])
        return float(np.sqrt(diff @ cov_inv @ diff))

    def outlier_pval(self, parray: ParameterArray) -> float:
        return 1 - chi2.cdf(self.mahalanobis(parray) ** 2, df=len(self.names))
