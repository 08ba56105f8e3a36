"""Data containers for the GP front end: ``Standardizer``, ``WideData``, ``TidyData``, ``DataSet``.

API-compatible re-implementation of the part of ``gumbi/aggregation.py`` the
``GP.fit() / prepare_grid() / predict_grid()`` path touches (reference:
``Standardizer`` :17-485, ``MetaFrame`` :488-589, ``WideData`` :592-668, ``TidyData`` :671-743,
``DataSet`` :746-956).  The numerical conventions are the reference's:

* a variable is transformed (``log`` for ``log_vars``, ``logit`` for ``logit_vars``, identity
  otherwise), then centred and scaled by the mean / variance *of the transformed values*
  (:388-400, :469-485);
* for a distribution only the mean is mapped through the transform, the variance is carried
  unchanged in transformed space (``mean_transforms`` / ``var_transforms`` :402-448);
* ``from_DataFrame`` uses only ``float64`` columns, sample variance (ddof = 1) per column and
  the pooled population variance for ``isotropic_vars`` (:224-258).
"""

from __future__ import annotations

import numpy as np
import pandas as pd
from scipy.special import expit, logit

from .utils.misc import identity, listify

__all__ = ["Standardizer", "TidyData", "WideData", "DataSet"]

_FORWARD = {"identity": identity, "log": np.log, "logit": logit}
_INVERSE = {"identity": identity, "log": np.exp, "logit": expit}


def _as_name_list(value, label):
    if value is None:
        return []
    if isinstance(value, str):
        return [value]
    if not isinstance(value, list):
        raise TypeError(f"{label} must be a list or str")
    return value


class Standardizer(dict):
    """``{name: {"μ": mean, "σ2": variance}}`` of every *transformed* variable, plus which
    variables are log- / logit-normal.  See the module docstring for the conventions."""

    def __init__(self, log_vars=None, logit_vars=None, isotropic_vars=None, **stats):
        self.validate(stats)
        cleaned = {}
        for name, entry in stats.items():
            entry = dict(entry)
            if "σ2" not in entry:
                entry["σ2"] = entry.pop("σ") ** 2
            cleaned[name] = entry
        super().__init__(**cleaned)
        self._kinds = {}
        for name in _as_name_list(log_vars, "log_vars"):
            self._kinds[name] = "log"
        for name in _as_name_list(logit_vars, "logit_vars"):
            self._kinds[name] = "logit"
        self._log_vars = listify(log_vars)
        self._logit_vars = listify(logit_vars)
        self._isotropic_vars = listify(isotropic_vars)

    # -- construction helpers ---------------------------------------------------------------
    @classmethod
    def validate(cls, dct: dict):
        for name, entry in dct.items():
            if "μ" not in entry or not ("σ" in entry or "σ2" in entry):
                raise AssertionError(f"Standardizer entry {name!r} needs 'μ' and one of 'σ'/'σ2'")

    @classmethod
    def from_DataFrame(cls, df: pd.DataFrame, log_vars=None, logit_vars=None, isotropic_vars=None):
        """Means / variances of the transformed ``float64`` columns of a wide-form frame."""
        iso = listify(isotropic_vars)
        new = cls(log_vars=log_vars, logit_vars=logit_vars)
        stats = {}
        float_cols = [c for c in df.columns if df[c].dtype == np.float64]
        for col in float_cols:
            if col in iso:
                continue
            t = np.asarray(new._transform_value(col, df[col].to_numpy(dtype=float)), dtype=float)
            series = pd.Series(t)
            stats[col] = {"μ": series.mean(), "σ2": series.var()}  # pandas: NaN-skipping, ddof=1
        if iso:
            pooled = np.column_stack(
                [np.asarray(new._transform_value(c, df[c].to_numpy(dtype=float)), dtype=float) for c in iso]
            )
            mu, var = pooled.mean(), pooled.var()
            for col in iso:
                stats[col] = {"μ": mu, "σ2": var}
        return new | stats

    def _clone_kinds_into(self, other: "Standardizer"):
        other._kinds = dict(self._kinds)
        other._log_vars = [v for v, k in other._kinds.items() if k == "log"]
        other._logit_vars = [v for v, k in other._kinds.items() if k == "logit"]

    def __or__(self, other) -> "Standardizer":
        merged = Standardizer(**{**self, **other})
        self._clone_kinds_into(merged)
        if isinstance(other, Standardizer):
            merged._kinds.update(other._kinds)
            merged._log_vars = [v for v, k in merged._kinds.items() if k == "log"]
            merged._logit_vars = [v for v, k in merged._kinds.items() if k == "logit"]
        return merged

    def __ror__(self, other) -> "Standardizer":
        merged = Standardizer(**{**other, **self})
        self._clone_kinds_into(merged)
        return merged

    def __repr__(self):
        return (
            "Standardizer:\n\tlog_vars: {}\n\tlogit_vars: {}\n\n{}".format(
                self.log_vars, self.logit_vars, dict(self)
            )
        )

    # -- transform bookkeeping --------------------------------------------------------------
    @property
    def log_vars(self) -> list:
        return self._log_vars

    @log_vars.setter
    def log_vars(self, names):
        names = _as_name_list(names, "log_vars")
        self._log_vars = names
        for n in names:
            self._kinds[n] = "log"

    @property
    def logit_vars(self) -> list:
        return self._logit_vars

    @logit_vars.setter
    def logit_vars(self, names):
        names = _as_name_list(names, "logit_vars")
        self._logit_vars = names
        for n in names:
            self._kinds[n] = "logit"

    def kind(self, name) -> str:
        """'identity', 'log' or 'logit'."""
        return self._kinds.get(name, "identity")

    @property
    def transforms(self) -> dict:
        """``{name: [forward, inverse]}`` for every known variable (reference attribute)."""
        names = list(self.keys()) + [n for n in self._kinds if n not in self]
        return {n: [_FORWARD[self.kind(n)], _INVERSE[self.kind(n)]] for n in names}

    @transforms.setter
    def transforms(self, dct):
        kinds = {}
        for name, (fwd, _inv) in dct.items():
            if fwd is np.log:
                kinds[name] = "log"
            elif fwd is logit:
                kinds[name] = "logit"
        self._kinds = kinds
        self._log_vars = [v for v, k in kinds.items() if k == "log"]
        self._logit_vars = [v for v, k in kinds.items() if k == "logit"]

    # -- public conversions: value, distribution or Series ------------------------------------
    def _dispatch(self, value_fn, dist_fn, name, μ, σ2, need_mu):
        if isinstance(name, pd.Series):
            return value_fn(name.name, name)
        if μ is None and need_mu:
            raise ValueError("μ cannot be None")
        if σ2 is None:
            return value_fn(name, μ)
        return dist_fn(name, μ, σ2)

    def transform(self, name, μ=None, σ2=None):
        return self._dispatch(self._transform_value, self._transform_dist, name, μ, σ2, True)

    def untransform(self, name, μ=None, σ2=None):
        return self._dispatch(self._untransform_value, self._untransform_dist, name, μ, σ2, False)

    def stdz(self, name, μ=None, σ2=None):
        return self._dispatch(self._stdz_value, self._stdz_dist, name, μ, σ2, False)

    def unstdz(self, name, μ=None, σ2=None):
        return self._dispatch(self._unstdz_value, self._unstdz_dist, name, μ, σ2, False)

    # -- element-wise workers ------------------------------------------------------------------
    def _loc_scale(self, name):
        entry = self.get(name)
        if entry is None:
            return 0, 1
        return entry["μ"], entry["σ2"]

    def _transform_value(self, name, x):
        return _FORWARD[self.kind(name)](x)

    def _untransform_value(self, name, x):
        return _INVERSE[self.kind(name)](x)

    def _stdz_value(self, name, x):
        mu, var = self._loc_scale(name)
        return np.divide(self._transform_value(name, x) - mu, np.sqrt(var))

    def _unstdz_value(self, name, z):
        mu, var = self._loc_scale(name)
        return self._untransform_value(name, np.multiply(z, np.sqrt(var)) + mu)

    def _transform_dist(self, name, mean, var):
        return _FORWARD[self.kind(name)](mean), var

    def _untransform_dist(self, name, mean, var):
        return _INVERSE[self.kind(name)](mean), var

    def _stdz_dist(self, name, mean, var):
        mu, s2 = self._loc_scale(name)
        tmean, tvar = self._transform_dist(name, mean, var)
        return (tmean - mu) / np.sqrt(s2), tvar / s2

    def _unstdz_dist(self, name, z_mean, z_var):
        mu, s2 = self._loc_scale(name)
        return self._untransform_dist(name, z_mean * np.sqrt(s2) + mu, z_var * s2)


# ------------------------------------------------------------------------------------------------
# DataFrame views
# ------------------------------------------------------------------------------------------------
def _melt(wide, outputs, names_column, values_column):
    id_vars = [c for c in wide.columns if c not in outputs]
    return pd.DataFrame(wide).melt(
        id_vars=id_vars, value_vars=outputs, var_name=names_column, value_name=values_column
    )


def _pivot(tidy, names_column, values_column):
    id_vars = [c for c in tidy.columns if c not in (names_column, values_column)]
    wide = pd.DataFrame(tidy).pivot(index=id_vars, columns=names_column, values=values_column)
    return wide.reset_index().rename_axis(columns=None)


class _MetaFrame(pd.DataFrame):
    """DataFrame that remembers which columns are outputs and how to standardize them.
    Slices deliberately fall back to plain ``pd.DataFrame`` (as the reference documents)."""

    _metadata = ["outputs", "names_column", "values_column", "stdzr", "log_vars", "logit_vars", "isotropic_vars"]

    @property
    def _constructor(self):
        return pd.DataFrame

    def _attach(self, outputs, names_column, values_column, stdzr, log_vars, logit_vars, isotropic_vars):
        self.outputs = outputs
        self.names_column = names_column
        self.values_column = values_column
        self.isotropic_vars = isotropic_vars
        if stdzr is None:
            stdzr = Standardizer.from_DataFrame(
                self._stats_frame(), log_vars=log_vars, logit_vars=logit_vars, isotropic_vars=isotropic_vars
            )
            self.log_vars, self.logit_vars = log_vars, logit_vars
        else:
            self.log_vars, self.logit_vars = stdzr.log_vars, stdzr.logit_vars
        self.stdzr = stdzr

    def _stats_frame(self):
        return pd.DataFrame(self)

    @property
    def specs(self) -> dict:
        return dict(
            outputs=self.outputs,
            names_column=self.names_column,
            values_column=self.values_column,
            stdzr=self.stdzr,
            log_vars=self.log_vars,
            logit_vars=self.logit_vars,
        )

    @property
    def inputs(self) -> list:
        return [c for c in self.columns if c not in self.outputs]

    @property
    def float_inputs(self) -> list:
        return [c for c in self.inputs if self[c].dtype == np.float64]

    def __repr__(self):
        head = f"{type(self).__name__}:\n\toutputs: {self.outputs}\n\tinputs: {self.inputs}\n\n"
        return head + pd.DataFrame.__repr__(self)


class WideData(_MetaFrame):
    """Wide-form table (one row per observation, one column per output); ``.z`` / ``.t`` give
    the standardized / transformed copies (reference ``WideData`` :592-668)."""

    def __init__(self, data=None, outputs=None, names_column="Variable", values_column="Value",
                 log_vars=None, logit_vars=None, isotropic_vars=None, stdzr=None, **kwargs):
        super().__init__(data, **kwargs)
        self._attach(outputs, names_column, values_column, stdzr, log_vars, logit_vars, isotropic_vars)

    def _converted(self, fn):
        out = pd.DataFrame(self).copy()
        for col in self.outputs + self.float_inputs:
            out[col] = fn(out[col])
        return out

    @property
    def z(self) -> pd.DataFrame:
        return self._converted(self.stdzr.stdz)

    @property
    def t(self) -> pd.DataFrame:
        return self._converted(self.stdzr.transform)

    def to_tidy(self) -> "TidyData":
        return TidyData(pd.DataFrame(self), **self.specs)

    @classmethod
    def from_tidy(cls, tidy, outputs=None, names_column="Variable", values_column="Value",
                  stdzr=None, log_vars=None, logit_vars=None):
        outputs = outputs if outputs is not None else list(tidy[names_column].unique())
        wide = _pivot(tidy, names_column, values_column)
        return cls(wide, outputs=outputs, names_column=names_column, values_column=values_column,
                   stdzr=stdzr, log_vars=log_vars, logit_vars=logit_vars)


class TidyData(_MetaFrame):
    """Tidy-form table built FROM a wide-form frame (reference ``TidyData`` :671-743): one row
    per (observation, output) with the output name in ``names_column`` and its value in
    ``values_column``."""

    def __init__(self, data=None, outputs=None, names_column="Variable", values_column="Value",
                 log_vars=None, logit_vars=None, isotropic_vars=None, stdzr=None, **kwargs):
        wide = pd.DataFrame(data)
        super().__init__(_melt(wide, outputs, names_column, values_column), **kwargs)
        self._wide_source = wide
        self._attach(outputs, names_column, values_column, stdzr, log_vars, logit_vars, isotropic_vars)

    _metadata = _MetaFrame._metadata + ["_wide_source"]

    def _stats_frame(self):
        return self._wide_source

    @property
    def inputs(self) -> list:
        return [c for c in self.columns if c not in self.outputs]

    def _converted(self, value_fn):
        out = pd.DataFrame(self).copy()
        names = out[self.names_column]
        vals = out[self.values_column].to_numpy(dtype=float).copy()
        for name in pd.unique(names):
            sel = (names == name).to_numpy()
            vals[sel] = value_fn(name, vals[sel])
        out[self.values_column] = vals
        for col in out.columns:
            if col not in (self.names_column, self.values_column) and out[col].dtype == np.float64:
                out[col] = value_fn(col, out[col].to_numpy())
        return out

    @property
    def z(self) -> pd.DataFrame:
        """Standardized values, row-aligned with the tidy frame itself."""
        return self._converted(self.stdzr._stdz_value)

    @property
    def t(self) -> pd.DataFrame:
        return self._converted(self.stdzr._transform_value)

    def to_wide(self) -> WideData:
        return WideData(_pivot(self, self.names_column, self.values_column), **self.specs)


class DataSet:
    """Wide + tidy views of one table sharing one :class:`Standardizer` (reference :746-956)."""

    def __init__(self, data, outputs=None, names_column="Variable", values_column="Value",
                 log_vars=None, logit_vars=None, isotropic_vars=None, stdzr=None):
        self.data = pd.DataFrame(data)
        self.outputs = outputs
        self.names_column = names_column
        self.values_column = values_column
        self.log_vars = log_vars
        self.logit_vars = logit_vars
        self.isotropic_vars = isotropic_vars
        self.stdzr = stdzr
        self._cache = {}
        if self.stdzr is None:
            self.stdzr = Standardizer.from_DataFrame(
                self.data, log_vars=log_vars, logit_vars=logit_vars, isotropic_vars=isotropic_vars
            )
        else:
            self.log_vars = self.stdzr.log_vars
            self.logit_vars = self.stdzr.logit_vars

    def __repr__(self):
        return (
            "DataSet:\n\twide: [{} rows x {} columns]\n\ttidy: [{} rows x {} columns]"
            "\n\toutputs: {}\n\tinputs: {}".format(*self.wide.shape, *self.tidy.shape, self.outputs, self.inputs)
        )

    @property
    def specs(self):
        return dict(
            outputs=self.outputs,
            names_column=self.names_column,
            values_column=self.values_column,
            stdzr=self.stdzr,
            log_vars=self.log_vars,
            logit_vars=self.logit_vars,
        )

    @property
    def inputs(self):
        return [c for c in self.data.columns if c not in self.outputs]

    @property
    def float_inputs(self):
        return [c for c in self.inputs if self.data[c].dtype == np.float64]

    def _view(self, kind):
        # the views are pure functions of (data, stdzr); rebuilding a 1e5-row melt on every
        # attribute access is what made the reference's plumbing slow, so memoise per frame
        key = (kind, id(self.data), id(self.stdzr))
        hit = self._cache.get(kind)
        if hit is None or hit[0] != key:
            cls = WideData if kind == "wide" else TidyData
            hit = (key, cls(self.data, **self.specs))
            self._cache[kind] = hit
        return hit[1]

    @property
    def wide(self) -> WideData:
        return self._view("wide")

    @wide.setter
    def wide(self, wide_df: pd.DataFrame):
        if not any(out in wide_df.columns for out in self.outputs):
            raise AssertionError(f"Dataframe must have at least one of outputs {self.outputs}")
        self.data = pd.DataFrame(wide_df)

    @property
    def tidy(self) -> TidyData:
        return self._view("tidy")

    @tidy.setter
    def tidy(self, tidy_df: pd.DataFrame):
        needed = [self.names_column, self.values_column]
        if not all(col in tidy_df.columns for col in needed):
            raise AssertionError(f"Dataframe must have both columns {needed}")
        self.wide = WideData.from_tidy(tidy_df, **self.specs)

    @classmethod
    def from_tidy(cls, tidy, outputs=None, names_column="Variable", values_column="Value",
                  stdzr=None, log_vars=None, logit_vars=None):
        needed = [names_column, values_column]
        if not all(col in tidy.columns for col in needed):
            raise AssertionError(f"Dataframe must have both columns {needed}")
        wide = WideData.from_tidy(tidy, outputs=outputs, names_column=names_column,
                                  values_column=values_column, stdzr=stdzr, log_vars=log_vars,
                                  logit_vars=logit_vars)
        return cls(pd.DataFrame(wide), **wide.specs)

    @classmethod
    def from_wide(cls, wide, outputs=None, names_column="Variable", values_column="Value",
                  stdzr=None, log_vars=None, logit_vars=None):
        return cls(wide, outputs=outputs, names_column=names_column, values_column=values_column,
                   stdzr=stdzr, log_vars=log_vars, logit_vars=logit_vars)

    def update_stdzr(self):
        """Refresh the means / variances from the current data (keeps the same object)."""
        self.stdzr.update(
            Standardizer.from_DataFrame(self.data, log_vars=self.log_vars, logit_vars=self.logit_vars,
                                        isotropic_vars=self.isotropic_vars)
        )
