"""Backend-agnostic GP front end: dimension parsing, data shaping, grids, prediction wrapping.

API-compatible counterpart of ``gumbi/regression/base.py`` (``Regressor`` :21): same method and
attribute names, same argument meaning, same error behaviour, so the reference's own parsing
tests (``tests/test_regression.py:49-112``) read unchanged against it.  Two algorithms differ on
purpose (SURVEY.md section 0.5 -- the reference versions are O(N^2) and take minutes at N >= 5e4):

* ``get_structured_data`` maps levels to coordinates per model dimension with a vectorised
  ``Series.map`` and skips dimensions whose coordinates are the identity, instead of
  ``DataFrame.replace`` over the whole frame with an N-entry dict (reference :418-422);
* the structured data is cached between ``build_model`` and ``prepare_grid`` (reference :646
  recomputes it).
Outputs are identical on the reference's fixtures (``tests/golden/plumbing_*.npz``).
"""

from __future__ import annotations

import warnings
from abc import ABC, abstractmethod
from itertools import product

import numpy as np
import pandas as pd

from ..aggregation import DataSet
from ..arrays import MVUncertainParameterArray as mvuparray
from ..arrays import ParameterArray
from ..arrays import ParameterArray as parray
from ..arrays import UncertainParameterArray as uparray
from ..utils.misc import assert_in, assert_is_subset

__all__ = ["Regressor"]

_NUMERIC_KINDS = "fiu"


class Regressor(ABC):
    """Surface learning and prediction from a :class:`DataSet` (see module docstring).

    Dimensions: *continuous* dims get the stationary kernel (``linear_dims`` ⊂ continuous get an
    extra linear kernel); *categorical* dims are coregionalised; any dim with a single level
    becomes a *filter* dim; with several outputs the DataSet's ``names_column`` is itself a
    categorical dim, always last (reference :180-265).
    """

    #: model description filled in by ``specify_model`` (lists / dicts, empty until then) ...
    _SPEC_LISTS = ("continuous_dims", "linear_dims", "categorical_dims")
    _SPEC_DICTS = ("continuous_levels", "continuous_coords", "categorical_levels", "categorical_coords", "filter_dims", "model_specs")
    #: ... and what the data / grid / prediction calls leave behind (None until they have run)
    _RESULT_SLOTS = ("X", "y", "grid_vectors", "grid_parray", "grid_points", "ticks", "predictions", "predictions_X", "_structured_cache")

    def __init__(self, dataset: DataSet, outputs=None, seed=2021):
        if not isinstance(dataset, DataSet):
            raise TypeError("Learner instance must be initialized with a DataSet object")
        self.data, self.stdzr, self.out_col, self.seed = dataset, dataset.stdzr, dataset.names_column, seed
        wanted = dataset.outputs if outputs is None else outputs
        self.outputs = list(wanted) if isinstance(wanted, list) else [wanted]
        for name in self._SPEC_LISTS:
            setattr(self, name, [])
        for name in self._SPEC_DICTS:
            setattr(self, name, {})
        for name in self._RESULT_SLOTS:
            setattr(self, name, None)
        self.additive = False

    # ------------------------------------------------------------------------------------------
    @abstractmethod
    def fit(self, *args, **kwargs):
        """Defined by the backend."""

    @abstractmethod
    def build_model(self, *args, **kwargs):
        """Defined by the backend."""

    @abstractmethod
    def predict(self, points_array, with_noise=True, **kwargs):
        """Backend seam (reference :477-495): ``points_array`` is float64 (M, len(dims)) in
        standardized units, columns in ``self.dims`` order; returns ``(mean, var)``, each (M,),
        standardized."""

    # -- conveniences ------------------------------------------------------------------------------
    def parray(self, **kwargs) -> parray:
        return parray(stdzr=self.stdzr, **kwargs)

    def uparray(self, name: str, μ, σ2, **kwargs) -> uparray:
        return uparray(name, μ, σ2, stdzr=self.stdzr, **kwargs)

    def mvuparray(self, *uparrays, cor, **kwargs) -> mvuparray:
        return mvuparray(*uparrays, cor=cor, stdzr=self.stdzr, **kwargs)

    @property
    def dims(self) -> list:
        return self.continuous_dims + self.categorical_dims

    @property
    def levels(self) -> dict:
        return {**self.continuous_levels, **self.categorical_levels}

    @property
    def coords(self) -> dict:
        return {**self.continuous_coords, **self.categorical_coords}

    # -- model specification -------------------------------------------------------------------------
    def specify_model(self, outputs=None, linear_dims=None, continuous_dims=None, continuous_levels=None,
                      continuous_coords=None, categorical_dims=None, categorical_levels=None, additive=False):
        """Validate and normalise dims / levels / coords (reference :180-265): same attributes, same errors."""
        self._structured_cache = None
        chosen = self.outputs if outputs is None else outputs
        assert_is_subset(self.out_col, chosen, self.data.outputs)
        self.outputs = chosen if isinstance(chosen, list) else [chosen]

        cont, lin, cat = (self._parse_dimensions(d) for d in (continuous_dims, linear_dims, categorical_dims))
        if not set(cat).isdisjoint(cont):
            raise ValueError("Overlapping items in categorical_dims and continuous_dims")
        cont_levels = self._parse_levels(cont, continuous_levels)
        cat_levels = self._parse_levels(cat, categorical_levels)
        cat = cat + [self.out_col]  # the output column is a categorical dimension too: always the last one
        cat_levels[self.out_col] = self.outputs

        # A dimension with one level does not vary: it SELECTS rows (get_filtered_data) instead of entering the model -- unless the
        # table has a single observation, where everything has one level.  Decided once, in the order of `dims`.
        by_dim = {**{d: cont_levels[d] for d in cont}, **{d: cat_levels[d] for d in cat}}
        selects = self.data.wide.shape[0] > 1
        self.filter_dims = {d: lv for d, lv in by_dim.items() if selects and len(lv) == 1}
        keep = lambda names: [d for d in names if d not in self.filter_dims]  # noqa: E731
        self.continuous_dims, self.categorical_dims, self.linear_dims = keep(cont), keep(cat), lin
        self.continuous_levels = {d: lv for d, lv in cont_levels.items() if d not in self.filter_dims}   # (the dicts keep the
        self.categorical_levels = {d: lv for d, lv in cat_levels.items() if d not in self.filter_dims}  # order they were given in)

        self.continuous_coords = self._parse_coordinates(self.continuous_dims, self.continuous_levels, continuous_coords)
        self.categorical_coords = self._parse_coordinates(self.categorical_dims, self.categorical_levels, None)
        assert_is_subset("continuous dimensions", self.linear_dims, self.continuous_dims)
        self.additive = additive
        return self

    def _parse_dimensions(self, dims) -> list:
        if dims is None:
            return []
        assert self.out_col not in dims
        dims = dims if isinstance(dims, list) else [dims]
        assert_is_subset("columns", dims, self.data.tidy.columns)
        return list(dims)

    def _parse_levels(self, dims: list, levels) -> dict:
        if not dims:
            return {}
        tidy = self.data.tidy
        all_levels = lambda dim: list(tidy[dim].unique())  # noqa: E731
        if levels is None:
            levels = {dim: all_levels(dim) for dim in dims}
        elif isinstance(levels, (str, list)):
            assert len(dims) == 1, "Non-dict argument for `levels` only allowed if `len(dims)==1`"
            levels = {dims[0]: levels if isinstance(levels, list) else [levels]}
        elif isinstance(levels, dict):
            levels = {d: (v if isinstance(v, list) else [v]) for d, v in levels.items()}  # no caller mutation
            bad = [d for d in levels if d not in dims]
            if bad:
                raise KeyError(f"Dimensions {bad} specified in *levels not found in *dims")
            seen = {k: set(tidy[k].unique().tolist()) for k in levels}
            bad = {k: v for k, vs in levels.items() for v in vs if v not in seen[k]}
            if bad:
                raise ValueError(f"Values specified in *levels not found in tidy: {bad}")
            for dim in dims:
                levels.setdefault(dim, all_levels(dim))
        else:
            raise TypeError("`levels` must be of type str, list, or dict")
        for dim in dims:
            assert_is_subset(f"data[{dim}]", levels[dim], tidy[dim])
        return levels

    def _parse_coordinates(self, dims: list, levels: dict, coords) -> dict:
        if coords is None:
            return {dim: self._make_coordinates(dim, lv) for dim, lv in levels.items()}
        if isinstance(coords, dict):
            want = [(d, lv) for d, lvs in levels.items() for lv in lvs]
            have = [(d, lv) for d, cd in coords.items() for lv in cd.keys()]
            assert_is_subset("coordinates", have, want)
            assert_is_subset("coordinates", want, have)
        elif isinstance(coords, list):
            assert len(levels) == 1, "Non-dict argument for `continuous_coords` only allowed if `len(continuous_dims)==1`"
            dim = dims[0]
            assert len(coords) == len(levels[dim])
            coords = {dim: dict(zip(levels[dim], coords))}
        else:
            raise TypeError("Coordinates must be of type list or dict")
        if not all(isinstance(c, (int, float)) for cd in coords.values() for c in cd.values()):
            raise TypeError("Coordinates must be numeric")
        return coords

    def _make_coordinates(self, dim: str, levels_list: list) -> dict:
        """Numeric columns are their own coordinates; others get the alphabetical category
        index among the considered levels (reference :342-353)."""
        col = self.data.tidy[dim]
        if col.dtype in (np.float32, np.float64, np.int32, np.int64):
            return {lv: lv for lv in levels_list}
        present = col[col.isin(levels_list)]
        cats = present.astype("category").cat.categories.to_list()
        return {lv: cats.index(lv) for lv in levels_list}

    # -- data extraction ---------------------------------------------------------------------------------
    def get_filtered_data(self, standardized=False, metric="mean"):
        """Rows of the tidy frame matching filter dims, levels and the metric (reference :355-387)."""
        df = self.data.tidy
        allowed = pd.Series(True, index=df.index)
        for dim, lv in self.filter_dims.items():
            allowed &= df[dim].isin(lv)
        if "Metric" in df.columns and metric == "mean":
            assert_in("Metric", metric, df["Metric"].unique())
            allowed &= df["Metric"] == metric
        elif "Metric" not in df.columns and metric != "mean":
            raise KeyError(f"No 'Metric' column found in dataset. Cannot filter by {metric}")
        elif metric != "mean":
            raise ValueError(f"Only 'mean' is supported for 'metric'. Got {metric}")
        for dim, lv in self.levels.items():
            allowed &= df[dim].isin(lv)
        source = self.data.tidy.z if standardized else pd.DataFrame(df)
        return source[allowed.to_numpy()]

    def get_structured_data(self, metric="mean"):
        """Inputs and observations as parrays (reference :389-433)."""
        if self._structured_cache is not None and self._structured_cache[0] == metric:
            return self._structured_cache[1], self._structured_cache[2]
        df = self.get_filtered_data(standardized=False, metric=metric)
        names = df[self.out_col].to_numpy()
        masks = {out: names == out for out in self.outputs}
        assert len({int(m.sum()) for m in masks.values()}) == 1
        inputs = df[masks[self.outputs[0]]]

        dim_values = {}
        for dim in self.dims:
            if dim == self.out_col:
                continue
            col = inputs[dim]
            mapping = self.coords[dim]
            if all(k is v or k == v for k, v in mapping.items()) and col.dtype.kind in _NUMERIC_KINDS:
                dim_values[dim] = col.to_numpy()
            else:
                dim_values[dim] = col.map(mapping).to_numpy()
        X = self.parray(**dim_values, stdzd=False)
        values = df[self.data.values_column if self.data.values_column in df.columns else "Value"].to_numpy()
        y = self.parray(**{out: values[masks[out]] for out in self.outputs}, stdzd=False)
        self._structured_cache = (metric, X, y)
        return X, y

    def get_shaped_data(self, metric="mean", dropna=True):
        """Plain arrays for the backend: X (n_obs, n_dims) C-contiguous, y (n_obs,), standardized;
        multi-output rows stacked output-major with the task coordinate last (reference :435-471)."""
        self.X, self.y = self.get_structured_data(metric=metric)
        if self.out_col in self.dims:
            ordered = sorted(self.coords[self.out_col].items(), key=lambda kv: kv[1])
            yz = self.y.z
            y = np.hstack([yz[name + "_z"].values() for name, _ in ordered])
            Xcol = self.X[:, None]
            stacked = parray.vstack([Xcol.add_layers(**{self.out_col: coord}) for _, coord in ordered])
            sz = stacked.z
            X = np.atleast_2d(np.column_stack([sz[dim + "_z"].values().squeeze() for dim in self.dims]))
        else:
            y = self.y.z.values().squeeze()
            xz = self.X.z
            X = np.atleast_2d(np.column_stack([xz[dim + "_z"].values().squeeze() for dim in self.dims]))
        keep = ~np.isnan(y)
        return np.ascontiguousarray(X[keep], dtype=np.float64), np.ascontiguousarray(y[keep], dtype=np.float64)

    # -- prediction ----------------------------------------------------------------------------------------
    def _check_has_prediction(self):
        if self.predictions is None:
            raise ValueError("No predictions found. Run self.predict_grid or related method first.")

    def _parse_prediction_output(self, output):
        if self.out_col in self.categorical_dims:
            allowed = self.categorical_levels[self.out_col]
            if output is None:
                return allowed
            if isinstance(output, str):
                output = [output]
            elif not isinstance(output, list):
                raise ValueError('"output" must be list, string, or None')
            assert_is_subset("Outputs", output, allowed)
            return output
        return self.filter_dims[self.out_col]

    def _prepare_points_for_prediction(self, points: ParameterArray, output):
        points = np.atleast_1d(points)
        assert points.ndim == 1
        assert set(self.dims) - {self.out_col} == set(points.names), \
            'All model dimensions must be present in "points" parray.'
        if self.out_col in self.categorical_dims:
            param_coords = [self.categorical_coords[self.out_col][p] for p in output]
            tall = parray.vstack([points.add_layers(**{self.out_col: c})[:, None] for c in param_coords])
        else:
            param_coords = None
            tall = points[:, None]
        tz = tall.z
        points_array = np.hstack([tz[dim + "_z"].values() for dim in self.dims])
        return points_array, tall, param_coords

    def predict_points(self, points, output=None, with_noise=True, **kwargs):
        """Predictions at a 1-D parray of points: a uparray for one output, an mvuparray with the
        coregion correlation for several (reference :548-601)."""
        output = self._parse_prediction_output(output)
        points_array, tall, param_coords = self._prepare_points_for_prediction(points, output=output)
        mean, var = self.predict(points_array, with_noise=with_noise, **kwargs)
        self.predictions_X = points
        if len(output) == 1:
            self.predictions = self.uparray(output[0], mean, var, stdzd=True)
            return self.predictions
        task = tall[self.out_col].values().squeeze()
        parts = []
        for name, coord in zip(output, param_coords):
            sel = task == coord
            parts.append(self.uparray(name, mean[sel], var[sel], stdzd=True))
        W = np.asarray(self.MAP[f"W_{self.out_col}"])[param_coords, :]
        kappa = np.asarray(self.MAP[f"κ_{self.out_col}"])[param_coords]
        B = W @ W.T + np.diag(kappa)
        sd = np.sqrt(np.diag(B))[None, :]
        self.predictions = self.mvuparray(*parts, cor=B / (sd.T @ sd))
        return self.predictions

    def _default_grid_limits(self, free_dims):
        """Standardized limits of the free continuous dims when the caller gives none: the range of the data, never
        narrower than [-2, 2], widened by a tenth of its width on either side."""
        X, _ = self.get_structured_data("mean")  # (cached since build_model)
        z = np.atleast_2d(X.z.values()).T        # one column per model dim, in ``dims`` order
        low, high = np.minimum(z.min(axis=0), -2.0), np.maximum(z.max(axis=0), 2.0)
        margin = 0.1 * (high - low)
        spans = {dim: (low[i] - margin[i], high[i] + margin[i]) for i, dim in enumerate(self.dims) if dim in free_dims}
        return self.parray(**{dim: np.asarray(span) for dim, span in spans.items()}, stdzd=True)

    def _grid_resolution(self, resolution):
        if isinstance(resolution, dict):
            assert_is_subset("continuous dimensions", resolution.keys(), self.continuous_dims)
            return resolution
        if isinstance(resolution, int):
            return dict.fromkeys(self.continuous_dims, resolution)
        raise TypeError('"resolution" must be a dictionary or an integer')

    def prepare_grid(self, limits=None, at=None, resolution=100):
        """Regular grid over the continuous dims not pinned by ``at``; every other continuous dim is held at its value in
        ``at`` (one point, any number of layers).  Counterpart of the reference's :603-728 -- same arguments, same
        attributes left behind (``grid_vectors``, ``grid_parray``, ``grid_points``, ``prediction_dims``)."""
        self.predictions = self.predictions_X = None
        pinned = {}
        if at is not None:
            if not isinstance(at, ParameterArray):
                raise TypeError('"at" must be a ParameterArray')
            if at.ndim != 0:
                raise ValueError('"at" must be single point, potentially with multiple layers')
            pinned = at.as_dict()
        free = set(self.continuous_dims) - set(pinned)
        if not free:
            raise ValueError("At least one dimension must be non-degenerate to generate grid.")

        defaults = self._default_grid_limits(free)
        if limits is None:
            limits = defaults
        elif not isinstance(limits, ParameterArray):
            raise TypeError('"limits" must be a ParameterArray')
        else:
            missing = free - set(limits.names)
            if missing:  # dims the caller left out keep their default range
                limits = limits.add_layers(**defaults[list(missing)].as_dict())
        gridded = set(limits.names)
        if gridded & set(pinned):
            raise ValueError('Dimensions specified via "limits" and in "at" must not overlap.')
        if not set(self.continuous_dims) <= (gridded | set(pinned)):
            raise ValueError('Not all continuous dimensions are specified by "limits" or "at".')

        per_dim = self._grid_resolution(resolution)
        lz = limits.z
        axes = {}
        for dim in gridded:
            start, stop = lz[dim + "_z"].values()
            axes[dim] = self.parray(**{dim: np.linspace(start, stop, per_dim[dim])[:, None]}, stdzd=True)
        order = [dim for dim in self.dims if dim in gridded]  # grid axes follow the model's dimension order
        mesh = np.meshgrid(*(axes[dim] for dim in order), indexing="ij")
        grid = self.parray(**{m.names[0]: m.values() for m in mesh})
        if pinned:
            grid = grid.add_layers(**{dim: np.full(grid.shape, value) for dim, value in pinned.items()})
        self.prediction_dims, self.grid_vectors = order, axes
        self.grid_parray, self.grid_points = grid, grid.ravel()
        return grid

    def marginal_grids(self, *dims):
        if self.grid_points is None:
            raise ValueError("Grid must first be specified with `prepare_grid`")
        assert_is_subset("GP dims", dims, self.prediction_dims)
        ordered = [d for d in self.dims if d in dims]
        grids = np.meshgrid(*[self.grid_vectors[d] for d in ordered], indexing="ij")
        return [grids[ordered.index(d)] for d in dims]

    def predict_grid(self, output=None, categorical_levels=None, with_noise=True, **kwargs):
        """Predictions on the grid ``prepare_grid`` left behind, in the grid's shape (reference :751-783); for models with
        categorical dims ``categorical_levels`` names the level of each."""
        if self.grid_points is None:
            raise ValueError("Grid must first be specified with `prepare_grid`")
        shape = self.grid_parray.shape
        where = self.append_categorical_points(self.grid_points, categorical_levels) if self.categorical_dims else self.grid_points
        flat = self.predict_points(where, output=output, with_noise=with_noise, **kwargs)
        self.predictions, self.predictions_X = flat.reshape(shape), self.predictions_X.reshape(shape)
        return self.predictions

    def append_categorical_points(self, continuous_parray, categorical_levels):
        """``continuous_parray`` with one constant layer per categorical dim: the coordinate of the level named for it."""
        if categorical_levels is None:
            return continuous_parray
        needed = set(self.categorical_dims) - {self.out_col}
        if set(categorical_levels) != needed:
            raise AttributeError("Must specify level for every categorical dimension")
        coords = {dim: self.categorical_coords[dim][level] for dim, level in categorical_levels.items()}
        return continuous_parray.fill_with(**coords)

    def get_conditional_prediction(self, **dim_values):
        """Slice of the gridded prediction at fixed values of some dims, by linear interpolation
        of mean and variance (reference :1111-1178)."""
        from scipy.interpolate import interpn

        self._check_has_prediction()
        margins = {d: v.squeeze() for d, v in self.grid_vectors.items() if d in self.prediction_dims}
        keep = [d for d in self.prediction_dims if d not in dim_values]
        kept = np.meshgrid(*[margins[d] for d in keep], indexing="ij")
        cond_grid = self.parray(**{g.names[0]: g.values() for g in kept})
        xi = cond_grid.add_layers(
            **{d: np.full(cond_grid.shape, v) for d, v in dim_values.items()}
        ).ravel()
        xz = xi.z
        pts = np.column_stack([xz[d + "_z"].values() for d in self.dims if d in xi.names])
        axes = [margins[d].z.values() for d in self.dims if d in self.prediction_dims]
        mu_i = interpn(axes, self.predictions.μ, pts)
        var_i = interpn(axes, self.predictions.σ2, pts)
        cond = self.uparray(self.predictions.name, μ=mu_i, σ2=var_i).reshape(*cond_grid.shape)
        return cond_grid.squeeze(), cond.squeeze()

    # -- evaluation ----------------------------------------------------------------------------------------------
    def propose(self, target, acquisition="EI"):
        """Next point to measure when steering ONE output towards ``target`` (reference :816-838): the
        maximiser, over the points of the last prediction, of the target-value expected improvement
        (``"EI"``, :meth:`UncertainArray.vEI` on the standardized scale) or of the posterior density at
        the target (``"PD"``).  The reference's version cannot run (it takes ``.z`` of a plain float,
        ``base.py:830-831``) and hands vEI a distance where the acquisition expects a squared distance;
        this one follows the published definition."""
        if self.predictions is None:
            raise ValueError("No predictions to make proposal from!")
        assert_in("acquisition", acquisition, ["EI", "PD"])
        output = self.predictions.name
        df = self.get_filtered_data(standardized=False)
        df = df[df[self.out_col] == output]
        vcol = self.data.values_column if self.data.values_column in df.columns else "Value"
        observed = self.parray(**{output: df[vcol].to_numpy()}, stdzd=False)
        goal = self.parray(**{output: target}, stdzd=False)
        tz = float(np.asarray(goal.z.values()).ravel()[0])
        best_yet = float(np.min((np.asarray(observed.z.values()).ravel() - tz) ** 2))
        pz = self.predictions.z
        if acquisition == "EI":
            self.proposal_surface = np.asarray(pz.vEI(tz, best_yet))
        else:
            self.proposal_surface = -np.asarray(pz.nlpd(tz))
        self.proposal_idx = int(np.argmax(self.proposal_surface))
        self.proposal = self.predictions_X.ravel()[self.proposal_idx]
        return self.proposal

    def cross_validate(self, unit=None, *, n_train=None, pct_train=None, train_only=None, warm_start=True,
                       seed=None, errors="natural", **MAP_kws):
        """Fit on a random subset, score on the rest (reference :844-1105).  Returns
        ``{"train"|"test": {"data", "NLPDs", "errors"}}``."""
        if not (n_train is None) ^ (pct_train is None):
            raise ValueError('Exactly one of "n_train" and "pct_train" must be specified')
        if unit is not None and not isinstance(unit, str):
            raise TypeError('Keyword "unit" must be a single string.')
        assert_in('Keyword "errors"', errors, ["natural", "standardized", "transformed"])
        seed = self.seed if seed is None else seed
        rg = np.random.default_rng(seed)
        df = pd.DataFrame(self.data.wide)

        n_entities = len(set(df.index)) if unit is None else len(set(df.set_index(unit).index))
        n_train = n_train if n_train is not None else int(np.floor(n_entities * pct_train))
        if n_train <= 0:
            raise ValueError("Size of training set must be strictly greater than zero.")
        if n_train > n_entities:
            raise ValueError("Size of training set must be not exceed number of observations or entities in dataset.")

        train_parts = []
        if train_only is not None:
            hit = pd.concat([df[d] == lv for d, lv in train_only.items()], axis=1).all(axis=1)
            idx = hit[hit].index
            part = df.loc[idx] if unit is None else df.loc[idx].set_index(unit)
            n_train -= len(set(part.index))
            if n_train < 0:
                raise ValueError("Adding `train_only` observations exceeded specified size of training set")
            train_parts.append(part)
            df = df.drop(index=idx)
        if unit is not None:
            df = df.set_index(unit)
        if n_train > len(df.index.unique()):
            raise ValueError("Specified size of training set exceeds number of unique combinations found in `dims`")

        cat_dims = [d for d in self.categorical_dims if d != self.out_col]
        if warm_start and cat_dims:
            combos = set(product(*[self.categorical_levels[d] for d in cat_dims]))
            keys = list(df[cat_dims].itertuples(index=False, name=None))
            grouped = df[[k in combos for k in keys]].groupby(cat_dims)
            if grouped.ngroups == 0:
                raise ValueError("None of the combinations of categorical levels were found in data.")
            warm_idx = grouped.sample(1, random_state=seed).index
            if len(set(warm_idx)) != len(warm_idx):
                warnings.warn("Duplicate entities specified by `unit` were selected during `warm_start`.")
            n_train -= len(set(warm_idx))
            if n_train < 0:
                raise ValueError("Adding `warm_start` observations exceeded specified size of training set")
            train_parts.append(df.loc[warm_idx])
            df = df.drop(index=warm_idx)

        train_idx = rg.choice(df.index.unique(), n_train, replace=False)
        train_parts.append(df.loc[train_idx])
        train_df = pd.concat(train_parts).reset_index(drop=unit is None)
        test_df = df.drop(train_idx).reset_index(drop=unit is None)

        base_specs = dict(outputs=self.outputs, linear_dims=self.linear_dims, continuous_dims=self.continuous_dims,
                          categorical_dims=cat_dims, additive=self.additive)

        def _restricted(frame):
            present = lambda dim, lv: lv in frame[dim].values  # noqa: E731
            return dict(
                base_specs,
                continuous_levels={d: [lv for lv in lvs if present(d, lv)] for d, lvs in self.continuous_levels.items()},
                categorical_levels={d: [lv for lv in lvs if present(d, lv)]
                                    for d, lvs in self.categorical_levels.items() if d != self.out_col},
                continuous_coords={d: {lv: c for lv, c in cd.items() if present(d, lv)}
                                   for d, cd in self.continuous_coords.items()},
            )

        ds_specs = dict(outputs=self.data.outputs, names_column=self.data.names_column,
                        values_column=self.data.values_column, log_vars=self.data.log_vars,
                        logit_vars=self.data.logit_vars, stdzr=self.data.stdzr)
        train_ds = DataSet(train_df, **ds_specs)
        test_ds = DataSet(test_df, **ds_specs)

        dev_kw = {"device": self.device} if hasattr(self, "device") else {}  # stay on this instance's GPU
        train_obj = self.__class__(train_ds, outputs=self.outputs, seed=seed, **dev_kw)
        train_obj.specify_model(**_restricted(train_df))
        train_obj.filter_dims = self.filter_dims
        train_obj.build_model(**self.model_specs)
        train_obj.find_MAP(**MAP_kws)

        def _score(obj_for_data, predictor):
            Xs, ys = obj_for_data.get_structured_data()
            pred = predictor.predict_points(Xs)
            err = {"natural": ys.values() - pred.μ, "transformed": ys.t.values() - pred.t.μ,
                   "standardized": ys.z.values() - pred.z.μ}[errors]
            return pred.nlpd(ys.values()), err

        train_nlpd, train_err = _score(train_obj, train_obj)
        if len(test_df.index.unique()) > 0:
            test_obj = self.__class__(test_ds, outputs=self.outputs, seed=seed, **dev_kw)
            test_obj.specify_model(**_restricted(test_df))
            test_obj.filter_dims = self.filter_dims
            test_nlpd, test_err = _score(test_obj, train_obj)
        else:
            test_nlpd, test_err = np.nan, np.nan
        engine = getattr(train_obj, "engine", None)
        if engine is not None:
            engine.close()  # the split's factor buffers go back to the device
        return {"train": {"data": train_ds, "NLPDs": train_nlpd, "errors": train_err},
                "test": {"data": test_ds, "NLPDs": test_nlpd, "errors": test_err}}

    def cross_validate_replicas(self, seeds, group=None, **kwargs):
        """Independent :meth:`cross_validate` splits, one per seed, as replica-parallel jobs: with an
        initialised ``torch.distributed`` process group (one process per GPU) the seeds are dealt
        round-robin over the ranks -- no data-path communication, each fit runs on the rank's own GPU --
        and every rank receives the complete list of results in seed order.  Without a process group
        the splits simply run one after the other.  ``group``: a ``torch.distributed`` group instead of the
        default one.  (the folds of a cross-validation are independent GPs.)"""
        seeds = list(seeds)
        rank, world, dist = 0, 1, None
        try:
            import torch.distributed as dist_mod

            if dist_mod.is_available() and dist_mod.is_initialized():
                dist, rank, world = dist_mod, dist_mod.get_rank(group), dist_mod.get_world_size(group)
        except ImportError:
            pass
        mine = {i: self.cross_validate(seed=seeds[i], **kwargs) for i in range(rank, len(seeds), world)}
        if dist is None:
            return [mine[i] for i in range(len(seeds))]
        gathered = [None] * world
        dist.all_gather_object(gathered, mine, group=group)
        merged = {}
        for part in gathered:
            merged.update(part)
        return [merged[i] for i in range(len(seeds))]
