"""Partial dependence and individual conditional expectation *data* (SURVEY.md 8f f1/f4 neighbours).

The reference computes these inside its plotting functions (``plot_pdp`` ``utils.py:312-487``,
``plot_ice`` ``utils.py:168-310``) and draws them with matplotlib; drawing is out of scope here,
the numbers are not: they are sweeps of posterior predictions -- per covariate ``samples x m x grid``
tree traversals with every OTHER covariate marginalised out by the trees' own training counts
(``excluded``) for the PDP, and ``instances x samples x m x n`` traversals for ICE -- i.e. work for
the ``k_predict`` kernel behind ``PosteriorSampler.sample_posterior``.

What the functions return is exactly what upstream hands to its axes: per covariate the grid
``x`` and the array of predictions; random draws follow the same call pattern (one
``rng.integers`` per prediction call, in the same order).
"""

from __future__ import annotations

import numpy as np

from .utils import _get_posterior_sampler, _resident_rows, _sample_posterior

DEFAULT_QUANTILES = (0.05, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.95)  # upstream's default grid


def _as_matrix(X):
    if hasattr(X, "columns") and hasattr(X, "to_numpy"):
        return np.asarray(X.to_numpy(), np.float64), [str(c) for c in X.columns]
    X = np.asarray(X, np.float64)
    return X, [f"X_{j}" for j in range(X.shape[1])]


def pdp_grid(X, xs_interval: str = "quantiles", xs_values=None) -> np.ndarray:
    """The rows at which the partial dependence is evaluated (one grid per column, side by side):
    ``"insample"`` -- the data themselves; ``"linear"`` -- ``xs_values`` (default 10) equally spaced
    points between each column's minimum and maximum; ``"quantiles"`` -- each column's quantiles
    ``xs_values`` (default 5 % ... 95 %)."""
    X = np.asarray(X, np.float64)
    if xs_interval == "insample":
        return X
    if xs_interval == "linear":
        k = 10 if xs_values is None else int(xs_values)
        return np.linspace(X.min(axis=0), X.max(axis=0), num=k, axis=0)
    if xs_interval == "quantiles":
        q = list(DEFAULT_QUANTILES if xs_values is None else xs_values)
        return np.quantile(X, q=q, axis=0)
    raise ValueError(f"{xs_interval} is not supported: use 'insample', 'linear' or 'quantiles'")


def _samplers(bart, backend):
    group = bart if isinstance(bart, list) else [bart]
    ops = [b.owner.op if getattr(b, "owner", None) is not None else b for b in group]
    got = [_get_posterior_sampler(op, backend=backend) for op in ops]
    return got if isinstance(bart, list) else got[0]


def partial_dependence(bart, X, var_idx=None, xs_interval: str = "quantiles", xs_values=None,
                       samples: int = 200, func=None, random_seed=None, backend=None) -> dict:
    """Partial dependence of the BART function on each covariate of ``var_idx``.

    For covariate ``j`` the forest is evaluated on the grid with all other covariates excluded:
    at a split on an excluded covariate a tree answers with the count-weighted mean of both
    subtrees, which is BART's own marginalisation.  Returns ``{"x": {j: grid_j}, "pd": {j: array
    (samples, grid, outputs)}, "labels": {j: name}, "reference": mean of all partial dependences}``
    (the dashed reference line of the upstream plot)."""
    Xm, names = _as_matrix(X)
    p = Xm.shape[1]
    cols = list(range(p)) if var_idx is None else [int(v) for v in var_idx]
    sampler = _samplers(bart, backend)
    rng = np.random.default_rng(random_seed)
    grid = pdp_grid(Xm, xs_interval, xs_values)
    rows = _resident_rows(sampler, grid)  # one upload for the sweep over the covariates
    out = {"x": {}, "pd": {}, "labels": {}, "reference": None}
    means = []
    for j in cols:
        others = [v for v in range(p) if v != j]
        pd_j = _sample_posterior(sampler, X=rows, rng=rng, size=samples, excluded=others)
        if func is not None:
            pd_j = func(pd_j)
        out["x"][j] = grid[:, j]
        out["pd"][j] = pd_j
        out["labels"][j] = names[j]
        means += [float(pd_j[:, :, k].mean()) for k in range(pd_j.shape[2])]
    out["reference"] = float(np.mean(means)) if means else None
    return out


def individual_conditional_expectation(bart, X, var_idx=None, instances: int = 30, samples: int = 100,
                                       centered: bool = True, func=None, random_seed=None,
                                       backend=None) -> dict:
    """ICE curves: for each of ``instances`` randomly chosen rows, the posterior-mean prediction
    along the observed values of covariate ``j`` with all other covariates held at that row's
    values.  Returns ``{"x": {j: X[:, j]}, "ice": {j: array (instances, n, outputs)}, "labels"}``;
    ``centered`` subtracts each curve's value at the first row, as the upstream plot does."""
    Xm, names = _as_matrix(X)
    n, p = Xm.shape
    cols = list(range(p)) if var_idx is None else [int(v) for v in var_idx]
    sampler = _samplers(bart, backend)
    rng = np.random.default_rng(random_seed)
    chosen = rng.choice(n, replace=False, size=min(int(instances), n))
    out = {"x": {}, "ice": {}, "labels": {}, "instances": chosen}
    for j in cols:
        others = [v for v in range(p) if v != j]
        curves = []
        for row in chosen:
            probe = Xm.copy()
            probe[:, others] = Xm[row, others]
            curves.append(_sample_posterior(sampler, X=probe, rng=rng, size=samples).mean(axis=0))
        ice_j = np.asarray(curves)
        if func is not None:
            ice_j = func(ice_j)
        if centered:
            ice_j = ice_j - ice_j[:, :1, :]
        out["x"][j] = Xm[:, j]
        out["ice"][j] = ice_j
        out["labels"][j] = names[j]
    return out
