"""Variable importance from the posterior of trees.

Counterpart of reference ``pymc_bart/utils.py:868-1090`` (``compute_variable_importance``) and its
helpers ``generate_sequences`` (``:1330-1336``) and ``pearsonr2`` (``:1340-1346``).  The expensive
part -- posterior predictions with a set of covariates marginalised out (``excluded``), O(p) sweeps
for "VI" and O(p^2) for the backward search -- runs in the ``k_predict`` kernel through
``PosteriorSampler.sample_posterior``; the correlations are a few n-vectors of NumPy per sweep.

The ranking logic is held to vectors produced by RUNNING the reference's own functions against deterministic
stand-ins (``tests/golden/variable_importance.json``, generator ``tests/golden/make_utils_golden.py``,
``tests/test_utils_golden.py``): same signature, same draws consumed in the same order, same ``indices``,
``r2_mean``, ``preds`` layout.  Two calls the reference's own code cannot finish work here: ``method="backward"``
on a single-output variable (upstream stops with a ValueError assigning a squeezed array, ``utils.py:1053``) and
``method="backward_VI"`` (upstream never computes ``predicted_all`` on that path, ``utils.py:956-959``: NameError).

* ``"VI"``: variables ordered by how often they were used for splitting; prediction j keeps only the
  j + 1 most used variables (the rest are excluded), the last keeps them all.
* ``"backward"``: greedy elimination -- at every stage the variable whose additional exclusion
  hurts the squared correlation with the full prediction least joins the excluded set.
* ``"backward_VI"``: the ``fixed`` least used variables are eliminated by inclusion counts, the
  others by the backward search.
"""

from __future__ import annotations

import numpy as np

from .utils import _decode_vi, _get_posterior_sampler, _resident_rows, _sample_posterior

CI_PROB = 0.94  # ArviZ's default rcParams["stats.ci_prob"], used by the reference for the r2 interval


def pearsonr2(a, b) -> float:
    """Squared Pearson correlation of two arrays (flattened)."""
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    da, db = a - a.mean(), b - b.mean()
    return float((da @ db) ** 2 / ((da @ da) * (db @ db)))


def generate_sequences(n_vars: int, i_var: int, include: list) -> list:
    """Candidate exclusion sets of stage ``i_var``: ``include`` plus one variable not yet in it
    (the empty set at stage 0)."""
    if not i_var:
        return [()]
    return [tuple(include + [v]) for v in range(n_vars) if v not in include]


def hdi(x, prob: float = CI_PROB) -> np.ndarray:
    """Narrowest interval holding ``prob`` of the sample (what ArviZ's ``hdi`` returns for a
    unimodal sample)."""
    x = np.sort(np.asarray(x, np.float64).ravel())
    n = x.size
    k = max(int(np.floor(prob * n)), 1)
    if k >= n:
        return np.array([x[0], x[-1]])
    widths = x[k:] - x[: n - k]
    i = int(np.argmin(widths))
    return np.array([x[i], x[i + k]])


def inclusion_counts(vi, n_vars: int, model=None, bart_var_name=None) -> np.ndarray:
    """Total split-variable counts from whatever carries them: an InferenceData-like mapping
    (``idata["sample_stats"]["variable_inclusion"]``), a sequence of the base64 stat strings the
    step method emits, or an array of counts ``(draws, p)`` / ``(p,)``.

    A model with several BART variables stores one string per variable and draw (``variable_inclusion_dim_0``);
    as upstream (``utils.py:779-789, 964-988``) the variable is then picked by its position among
    ``model.free_RVs``; ``bart_var_name`` may be a list of names, whose counts are added."""
    if hasattr(vi, "__getitem__") and not isinstance(vi, (list, tuple, np.ndarray)):
        vi = vi["sample_stats"]["variable_inclusion"]
        n_bart = int(getattr(getattr(vi, "variable_inclusion_dim_0", None), "size", 1))
        if n_bart > 1:
            if model is None or bart_var_name is None:
                raise ValueError(
                    "The InferenceData was generated from a model with multiple BART variables, \n"
                    "please provide the model and also the name of the BART variable \n"
                    "for which you want to compute the variable inclusion.")
            free = [var.name for var in model.free_RVs]
            names = bart_var_name if isinstance(bart_var_name, (list, tuple)) else [bart_var_name]
            return sum(inclusion_counts(vi.sel({"variable_inclusion_dim_0": free.index(nm)}).values, n_vars)
                       for nm in names)
        vi = getattr(vi, "values", vi)
    arr = np.asarray(vi)
    if arr.dtype.kind in "OUS":
        return np.array([_decode_vi(str(s), n_vars) for s in arr.ravel()], dtype=np.int64).sum(axis=0)
    arr = arr.astype(np.int64)
    return arr.reshape(-1, n_vars).sum(axis=0)


def _r2_against(full: np.ndarray, part: np.ndarray) -> np.ndarray:
    return np.array([pearsonr2(full[j], part[j]) for j in range(full.shape[0])])


def compute_variable_importance(idata, bartrv, X, model=None, method: str = "VI", fixed: int = 0, samples: int = 50,
                                random_seed=None, *, backend=None) -> dict:
    """Rank the covariates of a fitted BART variable and report how well the model restricted to
    the top-k of them reproduces the full posterior predictions.  Signature of the reference
    (``utils.py:868-877``; uses ``tests/test_bart.py:164,200-203``, ``tests/test_utils.py:78-80``).

    ``idata``: see :func:`inclusion_counts` (ignored by ``method="backward"``).  ``bartrv``: the BART variable,
    its op (needs ``all_trees``), or a list of 1-D BART variables whose predictions are stacked side by side
    (``utils.py:917-922``).  ``model``: only for models with several BART variables.  Returns the reference's
    dictionary: ``indices`` (most important first), ``labels``, ``r2_mean``, ``r2_hdi``, ``preds``, ``preds_all``.
    """
    if method not in ("VI", "backward", "backward_VI"):
        raise ValueError("method must be 'VI', 'backward' or 'backward_VI'")

    def op_of(rv):
        return rv.owner.op if getattr(rv, "owner", None) is not None else rv

    if isinstance(bartrv, list):
        if not all(getattr(rv, "ndim", 1) == 1 for rv in bartrv):
            raise ValueError("List inputs must contain only 1D BART variables")
        sampler = [_get_posterior_sampler(op_of(rv), backend=backend) for rv in bartrv]
        bart_var_name = [getattr(rv, "name", None) for rv in bartrv]
    else:
        sampler = _get_posterior_sampler(op_of(bartrv), backend=backend)
        bart_var_name = getattr(bartrv, "name", None)
    vi = idata
    rng = np.random.default_rng(random_seed)
    if hasattr(X, "columns") and hasattr(X, "to_numpy"):
        names = np.asarray(X.columns).astype(str)
        X = X.to_numpy()
    else:
        X = np.asarray(X, np.float64)
        names = np.arange(X.shape[1]).astype(str)
    p = X.shape[1]
    if method == "backward_VI" and not 1 <= fixed < p:
        raise ValueError("fixed must be greater than 0 and less than the number of variables")

    rows = _resident_rows(sampler, X)  # O(p) .. O(p^2) sweeps over the same rows: uploaded once

    def predict(excluded):
        return _sample_posterior(sampler, X=rows, rng=rng, size=samples,
                                 excluded=None if excluded is None else list(excluded))

    full = predict(None)
    # stage results, least restrictive model last: (excluded set) -> r2 sample, predictions
    order: list[int] = []       # variables, least important first
    stages: list[tuple] = []    # (r2 sample, predictions) once the variables of `order` so far are excluded

    if method in ("VI", "backward_VI"):
        # least used first; the reference's own call (`np.argsort(counts)`, default kind), so that ties fall as there
        by_use = np.argsort(inclusion_counts(vi, p, model, bart_var_name))
        n_by_vi = p if method == "VI" else fixed
        order = [int(v) for v in by_use[:n_by_vi]]
    n_seed = len(order)

    if method == "VI":
        # model k keeps the k + 1 most used variables; the last one keeps all of them
        for keep in range(1, p + 1):
            excl = order[: p - keep]
            pred = predict(excl if excl else None)
            stages.append((_r2_against(full, pred), pred))
        ranked = order[::-1]
    else:
        # backward_VI: the `fixed` least used variables are eliminated by inclusion counts; their
        # models run from "all of them excluded" to "none excluded"
        vi_stages = []
        for j in range(n_seed, 0, -1):
            pred = predict(order[:j])
            vi_stages.append((_r2_against(full, pred), pred))
        if n_seed:
            pred = predict(None)
            vi_stages.append((_r2_against(full, pred), pred))
        # greedy backward search over the remaining variables
        excluded = order[::-1]  # (the reference seeds its list in this order, utils.py:1012)
        back = []
        for stage in range(n_seed + (1 if n_seed else 0), p):
            best = None
            for cand in generate_sequences(p, stage, excluded):
                pred = predict(cand if cand else None)
                r2 = _r2_against(full, pred)
                if best is None or r2.mean() > best[0]:
                    best = (float(r2.mean()), cand, r2, pred)
            back.append((best[2], best[3]))
            for v in best[1]:
                if v not in excluded:
                    excluded.append(int(v))
        excluded += [v for v in range(p) if v not in excluded]
        ranked = excluded[::-1]
        stages = back[::-1] + vi_stages
    r2_mean = np.array([s[0].mean() for s in stages])
    r2_hdi = np.array([hdi(s[0]) for s in stages]).reshape(-1, 2)
    preds = np.array([s[1] for s in stages])
    labels = np.array([names[v] if k == 0 else "+ " + names[v] for k, v in enumerate(ranked)])
    return {
        "indices": np.asarray(ranked),
        "labels": labels,
        "r2_mean": r2_mean,
        "r2_hdi": r2_hdi,
        "preds": preds.squeeze(),
        "preds_all": full.squeeze(),
    }


def get_variable_inclusion(idata, X, model=None, bart_var_name=None, labels=None, to_kulprit: bool = False):
    """Normalised variable inclusion, most used covariate first (reference ``utils.py:747-806``, same signature;
    use: ``tests/test_bart.py:205-208``).

    ``idata``: see :func:`inclusion_counts`; ``model`` / ``bart_var_name`` select the variable when the model has
    several BART variables.  Returns ``(shares, labels)``, or with ``to_kulprit=True`` the nested list of label
    prefixes ``[[], [l0], [l0, l1], ...]`` that Kulprit's ``project`` takes as a path."""
    p = int(np.shape(X)[1])
    counts = inclusion_counts(idata, p, model, bart_var_name).astype(np.float64)
    order = np.argsort(counts / counts.sum())[::-1]
    if hasattr(X, "columns") and hasattr(X, "to_numpy"):
        names = [str(c) for c in np.asarray(X.columns)[order]]
    elif labels is not None:
        names = list(labels)  # as upstream: taken as given, i.e. already in the returned order
    else:
        names = [str(int(i)) for i in order]
    if to_kulprit:
        return [names[:k] for k in range(p + 1)]
    return (counts / counts.sum())[order], names


def vi_to_kulprit(vi_results: dict) -> list:
    """Label prefixes of a :func:`compute_variable_importance` result, "+ " markers removed
    (reference ``utils.py:1093-1108``): ``[[], [l0], [l0, l1], ...]`` without the full model."""
    clean = [str(lab).strip("+ ") for lab in vi_results["labels"]]
    return [clean[:k] for k in range(len(clean))]
