"""The posterior-sampling glue and the variable-importance helpers against vectors produced by RUNNING the
reference's own code (`pymc_bart/utils.py:26-107` `_sample_posterior` / `_MultiChainSampler`, `:1330-1346`
`generate_sequences` / `pearsonr2`; generator `tests/golden/make_utils_golden.py`, which executes the reference's
definitions against a deterministic stand-in for the native `PosteriorSampler`).  Pins, exactly: which draws are
selected from which chain for a given generator state, the order the chains are called in, how a list of samplers
is stacked, and the `(*size, n_rows, n_outputs)` layout (`utils.py:71`)."""
import json
import os

import numpy as np

from pymc_bart_amd.importance import generate_sequences, pearsonr2
from pymc_bart_amd.utils import _MultiChainSampler, _sample_posterior

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "utils_glue.json")))


class FakeChain:
    """The same stand-in the generator used (see its docstring): out[d, k, r] identifies chain, draw, output, row."""

    def __init__(self, chain, n_draws, n_outputs):
        self.chain, self.n_draws, self.n_outputs = chain, n_draws, n_outputs
        self.calls = []

    def sample_posterior(self, X, draw_indices, excluded):
        X = np.asarray(X, dtype=np.float64)
        idx = [int(i) for i in draw_indices]
        self.calls.append(idx)
        ex = 0.0 if not excluded else 0.5 * sum(int(e) + 1 for e in excluded)
        out = np.empty((len(idx), self.n_outputs, X.shape[0]))
        for a, d in enumerate(idx):
            for k in range(self.n_outputs):
                out[a, k] = 1000.0 * self.chain + 10.0 * d + k + ex + 0.001 * X.sum(axis=1)
        return out


def test_sample_posterior_and_multichain_dispatch_match_the_reference():
    for c in GOLD["sample_posterior"]:
        chains = [FakeChain(i, nd, c["K"]) for i, nd in enumerate(c["chains"])]
        sampler = _MultiChainSampler(chains)
        assert sampler.n_draws == c["n_draws"] and sampler.n_outputs == c["K"]
        size = tuple(c["size"]) if isinstance(c["size"], list) else c["size"]
        out = _sample_posterior(sampler, np.array(c["X"]), np.random.default_rng(c["seed"]), size=size,
                                excluded=c["excluded"])
        assert list(out.shape) == c["shape"]
        assert np.array_equal(out.ravel(), np.array(c["out"]))            # same draws, same chains, same layout
        # each chain was asked for the same local draw indices (the reference calls chains in ascending order,
        # each at most once, and skips chains no requested draw falls into)
        assert [ch.calls for ch in chains] == c["calls"]


def test_a_list_of_samplers_is_stacked_along_the_outputs_axis():
    s = GOLD["sampler_list"]
    group = [_MultiChainSampler([FakeChain(0, 4, 1)]), _MultiChainSampler([FakeChain(7, 4, 2)])]
    out = _sample_posterior(group, np.array(s["X"]), np.random.default_rng(s["seed"]), size=s["size"])
    assert list(out.shape) == s["shape"] and np.array_equal(out.ravel(), np.array(s["out"]))


def test_variable_importance_helpers_match_the_reference():
    for g in GOLD["generate_sequences"]:
        assert [list(t) for t in generate_sequences(g["n_vars"], g["i_var"], list(g["include"]))] == g["out"]
    for g in GOLD["pearsonr2"]:
        assert abs(pearsonr2(np.array(g["A"]), np.array(g["B"])) - g["out"]) <= 1e-14 * max(1.0, abs(g["out"]))


# ---------------------------------------------------------------------------------------------------------------
# compute_variable_importance / get_variable_inclusion / vi_to_kulprit against the reference's OWN functions, executed
# (utils.py:747-806, 868-1108; generator tests/golden/make_utils_golden.py::vi_cases).  The stand-ins are the
# generator's: WeightedChain predicts with an excluded column replaced by its mean, FakeVI is the xarray-like stat.
def _vi_setup(c, monkeypatch):
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_utils_golden import FakeVI, WeightedChain, _Named

    import pymc_bart_amd.importance as imp

    X = np.array(c["X"])
    chains = [WeightedChain(i, nd, c["K"], c["p"], 100 * c["seed"] + i) for i, nd in enumerate(c["chains"])]
    sampler = _MultiChainSampler(chains)
    monkeypatch.setattr(imp, "_get_posterior_sampler", lambda op, backend=None: op.sampler)
    n_bart = c.get("n_bart", 1)
    idata = {"sample_stats": {"variable_inclusion": FakeVI(np.array(c["strings"], dtype=object))}}
    names = [f"mu{i}" for i in range(n_bart)]
    rv = type("RV", (), {})()
    rv.owner = type("O", (), {})()
    rv.owner.op = type("Op", (), {"sampler": sampler})()
    rv.name, rv.ndim = names[c.get("which", 0)], c["ndim"]
    model = type("M", (), {"free_RVs": [_Named(nm) for nm in names]})() if n_bart > 1 else None
    return imp, X, idata, rv, model


GOLD_VI = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "variable_importance.json")))


def test_variable_importance_ranking_matches_the_reference_s_own_function(monkeypatch):
    ran = 0
    for c in GOLD_VI["cases"]:
        imp, X, idata, rv, model = _vi_setup(c, monkeypatch)
        out = imp.compute_variable_importance(idata, rv, X, model=model, method=c["method"], fixed=c.get("fixed", 0),
                                              samples=c["samples"], random_seed=c["seed"])
        assert set(out) == {"indices", "labels", "r2_mean", "r2_hdi", "preds", "preds_all"}
        if "raises" in c:
            # the reference's own code stops on this call (ValueError: "backward" on a single-output variable,
            # utils.py:1053; NameError: "backward_VI", utils.py:956-959) -- there is nothing to be equal to; here the
            # call completes with a full ranking
            assert c["raises"] in ("ValueError", "NameError") and sorted(out["indices"]) == list(range(c["p"]))
            assert out["r2_mean"].shape == (c["p"],)
            continue
        g = c["result"]
        ran += 1
        assert [int(v) for v in out["indices"]] == g["indices"], c["name"]
        assert [str(v) for v in out["labels"]] == g["labels"]
        assert list(out["preds"].shape) == g["preds_shape"] and list(out["preds_all"].shape) == g["preds_all_shape"]
        # same draws consumed in the same order: the predictions are the same numbers, not just close ones
        assert np.array_equal(out["preds_all"].ravel(), np.array(g["preds_all"])), c["name"]
        assert np.array_equal(out["preds"].ravel(), np.array(g["preds"])), c["name"]
        np.testing.assert_allclose(out["r2_mean"], g["r2_mean"], rtol=1e-13, atol=0)
        np.testing.assert_allclose(out["r2_hdi"], g["r2_hdi"], rtol=1e-13, atol=0)
        assert imp.vi_to_kulprit(out) == g["kulprit"]
    assert ran >= 5


def test_variable_inclusion_matches_the_reference_s_own_function(monkeypatch):
    for c in GOLD_VI["cases"]:
        imp, X, idata, rv, model = _vi_setup(c, monkeypatch)
        share, labels = imp.get_variable_inclusion(idata, X, model=model, bart_var_name=rv.name)
        assert np.array_equal(share, np.array(c["inclusion"]["share"])) and list(labels) == c["inclusion"]["labels"]
        assert imp.get_variable_inclusion(idata, X, model=model, bart_var_name=rv.name, to_kulprit=True) \
            == c["inclusion"]["kulprit"]
    multi = next(c for c in GOLD_VI["cases"] if c.get("n_bart", 1) > 1)
    imp, X, idata, rv, model = _vi_setup(multi, monkeypatch)
    import pytest

    with pytest.raises(ValueError, match="multiple BART variables"):
        imp.get_variable_inclusion(idata, X)
