"""The drop-in boundary as the reference's UNMODIFIED callers exercise it.

What the reference does after sampling (``pm.sample_posterior_predictive``, ``tests/test_bart.py:99-101,158-160``;
variable importance / PDP, ``utils.py:923,927``):

* ``BARTRV.rng_fn`` is a *classmethod* (``bart.py:47-49``): it hands ``cls`` -- the per-variable class ``BART_<name>``
  whose CLASS attributes are ``all_trees`` (a ``Manager().list()``), ``X``, ``Y``, ``m``, ... (``bart.py:133-158``) --
  to ``_get_posterior_sampler(cls)`` (``:66``);
* that reads ``op.all_trees``, ``op.m`` and ``op.n_outputs`` and calls
  ``PosteriorSampler.from_history(batches, baseline_forest, op.m, op.n_outputs)`` per chain (``utils.py:122-127``) --
  no split rules, no backend, nothing this package could smuggle in;
* ``_sample_posterior(sampler, X, rng=rng, size=shape)`` (``utils.py:26-71``), then ``pred.squeeze().T`` (``bart.py:68``).

Round 4 failed this path twice: the history carried no split rules (one-hot / subset models were walked
as ``x <= v``: RMSE 2.17 on the training rows) and ``n_outputs`` was set on the op *instance* (``AttributeError`` from
``cls``).  Here the reference's own definitions -- ``_get_posterior_sampler``, ``_sample_posterior``,
``_MultiChainSampler`` and ``BARTRV.rng_fn`` -- are extracted with ``ast`` from ``/root/reference`` and EXECUTED (in the
build container; the GPU box has no reference tree, there the same calls go through this package's mirror of the glue
and are held to vectors the reference-executed run committed: ``tests/golden/reference_callers.json``, generator
``tests/golden/make_reference_callers_golden.py``) against an op double that is a class with class attributes, with
``pymc_bart.pymc_bart.PosteriorSampler`` bound to this package's class.
"""

from __future__ import annotations

import ast
import json
import os
import sys
import types
import warnings
from multiprocessing import Manager

import numpy as np
import pytest

from pymc_bart_amd.pgbart import PGBART, CategoricalLikelihood, NormalLikelihood
from pymc_bart_amd.trees import PosteriorSampler

REF_DIR = "/root/reference/pymc_bart"
HAVE_REF = os.path.exists(os.path.join(REF_DIR, "utils.py"))
GOLD_PATH = os.path.join(os.path.dirname(__file__), "golden", "reference_callers.json")


# ----------------------------------------------------------------------------- the reference's code, executed
def _strip(fn: ast.FunctionDef) -> ast.FunctionDef:
    """Annotations name types of modules that cannot be imported here (pytensor, numpy.typing aliases)."""
    fn.returns = None
    for a in fn.args.args + fn.args.kwonlyargs:
        a.annotation = None
    for node in ast.walk(fn):
        if isinstance(node, ast.AnnAssign):  # `s: int` / `size_iter: tuple[...] = ()` inside _sample_posterior
            node.annotation = ast.copy_location(ast.Name(id="object", ctx=ast.Load()), node.annotation)
    return fn


def reference_glue(sampler_cls):
    """``_get_posterior_sampler``, ``_sample_posterior``, ``_MultiChainSampler`` (``utils.py:26-130``) and
    ``BARTRV.rng_fn`` (``bart.py:47-68``) compiled from the reference's source, with the import inside
    ``_get_posterior_sampler`` (``from pymc_bart.pymc_bart import PosteriorSampler``, ``utils.py:120``) resolving to
    ``sampler_cls``.  Returns the namespace; ``ns["rng_fn"](cls, rng=..., X=..., size=...)`` is the classmethod's
    function."""
    utree = ast.parse(open(os.path.join(REF_DIR, "utils.py")).read())
    names = {"_sample_posterior", "_MultiChainSampler", "_get_posterior_sampler"}
    body = []
    for n in utree.body:
        if isinstance(n, ast.FunctionDef) and n.name in names:
            body.append(_strip(n))
        elif isinstance(n, ast.ClassDef) and n.name in names:
            for sub in n.body:
                if isinstance(sub, ast.FunctionDef):
                    _strip(sub)
            body.append(n)
    assert {n.name for n in body} == names
    btree = ast.parse(open(os.path.join(REF_DIR, "bart.py")).read())
    rv = next(n for n in btree.body if isinstance(n, ast.ClassDef) and n.name == "BARTRV")
    rng_fn = _strip(next(n for n in rv.body if isinstance(n, ast.FunctionDef) and n.name == "rng_fn"))
    rng_fn.decorator_list = []  # @classmethod: called below as rng_fn(cls, ...)
    body.append(rng_fn)

    class _NoTensor:  # isinstance(cls.Y, (TensorSharedVariable, TensorVariable)) -> False for ndarrays
        pass

    ns = {"np": np, "_posterior_sampler_cache": {}, "TensorSharedVariable": _NoTensor, "TensorVariable": _NoTensor}
    exec(compile(ast.fix_missing_locations(ast.Module(body=body, type_ignores=[])), REF_DIR, "exec"), ns)  # noqa: S102
    pkg, mod = types.ModuleType("pymc_bart"), types.ModuleType("pymc_bart.pymc_bart")
    mod.PosteriorSampler = sampler_cls
    pkg.pymc_bart = mod
    ns["_modules"] = {"pymc_bart": pkg, "pymc_bart.pymc_bart": mod}
    return ns


def package_glue(backend=None):
    """The same four pieces as this package mirrors them (what runs where there is no reference tree).  The mirror of
    ``_get_posterior_sampler`` has a ``backend`` test hook the reference's function lacks (None: the HIP library)."""
    from pymc_bart_amd import utils as U

    def get(op):
        return U._get_posterior_sampler(op, backend=backend)

    def rng_fn(cls, rng=None, X=None, Y=None, m=None, alpha=None, beta=None, size=None):  # bart.py:47-68
        if not size:
            size = None
        if not hasattr(cls, "all_trees") or not cls.all_trees:
            Yv = cls.Y
            return np.full((size[0], Yv.shape[0]), Yv.mean()) if size is not None else np.full(Yv.shape[0], Yv.mean())
        shape = size[0] if size is not None else 1
        pred = U._sample_posterior(get(cls), X, rng=rng, size=shape)
        return pred.squeeze().T

    return {"_get_posterior_sampler": get, "_sample_posterior": U._sample_posterior,
            "_MultiChainSampler": U._MultiChainSampler, "rng_fn": rng_fn, "_modules": {}}


def bound_to(backend):
    """``PosteriorSampler`` predicting on ``backend`` (the reference's call names none): on the CPU the oracle is
    the checker; on the GPU box the package's class is bound as it is and predicts through ``libpgbart_hip.so``."""
    if backend is None:
        return PosteriorSampler

    class Bound(PosteriorSampler):
        @classmethod
        def from_history(cls, batches, baseline_forest, m, n_outputs):  # the reference's four arguments
            return super().from_history(batches, baseline_forest, m, n_outputs, backend=backend)

    return Bound


# ----------------------------------------------------------------------------- the op double: a CLASS, as bart.py builds it
class BARTRVDouble:
    """Stands where ``BARTRV`` does: the per-variable classes below derive from it."""

    name = "BART"


def make_bart_class(name, X, Y, m, manager, response="constant", split_rules=None, alpha=0.95, beta=2.0):
    """``type(f"BART_{name}", (BARTRV,), {...})`` of ``bart.py:141-158``: every setting a class attribute, the
    history a ``Manager().list()``; NO ``n_outputs`` (nothing in ``bart.py`` sets it)."""
    return type(f"BART_{name}", (BARTRVDouble,), {
        "name": "BART", "all_trees": manager.list(), "inplace": False, "initval": Y.mean(), "X": X, "Y": Y, "m": m,
        "response": response, "alpha": alpha, "beta": beta, "split_prior": np.array([]), "split_rules": split_rules})


def _data(kind, rng):
    """Small problems, one per kind of history the boundary has to carry."""
    if kind == "continuous":
        X = rng.normal(size=(90, 3))
        Y = np.sin(2 * X[:, 0]) + 0.5 * X[:, 1] + rng.normal(0, 0.1, 90)
        return dict(X=X, Y=Y, m=8, rules=None, K=1)
    if kind == "onehot":  # tests/test_bart.py:140-164 puts OneHotSplit on integer columns
        cat = rng.integers(0, 4, 120)
        X = np.column_stack([cat, rng.integers(0, 5, size=(120, 2))]).astype(float)
        Y = np.array([-2.0, 0.0, 1.0, 3.0])[cat] + rng.normal(0, 0.2, 120)
        return dict(X=X, Y=Y, m=8, rules=["OneHotSplit"] * 3, K=1)
    if kind == "mixed_subset":
        cat = rng.integers(0, 6, 120)
        X = np.column_stack([rng.normal(size=120), cat, rng.integers(0, 3, 120)]).astype(float)
        Y = X[:, 0] + np.where(np.isin(cat, [1, 4]), 2.0, -1.0) + rng.normal(0, 0.2, 120)
        return dict(X=X, Y=Y, m=8, rules=["ContinuousSplit", "SubsetSplit", "OneHotSplit"], K=1)
    if kind == "linear":
        X = rng.normal(size=(100, 2))
        Y = 2 * X[:, 0] - X[:, 1] + rng.normal(0, 0.1, 100)
        return dict(X=X, Y=Y, m=6, rules=None, K=1, response="linear")
    if kind == "softmax3_onehot":  # the reference's categorical test: shape=(3, 9), OneHotSplit (:140-164)
        Y = np.array([0, 0, 0, 1, 1, 1, 2, 2, 2] * 4, float)
        X = np.concatenate([Y[:, None], rng.integers(0, 6, size=(36, 4))], axis=1).astype(float)
        return dict(X=X, Y=Y, m=4, rules=["OneHotSplit"] * 5, K=3)
    raise KeyError(kind)


KINDS = ["continuous", "onehot", "mixed_subset", "linear", "softmax3_onehot"]


def run_chains(kind, backend, manager, chains=2, tune=12, draws=9):
    """``chains`` chains of PGBART on a class-attribute op; returns the class, the per-chain draws of ``sum_trees``
    and the matrix the samplers hold (whole-number continuous columns are jittered: [U] CHANGELOG.md:329-332)."""
    d = _data(kind, np.random.default_rng(20261003))
    cls = make_bart_class("mu", d["X"], d["Y"], d["m"], manager, response=d.get("response", "constant"),
                          split_rules=d["rules"])
    op = cls()  # `bart_op = bart_op_type()` (bart.py:160): the step method holds an INSTANCE (rv.owner.op)
    mus = []
    for c in range(chains):
        lik = NormalLikelihood(0.3) if d["K"] == 1 else CategoricalLikelihood(d["K"])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            step = PGBART([op], num_particles=8, likelihood=lik, observed=d["Y"], random_seed=7, chain=c,
                          backend=backend, batch=(0.5, 0.5))
        out = []
        for it in range(tune + draws):
            if it == tune:
                step.stop_tuning()
            mu, _ = step.astep(None)
            if it >= tune:
                out.append(np.array(mu, copy=True))
        # (the jitter is keyed by the seed and the chain: every chain is compared on its own matrix)
        mus.append((np.stack(out), np.array(step._X, copy=True)))
    return cls, d, mus


def _check_unmodified_path(glue, cls, d, mus, draws=9, tol=1e-9):
    """Everything the reference's callers do with the history, and what has to come out."""
    K, n = d["K"], d["X"].shape[0]
    assert "n_outputs" in vars(cls) and cls.n_outputs == K            # readable from `cls` (bart.py:66 -> utils.py:125)
    sampler = glue["_get_posterior_sampler"](cls)                       # utils.py:113-130, op = the CLASS
    assert sampler.n_draws == draws * len(mus) and sampler.n_outputs == K
    assert glue["_get_posterior_sampler"](cls) is sampler               # cached by id(op) + chain count
    # every stored draw, evaluated on the rows its chain trained on, is what astep returned for that draw
    for c, (mu, Xc) in enumerate(mus):
        idx = list(range(c * draws, (c + 1) * draws))
        pred = sampler.sample_posterior(np.ascontiguousarray(Xc), idx, None)  # (draws, K, n)
        assert pred.shape == (draws, K, n)
        np.testing.assert_allclose(pred if K > 1 else pred[:, 0, :], mu, rtol=0, atol=tol)
    # BARTRV.rng_fn as pm.sample_posterior_predictive reaches it (classmethod, cls): shapes of bart.py:68
    Xnew = np.ascontiguousarray(mus[0][1][:7])
    out = glue["rng_fn"](cls, rng=np.random.default_rng(5), X=Xnew, size=(11,))
    assert out.shape == ((7, 11) if K == 1 else (K, 7, 11))
    one = glue["rng_fn"](cls, rng=np.random.default_rng(5), X=Xnew, size=None)
    assert one.shape == ((7,) if K == 1 else (K, 7))
    # the draws behind it: one rng.integers call over all chains (utils.py:63)
    picks = np.random.default_rng(5).integers(0, sampler.n_draws, size=11).tolist()
    want = sampler.sample_posterior(Xnew, picks, None)                  # (11, K, 7)
    np.testing.assert_array_equal(out, want.transpose(0, 2, 1).squeeze().T)
    return sampler, out


@pytest.fixture(scope="module")
def manager():
    with Manager() as m:
        yield m


@pytest.fixture()
def ref_modules(monkeypatch):
    def install(glue):
        for k, v in glue["_modules"].items():
            monkeypatch.setitem(sys.modules, k, v)
    return install


@pytest.mark.skipif(not HAVE_REF, reason="the reference tree exists in the build container only")
@pytest.mark.parametrize("kind", KINDS)
def test_the_reference_s_own_callers_predict_from_the_history_unmodified(oracle, manager, ref_modules, kind):
    glue = reference_glue(bound_to(oracle))
    ref_modules(glue)
    cls, d, mus = run_chains(kind, oracle, manager)
    sampler, out = _check_unmodified_path(glue, cls, d, mus)
    # ... and the committed vectors are these numbers (the GPU box compares against them)
    gold = json.load(open(GOLD_PATH))["cases"][kind]
    np.testing.assert_allclose(out, np.array(gold["rng_fn"]).reshape(gold["shape"]), rtol=0, atol=1e-12)
    # before round 5 this is what went wrong: the same trees walked as if every split were `x <= v`
    if d["rules"] and any(r != "ContinuousSplit" for r in d["rules"]):
        for ps in sampler._chain_samplers:
            assert (ps.pool.rule[ps.pool.var >= 0] > 0).any()
        mu, Xc = mus[0]
        blind = sampler._chain_samplers[0]
        kept = blind.pool.rule.copy()
        try:
            blind.pool.rule[:] = 0
            bad = blind.sample_posterior(np.ascontiguousarray(Xc), list(range(9)), None)
        finally:
            blind.pool.rule[:] = kept
        assert np.abs((bad if d["K"] > 1 else bad[:, 0, :]) - mu).max() > 0.1


@pytest.mark.parametrize("kind", KINDS)
def test_the_package_s_mirror_of_those_callers_gives_the_same_predictions(oracle, manager, kind):
    glue = package_glue(oracle)
    cls, d, mus = run_chains(kind, oracle, manager)
    _, out = _check_unmodified_path(glue, cls, d, mus)
    gold = json.load(open(GOLD_PATH))["cases"][kind]
    np.testing.assert_allclose(out, np.array(gold["rng_fn"]).reshape(gold["shape"]), rtol=0, atol=1e-12)


def test_n_outputs_lands_on_the_per_variable_class_only(oracle, manager):
    """``n_outputs`` goes on ``type(op)`` when that class is a per-variable ``BART_<name>`` (it carries its own
    ``all_trees``) -- never on a class other variables share."""
    from pymc_bart_amd.pgbart import BARTOp

    rng = np.random.default_rng(0)
    X, Y = rng.normal(size=(30, 2)), rng.normal(size=30)
    a, b = make_bart_class("a", X, Y, 3, manager), make_bart_class("b", X, Y, 3, manager)
    PGBART([a()], num_particles=4, likelihood=CategoricalLikelihood(3), observed=rng.integers(0, 3, 30).astype(float),
           random_seed=1, backend=oracle)
    assert a.n_outputs == 3 and not hasattr(b, "n_outputs") and not hasattr(BARTRVDouble, "n_outputs")
    PGBART([b()], num_particles=4, random_seed=1, backend=oracle)
    assert b.n_outputs == 1 and a.n_outputs == 3
    op = BARTOp(X, Y, m=3)
    PGBART([op], num_particles=4, random_seed=1, backend=oracle)
    assert op.n_outputs == 1 and "n_outputs" not in vars(BARTOp)


def test_a_history_written_before_the_nodes_carried_rules_still_loads(oracle, tmp_path):
    """Format 1 of ``save_history`` kept per-column rules beside the trees; ``load_history`` folds them into the nodes."""
    from pymc_bart_amd.chains import sample_chain
    from pymc_bart_amd.pgbart import BARTOp
    from pymc_bart_amd.trees import _HISTORY_FIELDS, TreeArrays, _as_list, load_history, save_history

    rng = np.random.default_rng(3)
    X = np.column_stack([rng.normal(size=80), rng.integers(0, 3, 80)]).astype(float)
    Y = X[:, 0] + (X[:, 1] == 2) + rng.normal(0, 0.1, 80)
    op = BARTOp(X, Y, m=5, split_rules=["ContinuousSplit", "OneHotSplit"])
    sample_chain(op, tune=10, draws=5, random_seed=2, backend=oracle)
    save_history(tmp_path / "new.npz", op.all_trees, m=5)
    new, m = load_history(tmp_path / "new.npz")
    # the same history in the old layout: no per-node rules, a `rules` vector per column
    out = {"format": np.array("pgbart-history-1"), "n_chains": np.array(1), "m": np.array(5),
           "rules": np.array([0, 1], np.int32)}
    base, batches = op.all_trees[0]
    parts = [base] + _as_list(batches)
    cat = TreeArrays.concat(parts)
    out["c0_n_outputs"] = np.array(1)
    out["c0_sizes"] = np.array([p.n_trees for p in parts], np.int64)
    for f in _HISTORY_FIELDS:
        if f != "rule":
            out[f"c0_{f}"] = getattr(cat, f)
    np.savez_compressed(tmp_path / "old.npz", **out)
    old, m_old = load_history(tmp_path / "old.npz")
    assert m == m_old == 5
    for (b0, bs0), (b1, bs1) in zip(new, old):
        for t0, t1 in zip([b0] + bs0, [b1] + bs1):
            assert np.array_equal(t0.rule, t1.rule) and np.array_equal(t0.var, t1.var)


# ----------------------------------------------------------------------------- the same calls on the MI355X
@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_unmodified_callers_on_the_hip_backend(hip, manager, ref_modules, kind):
    """Chains sampled by ``libpgbart_hip.so``; predictors rebuilt by the reference's four-argument call with this
    package's ``PosteriorSampler`` bound as it ships (default backend = HIP, ``k_predict``); the reference's glue is
    executed when its tree is present, this package's mirror otherwise; either way the result is held to the
    vectors the reference-executed run on the oracle committed."""
    glue = reference_glue(PosteriorSampler) if HAVE_REF else package_glue()
    ref_modules(glue)
    cls, d, mus = run_chains(kind, hip, manager)
    _, out = _check_unmodified_path(glue, cls, d, mus)
    gold = json.load(open(GOLD_PATH))["cases"][kind]
    np.testing.assert_allclose(out, np.array(gold["rng_fn"]).reshape(gold["shape"]), rtol=0, atol=1e-9)
