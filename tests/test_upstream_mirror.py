"""The oracle's deviations from upstream, MEASURED (round-3 VERDICT #3).

`tests/_upstream_mirror.py` restates upstream's PGBART semantics (Appendix A) in plain NumPy -- sequential RNG,
fresh-particle ``log_weight = 0``, ``systematic(W)[randint]``, missing values filtered before the candidate test,
unbounded trees, full-n ``update_weight`` -- and shares no code with the oracle.  Both samplers run the SAME three
small problems of the reference's own tests over many seeds; what a user sees must agree within Monte-Carlo error:
posterior-mean error, normalised variable inclusion, leaves per tree, tuned ``leaf_sd``, class recovery.

Each problem is run by FOUR samplers (the fourth: the oracle under the upstream-semantics switches of
`include/pgbart_spec.h`, `PGB_COMPAT_*` -- it must land on the mirror as upstream is recalled): the oracle; the mirror with ONE deviation switched on (`fresh_weight="stump"`:
deviation 2, a fresh particle carries the likelihood of its stump) -- the two must agree within Monte-Carlo error,
which bounds the other eleven deviations in distribution; and the mirror as upstream is recalled
(`fresh_weight="zero"`), whose distance from the other two MEASURES deviation 2 (a fresh particle that fails its first
split keeps log-weight 0 upstream and out-weighs every particle that carries a real log-likelihood: whenever one of
the P - 1 particles does not split the root -- probability 1 - 0.95^(P-1) = 37 % at P = 10 -- the tree update ends
in a stump).  The numbers are in DESIGN.md section 0 (`python -m pytest tests/test_upstream_mirror.py -s` prints them)."""
import numpy as np
import pytest

from _upstream_mirror import UpstreamMirror
from pymc_bart_amd.pgbart import PGBART, BARTOp, CategoricalLikelihood, NormalLikelihood

SEEDS = list(range(100, 124))  # 24 seeds per problem and sampler


def _oracle_chain(oracle, X, Y, m, P, tune, draws, seed, family="normal", K=1, rules=None, semantics=None):
    lik = NormalLikelihood(1.0) if family == "normal" else CategoricalLikelihood(K)
    op = BARTOp(X, Y, m=m, split_rules=rules)
    step = PGBART([op], num_particles=P, likelihood=lik, random_seed=seed, backend=oracle, semantics=semantics)
    p = X.shape[1]
    vi, mu, leaves = np.zeros(p), 0.0, []
    for it in range(tune + draws):
        if it == tune:
            step.stop_tuning()
        st, stats = step.astep(None)
        if it >= tune:
            mu = mu + st
            vi += np.asarray(step.last_vi_counts if hasattr(step, "last_vi_counts") else _decode(stats, p))
    step.flush_history()
    _, batches = step._baseline, step._batches
    filled = []
    for b in batches:
        ta = b.decoded() if hasattr(b, "decoded") else b
        for t in range(ta.n_trees):
            lo, hi = ta.node_off[t], ta.node_off[t + 1]
            leaves.append(int((ta.var[lo:hi] < 0).sum()))
            filled.append(int(((ta.var[lo:hi] < 0) & (ta.count[lo:hi] > 0)).sum()))
    # (under the one-hot rule the default sampler never makes an empty leaf -- deviation 13 -- the upstream-semantics
    #  mode does)
    return {"mu": mu / draws, "vi": vi, "leaves": float(np.mean(leaves)), "leaf_sd": step.sampler.state()["leaf_sd"].copy(),
            "filled_leaves": float(np.mean(filled))}


def _decode(stats, p):
    from pymc_bart_amd.utils import _decode_vi

    return _decode_vi(stats[0]["variable_inclusion"], p)


def _mirror_chain(X, Y, m, P, tune, draws, seed, family="normal", K=1, rules=None, fresh="stump"):
    s = UpstreamMirror(X, Y, m=m, num_particles=P, family=family, K=K, split_rules=rules, seed=seed, fresh_weight=fresh)
    p = X.shape[1]
    vi, mu, leaves, filled = np.zeros(p), 0.0, [], []
    for it in range(tune + draws):
        if it == tune:
            s.tune = False
        lo = s.lower
        st, v = s.astep()
        if it >= tune:
            mu = mu + (st[0] if K == 1 else st)
            vi += v
            ids = range(lo, min(lo + s.batch[1], s.m))
            leaves += [s.trees[t].tree.n_leaves() for t in ids]
            filled += [s.trees[t].tree.n_leaves(nonempty=True) for t in ids]
    return {"mu": mu / draws, "vi": vi, "leaves": float(np.mean(leaves)), "leaf_sd": s.leaf_sd.copy(),
            "filled_leaves": float(np.mean(filled))}


def _agree(a, b, what, sigmas=4.0, floor=0.0):
    """Means over the seeds agree within `sigmas` standard errors of their difference (+ an absolute floor)."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    se = np.sqrt(a.var(ddof=1) / a.size + b.var(ddof=1) / b.size)
    d = abs(a.mean() - b.mean())
    assert d <= sigmas * se + floor, f"{what}: oracle {a.mean():.4g} vs mirror {b.mean():.4g} (diff {d:.3g}, se {se:.3g})"
    return a.mean(), b.mean(), se


def _friedman(seed, n=500, p=5):
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (n, p))
    f = 10 * np.sin(np.pi * X[:, 0] * X[:, 1]) + 20 * (X[:, 2] - 0.5) ** 2 + 10 * X[:, 3] + 5 * X[:, 4]
    return X, f + rng.normal(0, 1, n), f


def _three(oracle, problem, seeds, metrics):
    """Run the three samplers over the seeds; returns {sampler: array [seeds, metrics]}."""
    out = {"oracle": [], "mirror+dev2": [], "upstream": [], "oracle/upstream": []}
    for seed in seeds:
        args, kw, truth = problem(seed)
        runs = (("oracle", lambda: _oracle_chain(oracle, *args, seed, **kw)),
                ("mirror+dev2", lambda: _mirror_chain(*args, seed, fresh="stump", **kw)),
                ("upstream", lambda: _mirror_chain(*args, seed, fresh="zero", **kw)),
                # the oracle with both upstream-semantics switches on (pgb_settings.compat, PGB_COMPAT_*): it has to
                # land on the mirror-as-recalled, the way the default lands on the mirror with deviation 2
                ("oracle/upstream", lambda: _oracle_chain(oracle, *args, seed, semantics="upstream", **kw)))
        for key, run in runs:
            out[key].append(metrics(run(), truth))
    return {k: np.array(v) for k, v in out.items()}


def _report(title, names, res):
    print(f"\n{title}")
    for k, a in res.items():
        se = a.std(axis=0, ddof=1) / np.sqrt(len(a))
        print(f"  {k:12s} " + "  ".join(f"{n} {m:.4g} +- {e:.2g}" for n, m, e in zip(names, a.mean(axis=0), se)))


def test_friedman_fit_inclusion_tree_size_and_leaf_sd_against_upstream_semantics(oracle):
    """BASELINE.json configs[0] (Friedman, n = 500, p = 5 + 5 noise columns, m = 50, P = 10), sigma fixed at 1."""
    def problem(seed):
        X, Y, f = _friedman(1000 + seed, p=5)
        X = np.concatenate([X, np.random.default_rng(seed).uniform(0, 1, (X.shape[0], 5))], axis=1)
        return (X, Y, 50, 10, 60, 40), {}, f

    def metrics(r, f):
        share = r["vi"] / max(r["vi"].sum(), 1)
        return (np.sqrt(np.mean((r["mu"] - f) ** 2)), share[:5].sum(), r["leaves"], r["leaf_sd"][0])

    names = ("RMSE(posterior mean, f)", "inclusion share of the informative columns", "leaves per tree", "leaf_sd")
    res = _three(oracle, problem, SEEDS, metrics)
    _report("Friedman n=500 p=10 m=50 P=10, 60 tune + 40 draws, 24 seeds", names, res)
    o, s, z = res["oracle"], res["mirror+dev2"], res["upstream"]
    for i, (name, fl) in enumerate(zip(names, (0.03, 0.02, 0.08, 0.01))):
        _agree(o[:, i], s[:, i], name, floor=fl)  # the other eleven deviations: inside Monte-Carlo error
    assert o[:, 0].mean() < 1.6 and s[:, 0].mean() < 1.6 and o[:, 1].mean() > 0.75  # (sd of f is 4.9)
    # deviation 2, measured: upstream's zero-weight stumps cost it fit and tree size in a chain of this length
    assert z[:, 0].mean() > o[:, 0].mean() + 0.3 and z[:, 2].mean() < o[:, 2].mean() - 0.3
    # ... and switched back: the oracle under the upstream-semantics switches lands on the mirror as recalled
    u = res["oracle/upstream"]
    for i, (name, fl) in enumerate(zip(names, (0.03, 0.02, 0.08, 0.01))):
        _agree(u[:, i], z[:, i], name + " [upstream semantics]", floor=fl)
    assert u[:, 2].mean() < o[:, 2].mean() - 0.3


def test_missing_values_case_against_upstream_semantics(oracle):
    """reference tests/test_bart.py:67-81: n = 50, p = 2, X[10:20, 0] = NaN, m = 10; deviation 6 (a NaN candidate is
    redrawn instead of filtered) and deviation 8 (a failed split sheds the rows with a missing value)."""
    def problem(seed):
        rng = np.random.default_rng(seed)
        X = rng.normal(0, 1, size=(50, 2))
        Y = rng.normal(0, 1, size=50)
        X[10:20, 0] = np.nan
        return (X, Y, 10, 10, 100, 100), {}, Y

    def metrics(r, Y):
        share = r["vi"] / max(r["vi"].sum(), 1)
        return (np.sqrt(np.mean((r["mu"] - Y) ** 2)), share[0], r["leaves"], r["leaf_sd"][0])

    names = ("in-sample RMSE", "inclusion share of the column with NaNs", "leaves per tree", "leaf_sd")
    res = _three(oracle, problem, SEEDS, metrics)
    _report("missing values n=50 p=2 m=10 P=10 (tests/test_bart.py:67-81), 100 tune + 100 draws, 24 seeds", names, res)
    o, s = res["oracle"], res["mirror+dev2"]
    for i, (name, fl) in enumerate(zip(names, (0.04, 0.05, 0.12, 0.015))):
        _agree(o[:, i], s[:, i], name, floor=fl)
        _agree(res["oracle/upstream"][:, i], res["upstream"][:, i], name + " [upstream semantics]", floor=fl)


def test_three_class_softmax_case_against_upstream_semantics(oracle):
    """reference tests/test_bart.py:140-164: 9 rows, 3 classes, m = 2, shape (3, 9); continuous and one-hot rule.
    The reference asserts class recovery; every sampler must deliver it, with trees of the same size."""
    Y = np.array([0, 0, 0, 1, 1, 1, 2, 2, 2], float)
    for rule in ("ContinuousSplit", "OneHotSplit"):
        def problem(seed):
            rng = np.random.default_rng(12345 + seed)
            X = np.concatenate([Y[:, None], rng.integers(0, 6, size=(9, 4))], axis=1).astype(float)
            return (X, Y, 2, 10, 300, 300), {"family": "categorical", "K": 3, "rules": [rule] * 5}, Y

        def metrics(r, Yt):
            return (float((r["mu"].argmax(axis=0) == Yt).mean()), r["filled_leaves"], r["leaf_sd"].mean(), r["leaves"])

        # (a split whose right child is empty -- the split value is the largest of the leaf / its only category:
        #  upstream grows the empty leaf under either rule; this sampler does the same under the continuous rule and
        #  lets the grow FAIL under the one-hot rule (DESIGN.md deviation 13: an empty one-hot leaf would predict 0 for
        #  every unseen category).  So: all leaves are compared under the continuous rule, leaves WITH rows under
        #  the one-hot rule, where the oracle has no others)
        names = ("class recovery", "leaves with rows per tree", "leaf_sd", "leaves incl. empty")
        res = _three(oracle, problem, SEEDS[:16], metrics)
        _report(f"3-class softmax n=9 m=2 P=10 {rule} (tests/test_bart.py:140-164), 300 + 300, 16 seeds", names, res)
        o, s, z = res["oracle"], res["mirror+dev2"], res["upstream"]
        assert o[:, 0].mean() >= 0.95 and s[:, 0].mean() >= 0.95 and z[:, 0].mean() >= 0.9
        if rule == "ContinuousSplit":
            _agree(o[:, 3], s[:, 3], f"leaves per tree ({rule})", floor=0.06)
        else:
            _agree(o[:, 3], s[:, 1], f"leaves with rows per tree ({rule})", floor=0.06)
        _agree(o[:, 2], s[:, 2], f"leaf_sd ({rule})", floor=0.05)
        # upstream semantics on both sides: recovery, leaves with rows, leaf_sd and ALL leaves -- the empty right
        # leaves of one-hot splits included (PGB_COMPAT_ONEHOT_EMPTY_CHILD)
        u = res["oracle/upstream"]
        for i, fl in ((0, 0.05), (1, 0.06), (2, 0.05), (3, 0.06)):
            _agree(u[:, i], z[:, i], f"{names[i]} ({rule}) [upstream semantics]", floor=fl)
