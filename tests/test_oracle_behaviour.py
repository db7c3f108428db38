"""The CPU oracle against the behavioural pins of the reference's own tests.

The reference pins no numeric outputs on the sampler path, only properties; each test cites the
reference assertion it restates.  Passing these is what licenses the oracle as the checker of
the HIP kernels.
"""
import ctypes as C

import numpy as np
import pytest

from pymc_bart_amd import _abi
from pymc_bart_amd.chains import sample_chain
from pymc_bart_amd.pgbart import PGBART, BARTOp, NormalLikelihood
from pymc_bart_amd.sampler import PyBartSettings, PySampler
from pymc_bart_amd.trees import PosteriorSampler, predict_numpy
from pymc_bart_amd.utils import _decode_vi, _get_posterior_sampler, _sample_posterior


def _sampler(oracle, X, Y, m=10, P=10, seed=3415, rules=None, batch=(0.1, 0.1), prior=None):
    p = X.shape[1]
    st = PyBartSettings.from_data(X, Y, m=m, num_particles=P, seed=seed, batch=batch)
    return PySampler(st, X, Y, np.zeros(p, np.int32) if rules is None else rules,
                     np.ones(p) if prior is None else prior, backend=oracle)


@pytest.mark.filterwarnings("ignore:response=")
@pytest.mark.parametrize("response", ["constant", "linear"])  # as the reference parametrises it
def test_bart_vi_important_variable_dominates(oracle, response):
    # reference tests/test_bart.py:44-64: X[:,0] ~ Y, m=10, tune=draws=200 -> var_imp[0] > rest
    rng = np.random.default_rng(3415)
    X = rng.normal(0, 1, size=(250, 3))
    Y = rng.normal(0, 1, size=250)
    X[:, 0] = rng.normal(Y, 0.1)
    res = sample_chain(BARTOp(X, Y, m=10, response=response), tune=200, draws=200, random_seed=3415,
                       backend=oracle)
    vi_vals = res["variable_inclusion"]
    var_imp = np.array([_decode_vi(v, 3) for v in vi_vals]).sum(axis=0)
    var_imp = var_imp / var_imp.sum()
    assert var_imp[0] > var_imp[1:].sum()
    np.testing.assert_almost_equal(var_imp.sum(), 1)
    assert res["mu"].shape == (200, 250)


@pytest.mark.filterwarnings("ignore:response=")
@pytest.mark.parametrize("response", ["constant", "linear"])
def test_missing_data_samples_without_error(oracle, response):
    # reference tests/test_bart.py:67-81
    rng = np.random.default_rng(0)
    X = rng.normal(0, 1, size=(50, 2))
    Y = rng.normal(0, 1, size=50)
    X[10:20, 0] = np.nan
    res = sample_chain(BARTOp(X, Y, m=10, response=response), tune=100, draws=100, random_seed=3415,
                       backend=oracle)
    assert np.all(np.isfinite(res["mu"]))
    assert res["counters"]["saturations"] == 0


def test_same_seed_same_draws_and_different_seed_differs(oracle):
    rng = np.random.default_rng(1)
    X = rng.normal(size=(300, 4))
    Y = X[:, 0] + rng.normal(0, 0.3, 300)
    a = sample_chain(BARTOp(X, Y, m=8), 20, 20, random_seed=5, backend=oracle)
    b = sample_chain(BARTOp(X, Y, m=8), 20, 20, random_seed=5, backend=oracle)
    c = sample_chain(BARTOp(X, Y, m=8), 20, 20, random_seed=6, backend=oracle)
    assert np.array_equal(a["mu"], b["mu"]) and a["variable_inclusion"] == b["variable_inclusion"]
    assert not np.array_equal(a["mu"], c["mu"])
    d = sample_chain(BARTOp(X, Y, m=8), 20, 20, random_seed=5, chain=1, backend=oracle)
    assert not np.array_equal(a["mu"], d["mu"])  # chains are independent streams


def test_sum_trees_equals_sum_of_tree_predictions(oracle):
    # the sampler's running sum must be the sum of its stored trees evaluated on the training X
    rng = np.random.default_rng(2)
    X = rng.uniform(size=(400, 3))
    Y = np.sin(6 * X[:, 0]) + X[:, 1] + rng.normal(0, 0.2, 400)
    s = _sampler(oracle, X, Y, m=12)
    s.set_likelihood([0.3])
    for it in range(40):
        st, _ = s.step(tune=it < 20)
    forest = s.export_trees(1)
    pred = predict_numpy(forest, np.arange(12)[None, :], X)[0, 0]
    np.testing.assert_allclose(pred, st, rtol=0, atol=1e-9)
    # structural invariants of every tree
    for t in range(forest.n_trees):
        a, b = forest.node_off[t], forest.node_off[t + 1]
        var, left, right, cnt = forest.var[a:b], forest.left[a:b], forest.right[a:b], forest.count[a:b]
        assert cnt[0] == 400
        for k in range(b - a):
            if var[k] >= 0:
                assert cnt[left[k]] + cnt[right[k]] == cnt[k]  # no NaNs -> nothing dropped
                assert left[k] > k and right[k] == left[k] + 1   # breadth-first creation order
            else:
                assert left[k] == -1 and right[k] == -1


def test_sample_posterior_row_subset_consistency(oracle):
    # reference tests/test_utils.py:24-32: same rng seed => prediction on X[:10] equals the first
    # 10 rows of the prediction on X (4 decimals); shapes (2, 50, 1) and (10, 1)
    rng0 = np.random.default_rng(3415)
    X = np.hstack([rng0.normal(0, 1, size=(50, 2)), rng0.binomial(1, 0.5, size=(50, 1))])
    Y = rng0.normal(0, 1, size=50)
    op = BARTOp(X, Y, m=10)
    sample_chain(op, tune=50, draws=50, random_seed=3415, backend=oracle)
    sampler = _get_posterior_sampler(op, backend=oracle)
    assert sampler.n_draws == 50
    pred_all = _sample_posterior(sampler, X=X, rng=np.random.default_rng(3), size=2)
    pred_first = _sample_posterior(sampler, X=X[:10], rng=np.random.default_rng(3))
    np.testing.assert_almost_equal(pred_first, pred_all[0, :10], decimal=4)
    assert pred_all.shape == (2, 50, 1)
    assert pred_first.shape == (10, 1)


def test_posterior_draw_reproduces_the_sampled_sum_trees(oracle):
    # draw d of the history (baseline + batches 0..d) evaluated on the training X is astep's output
    rng = np.random.default_rng(4)
    X = rng.normal(size=(120, 3))
    Y = X[:, 0] ** 2 + rng.normal(0, 0.1, 120)
    op = BARTOp(X, Y, m=6)
    res = sample_chain(op, tune=15, draws=12, random_seed=11, backend=oracle)
    base, batches = res["history"]
    ps = PosteriorSampler.from_history(batches, base, 6, 1, backend=oracle)
    pred = ps.sample_posterior(res["step"].sampler.settings and X, list(range(12)))
    np.testing.assert_allclose(pred[:, 0, :], res["mu"], rtol=0, atol=1e-9)


def test_excluded_variable_is_marginalised(oracle):
    # [U] CHANGELOG.md:410-411: a split on an excluded variable averages both subtrees by count
    rng = np.random.default_rng(5)
    X = rng.normal(size=(200, 2))
    Y = 3 * X[:, 0] + rng.normal(0, 0.1, 200)
    op = BARTOp(X, Y, m=5)
    res = sample_chain(op, tune=30, draws=5, random_seed=1, backend=oracle)
    base, batches = res["history"]
    ps = PosteriorSampler.from_history(batches, base, 5, 1, backend=oracle)
    full = ps.sample_posterior(X, [4])
    excl_all = ps.sample_posterior(X, [4], excluded=[0, 1])
    assert np.allclose(excl_all, excl_all[..., :1])  # no covariate left => constant prediction
    ref = predict_numpy(ps.pool, ps.forest_idx[[4]], X, excluded=[0])
    np.testing.assert_allclose(ps.sample_posterior(X, [4], excluded=[0]), ref, atol=1e-12)
    assert not np.allclose(full, excl_all)
    # NaN in a used covariate behaves like an exclusion for that row
    Xn = X.copy()
    Xn[:, 0] = np.nan
    np.testing.assert_allclose(ps.sample_posterior(Xn, [4]), ps.sample_posterior(X, [4], excluded=[0]),
                               atol=1e-12)


def test_onehot_rule_recovers_a_categorical_effect(oracle):
    # reference tests/test_bart.py:140-164 uses split_rules=["OneHotSplit"]*5 on integer covariates
    rng = np.random.default_rng(12345)
    cat = rng.integers(0, 4, size=300)
    X = np.column_stack([cat, rng.integers(0, 6, size=(300, 2))]).astype(float)
    eff = np.array([-2.0, 0.0, 1.0, 3.0])
    Y = eff[cat] + rng.normal(0, 0.2, 300)
    op = BARTOp(X, Y, m=10, split_rules=["OneHotSplit"] * 3)
    res = sample_chain(op, tune=150, draws=50, random_seed=3415, backend=oracle)
    fit = res["mu"].mean(axis=0)
    assert np.sqrt(np.mean((fit - eff[cat]) ** 2)) < 0.35
    assert res["vi_counts"].sum(axis=0)[0] > res["vi_counts"].sum(axis=0)[1:].sum()
    # one-hot splits never produce an empty right child ([U] needs two distinct values)
    base, batches = res["history"]
    for ta in [base] + batches:
        inner = ta.var >= 0
        assert np.all(ta.count[ta.right[inner] + np.repeat(ta.node_off[:-1], np.diff(ta.node_off))[inner]] > 0)


def test_batch_cursor_visits_every_tree(oracle):
    # [U] astep: batches of max(1, int(0.1 m)) trees, cursor wraps at m
    rng = np.random.default_rng(6)
    X = rng.normal(size=(60, 2))
    Y = rng.normal(size=60)
    s = _sampler(oracle, X, Y, m=25)
    s.set_likelihood([1.0])
    assert s.settings.batch_sizes() == (2, 2)
    seen = []
    for _ in range(13):
        s.step(False)
        seen += list(s.export_trees(0).tree_id)
    assert seen == list(range(25))  # the last batch is cut at m, then the cursor wraps
    assert s.counters.tree_updates == 25 and s.state()["iter"] == 25 and s.state()["lower"] == 0
    s.step(False)
    assert list(s.export_trees(0).tree_id) == [0, 1]


def test_single_tree_and_two_particles_corner(oracle):
    rng = np.random.default_rng(7)
    X = rng.normal(size=(40, 2))
    Y = X[:, 0] + rng.normal(0, 0.1, 40)
    s = _sampler(oracle, X, Y, m=1, P=2)
    s.set_likelihood([0.5])
    for it in range(30):
        st, _ = s.step(tune=it < 10)
    f = s.export_trees(1)
    pred = predict_numpy(f, np.zeros((1, 1), np.int64), X)[0, 0]
    np.testing.assert_allclose(pred, st, atol=1e-12)


def test_tuning_updates_split_prior_and_leaf_sd(oracle):
    # [U] alpha_vec[j] += 1 per split variable while tuning; leaf_sd follows the running std
    rng = np.random.default_rng(8)
    X = rng.normal(size=(200, 3))
    Y = 2 * X[:, 1] + rng.normal(0, 0.2, 200)
    s = _sampler(oracle, X, Y, m=10)
    s.set_likelihood([0.2])
    sd0 = s.state()["leaf_sd"][0]
    assert np.isclose(sd0, Y.std() / np.sqrt(10))
    for _ in range(100):
        s.step(True)
    w = s.split_weights()
    assert w.sum() > 3 and np.all(w >= 1) and np.argmax(w) == 1
    assert s.state()["leaf_sd"][0] != sd0
    frozen = w.copy()
    _, vi = s.step(False)  # draws no longer touch the prior, they count inclusion instead
    assert np.array_equal(s.split_weights(), frozen) and vi.sum() >= 0


def test_leaf_sd_is_the_running_sd_of_the_accepted_trees_recomputed_in_numpy(oracle):
    """[U] RunningSd (Welford per row over the predictions of every accepted tree, count from 1, mean and m2 from 0),
    leaf_sd = mean over the rows of sqrt(m2 / count) from the third tree update on -- recomputed here from the
    exported trees alone and compared with what the sampler tuned itself to."""
    rng = np.random.default_rng(18)
    n, m = 150, 4
    X = rng.normal(size=(n, 3))
    Y = 2 * X[:, 1] + rng.normal(0, 0.3, n)
    st = PyBartSettings.from_data(X, Y, m=m, num_particles=8, seed=5, batch=(1.0, 1.0))
    s = PySampler(st, X, Y, np.zeros(3, np.int32), np.ones(3), backend=oracle)
    s.set_likelihood([0.3])
    rules = np.zeros(3, np.int32)
    count, mean, m2, want = 0, np.zeros(n), np.zeros(n), st.init_leaf_sd
    w_before = s.split_weights()
    for _ in range(7):
        s.step(True)
        step_trees = s.export_trees(0)                       # the m trees this step accepted, in update order
        assert step_trees.n_trees == m
        # [U] alpha_vec[j] += 1 for every split on column j in an accepted tree (while tuning)
        used = np.asarray(step_trees.var)
        w_after = s.split_weights()
        assert np.array_equal(w_after - w_before, np.bincount(used[used >= 0], minlength=3))
        w_before = w_after
        for k in range(m):
            nv = predict_numpy(step_trees, np.array([[k]]), X)[0, 0]
            count += 1
            delta = nv - mean
            mean = mean + delta / count
            m2 = m2 + delta * (nv - mean)
            sd = float(np.mean(np.sqrt(m2 / count)))
            if count > 2 and sd > 0:
                want = sd
        assert s.state()["leaf_sd"][0] == pytest.approx(want, rel=1e-9, abs=1e-9)
    assert want != st.init_leaf_sd


def _scale_equivariance(backend, n=700):
    """Y -> c Y with sigma -> c sigma leaves every likelihood ratio where it was, so a Normal-family chain must come
    out scaled: the same splits, every sum_trees times c -- EXACTLY for a power of two (the fixed-point range
    follows the data), to rounding otherwise.  Nothing in the sampler may depend on the units of the response."""
    rng = np.random.default_rng(3)
    X = rng.normal(size=(n, 4))
    X[rng.random(n) < 0.1, 2] = np.nan
    Y = np.sin(X[:, 0]) + 0.5 * X[:, 1] + rng.normal(0, 0.3, n)

    def run(c):
        st = PyBartSettings.from_data(X, Y * c, m=12, num_particles=8, seed=21, batch=(0.5, 0.5))
        s = PySampler(st, X, Y * c, np.zeros(4, np.int32), np.ones(4), backend=backend)
        s.set_likelihood([0.3 * c])
        mus = np.array([s.step(it < 15)[0] for it in range(30)])
        return st, mus, s.export_trees(1), s.state()["leaf_sd"][0]

    st1, m1, t1, sd1 = run(1.0)
    assert np.std(m1[-1]) > 0.3
    for c in (4.0, 0.125, 3.0):
        stc, mc, tc, sdc = run(c)
        assert np.array_equal(t1.var, tc.var) and np.array_equal(t1.split, tc.split) and np.array_equal(t1.count, tc.count)
        if c != 3.0:
            assert stc.range_exp != st1.range_exp                   # the fixed-point range moved with the data ...
            assert np.array_equal(mc, m1 * c) and sdc == sd1 * c    # ... and the chain is the same chain, exactly
            assert np.array_equal(np.asarray(tc.value), np.asarray(t1.value) * c)
        else:
            np.testing.assert_allclose(mc, m1 * c, rtol=0, atol=1e-12 * np.abs(mc).max())


def _monotone_invariance(backend, n=900):
    """Trees with ContinuousSplit see a column only through the ORDER of its values: x -> g(x), g strictly
    increasing, must leave every partition, every weight and so the whole chain where it was -- bit for bit, with
    each split value mapped through g (constant leaves; NaNs stay NaN; a whole-number column with heavy ties in it).
    On the GPU this also crosses the float32 shadow of the split columns when that path is forced on."""
    rng = np.random.default_rng(4)
    X = rng.normal(size=(n, 4))
    X[rng.random(n) < 0.1, 2] = np.nan
    X[:, 3] = np.round(X[:, 3])
    Y = np.sin(X[:, 0]) + 0.5 * X[:, 1] + 0.3 * X[:, 3] + rng.normal(0, 0.3, n)

    def g(x):
        return np.sinh(x) * 8.0 - 3.0

    Xg = g(X)
    for j in range(4):                                            # g merged no two values in floating point
        assert len(np.unique(X[~np.isnan(X[:, j]), j])) == len(np.unique(Xg[~np.isnan(Xg[:, j]), j]))

    def run(Xv, family, Yv):
        st = PyBartSettings.from_data(Xv, Yv, m=12, num_particles=8, seed=21, batch=(0.5, 0.5), family=family)
        s = PySampler(st, Xv, Yv, np.zeros(4, np.int32), np.ones(4), backend=backend)
        s.set_likelihood([0.3] if family == "normal" else [])
        return np.array([s.step(it < 15)[0] for it in range(30)]), s.export_trees(1)

    for family, Yv in (("normal", Y), ("bernoulli_probit", (Y > Y.mean()).astype(float))):
        m1, t1 = run(X, family, Yv)
        m2, t2 = run(Xg, family, Yv)
        assert np.std(m1[-1]) > 0.2
        assert np.array_equal(m1, m2) and np.array_equal(t1.var, t2.var) and np.array_equal(t1.count, t2.count)
        inner = np.asarray(t1.var) >= 0
        assert np.array_equal(g(np.asarray(t1.split)[inner]), np.asarray(t2.split)[inner])


def test_chain_is_invariant_under_monotone_transforms_of_the_covariates(oracle):
    _monotone_invariance(oracle)


def test_normal_chain_is_equivariant_under_rescaling_of_the_response(oracle):
    _scale_equivariance(oracle)


def test_error_paths(oracle):
    X = np.zeros((10, 2))
    Y = np.zeros(10)
    st = PyBartSettings.from_data(X, Y, m=3, num_particles=1)
    with pytest.raises(_abi.PGBError, match="num_particles"):
        PySampler(st, X, Y, np.zeros(2, np.int32), np.ones(2), backend=oracle)
    st = PyBartSettings.from_data(X, Y, m=3, num_particles=4)
    with pytest.raises(_abi.PGBError, match="split_prior"):
        PySampler(st, X, Y, np.zeros(2, np.int32), np.array([1.0, 0.0]), backend=oracle)
    s = PySampler(st, X, Y, np.zeros(2, np.int32), np.ones(2), backend=oracle)
    with pytest.raises(_abi.PGBError, match="sigma"):
        s.set_likelihood([-1.0])
    with pytest.raises(ValueError):
        BARTOp(X, Y, response="quadratic")
    with pytest.raises(NotImplementedError):
        PGBART([BARTOp(X, Y, split_rules=["NoSuchSplit", "ContinuousSplit"])], backend=oracle)
    with pytest.raises(_abi.PGBError, match="unknown split rule"):
        PySampler(st, X, Y, np.array([0, 7], np.int32), np.ones(2), backend=oracle)


def test_constant_response_and_duplicate_rows(oracle):
    # degenerate data: all rows identical -> continuous splits put everything left; must not hang
    X = np.ones((64, 2))
    Y = np.full(64, 2.5)
    s = _sampler(oracle, X, Y, m=4, P=5)
    s.set_likelihood([1.0])
    for it in range(10):
        st, _ = s.step(it < 5)
    assert np.all(np.isfinite(st))


def test_pgbart_astep_contract(oracle):
    # astep -> (sum_trees ndarray (n,), [ {"variable_inclusion": b64, "tune": bool} ])
    rng = np.random.default_rng(9)
    X = rng.normal(size=(30, 2))
    Y = rng.normal(size=30)
    op = BARTOp(X, Y, m=3)
    step = PGBART([op], num_particles=5, likelihood=NormalLikelihood(1.0), random_seed=1, backend=oracle)
    assert PGBART.competence(op) and not PGBART.competence(object())
    assert step.generates_stats and "variable_inclusion" in step.stats_dtypes_shapes
    out, stats = step.astep(None)
    assert out.shape == (30,) and stats[0]["tune"] is True
    assert _decode_vi(stats[0]["variable_inclusion"], 2) == [0, 0]  # no inclusion counts while tuning
    assert len(op.all_trees) == 0
    step.stop_tuning()
    point, stats = step.step({"sigma": 1.0})
    assert stats[0]["tune"] is False and point["mu"].shape == (30,)
    assert len(op.all_trees) == 1 and op.n_outputs == 1
    step.astep(None)
    base, batches = op.all_trees[0]
    assert base.n_trees == 3 and len(batches) == 2


def test_fixed_point_saturation_is_reported(oracle):
    # a deliberately tiny range: the sampler must refuse to return silently wrong sums
    rng = np.random.default_rng(10)
    X = rng.normal(size=(100, 2))
    Y = 50.0 * X[:, 0] + rng.normal(size=100)
    st = PyBartSettings.from_data(X, Y, m=5, num_particles=5)
    st.range_exp = 2
    s = PySampler(st, X, Y, np.zeros(2, np.int32), np.ones(2), backend=oracle)
    s.set_likelihood([1.0])
    with pytest.raises(_abi.PGBError, match="saturation"):
        for _ in range(5):
            s.step(True)


@pytest.mark.parametrize("split_rule", ["ContinuousSplit", "OneHotSplit"])
def test_categorical_model_recovers_classes(oracle, split_rule):
    # reference tests/test_bart.py:140-164: 3-class softmax, shape=(3, 9), m=2, tune=draws=600:
    # the most probable class equals Y for all 9 rows; astep returns (K, n)
    from pymc_bart_amd.pgbart import CategoricalLikelihood

    Y = np.array([0, 0, 0, 1, 1, 1, 2, 2, 2])
    rng = np.random.default_rng(12345)
    X = np.concatenate([Y[:, None], rng.integers(0, 6, size=(9, 4))], axis=1)
    op = BARTOp(X, Y, m=2, split_rules=[split_rule] * 5)
    step = PGBART([op], num_particles=10, likelihood=CategoricalLikelihood(3), random_seed=3415,
                  backend=oracle)
    assert step.shape == (3, 9)
    votes = np.zeros((3, 9))
    for it in range(1200):
        if it == 600:
            step.stop_tuning()
        lo, stats = step.astep(None)
        assert lo.shape == (3, 9)
        if it >= 600:
            p = np.exp(lo - lo.max(axis=0))
            votes += p / p.sum(axis=0)
    assert (votes.argmax(axis=0) == Y).all()
    assert op.n_outputs == 3
    # stored trees reproduce the K-vector prediction (reference shape (…, 9, 3) after transpose)
    base, batches = op.all_trees[0]
    ps = PosteriorSampler.from_history(batches, base, 2, 3, backend=oracle)
    pred = ps.sample_posterior(step.sampler and np.asarray(X, float), [599])
    assert pred.shape == (1, 3, 9)
    # X was jittered for the continuous rule inside PGBART: compare on the sampler's own matrix
    if split_rule == "OneHotSplit":
        np.testing.assert_allclose(pred[0], lo, atol=1e-9)


@pytest.mark.filterwarnings("ignore:response=")
@pytest.mark.parametrize("response", ["constant", "linear"], ids=["constant", "linear-response"])
def test_shape_two_outputs_mean_and_scale(oracle, response):
    # reference tests/test_bart.py:107-123: w = BART(shape=(2, 250)); y ~ Normal(w[0], |w[1]|):
    # astep returns (2, n); the scale output must pick up heteroscedasticity
    from pymc_bart_amd.pgbart import NormalMeanScaleLikelihood

    rng = np.random.default_rng(3)
    X = rng.normal(0, 1, size=(250, 3))
    sd = np.where(X[:, 0] > 0, 2.0, 0.3)
    Y = X[:, 1] + sd * rng.normal(0, 1, size=250)
    op = BARTOp(X, Y, m=10, response=response)
    step = PGBART([op], num_particles=10, likelihood=NormalMeanScaleLikelihood(), random_seed=3415,
                  backend=oracle)
    assert step.shape == (2, 250)
    acc = np.zeros((2, 250))
    for it in range(400):
        if it == 200:
            step.stop_tuning()
        w, _ = step.astep(None)
        assert w.shape == (2, 250)
        if it >= 250:
            acc += np.abs(w) if False else np.stack([w[0], np.abs(w[1])])
    acc /= 150
    assert np.corrcoef(acc[0], X[:, 1])[0, 1] > 0.7
    assert acc[1][X[:, 0] > 0].mean() > 1.5 * acc[1][X[:, 0] <= 0].mean()


def test_multiple_bart_variables_manual_step(oracle):
    # reference tests/test_bart.py:211-241: two BART variables with their own PGBART step methods
    # (num_particles=5) in one additive Normal model; both are sampled, shapes (draws, n), separate
    # all_trees objects
    rng = np.random.default_rng(0)
    X1 = rng.normal(0, 1, size=(60, 2))
    X2 = rng.normal(0, 1, size=(60, 2))
    Y = 2.0 * X1[:, 0] - 1.5 * X2[:, 1] + rng.normal(0, 0.1, size=60)
    mu1 = BARTOp(X1, Y, m=8, name="mu1")
    mu2 = BARTOp(X2, Y, m=8, name="mu2")
    lik = NormalLikelihood("sigma")
    step1 = PGBART([mu1], num_particles=5, likelihood=lik, random_seed=3415, backend=oracle)
    step2 = PGBART([mu2], num_particles=5, likelihood=lik, random_seed=3416, backend=oracle)
    cur1 = np.full(60, Y.mean() / 2)
    cur2 = np.full(60, Y.mean() / 2)
    d1, d2 = [], []
    for it in range(200):
        if it == 100:
            step1.stop_tuning()
            step2.stop_tuning()
        point = {"sigma": 0.3}
        cur1, _ = step1.astep(None, point, offset=cur2)
        cur2, _ = step2.astep(None, point, offset=cur1)
        if it >= 100:
            d1.append(cur1)
            d2.append(cur2)
    d1, d2 = np.array(d1), np.array(d2)
    assert d1.shape == (100, 60) and d2.shape == (100, 60)
    assert mu1.all_trees is not mu2.all_trees and len(mu1.all_trees) == 1 and len(mu2.all_trees) == 1
    # each variable picks up ITS covariate's effect; the sum explains Y
    assert np.corrcoef(d1.mean(0), X1[:, 0])[0, 1] > 0.8
    assert np.corrcoef(d2.mean(0), -X2[:, 1])[0, 1] > 0.8
    assert np.sqrt(np.mean((d1.mean(0) + d2.mean(0) - Y) ** 2)) < 0.6


def test_checkpoint_rejects_foreign_images(oracle):
    rng = np.random.default_rng(2)
    X = rng.normal(size=(200, 3))
    Y = X[:, 0] + rng.normal(0, 0.1, 200)
    mk = lambda m: PySampler(PyBartSettings.from_data(X, Y, m=m, num_particles=5, seed=1), X, Y,  # noqa: E731
                             np.zeros(3, np.int32), np.ones(3), backend=oracle)
    a, b = mk(5), mk(6)
    a.set_likelihood([1.0])
    a.step(True)
    blob = a.checkpoint()
    with pytest.raises(_abi.PGBError, match="settings differ"):
        b.restore(blob)
    with pytest.raises(_abi.PGBError, match="not a pgbart checkpoint"):
        a.restore(b"x" * len(blob))
    with pytest.raises(_abi.PGBError, match="truncated"):
        a.restore(blob[:16])
    a.restore(blob)  # its own image is fine


def test_pgbart_pickle_round_trip_resumes_the_chain(oracle, monkeypatch):
    """the step method must be picklable (PyMC sends it to worker processes).  A
    step method pickled mid-run continues exactly where it was."""
    import pickle

    from pymc_bart_amd import sampler as sampler_mod

    monkeypatch.setattr(sampler_mod, "_DEFAULT_BACKEND", oracle)  # what default_backend() returns
    rng = np.random.default_rng(4)
    X = rng.normal(size=(400, 4))
    Y = np.sin(2 * X[:, 0]) + rng.normal(0, 0.2, 400)

    def run(pickle_at):
        step = PGBART([BARTOp(X, Y, m=10)], num_particles=6, likelihood=NormalLikelihood("sigma"),
                      random_seed=9, backend=oracle)
        out = []
        for it in range(12):
            if it == 6:
                step.stop_tuning()
            if it in pickle_at:
                step = pickle.loads(pickle.dumps(step))
            mu, stats = step.astep(None, {"sigma": 0.7})
            out.append((mu, stats[0]["variable_inclusion"]))
        return out, step

    a, sa = run(())
    b, sb = run((0, 3, 9))
    for (m1, v1), (m2, v2) in zip(a, b):
        assert np.array_equal(m1, m2) and v1 == v2
    # the tree history travels with the pickle too
    assert len(sb._batches) == len(sa._batches) == 6
    assert np.array_equal(sa._batches[-1].value, sb._batches[-1].value)


def test_subset_rule_recovers_a_set_valued_effect(oracle):
    # bart.py:100-103 names SubsetSplitRule next to OneHotSplitRule: the split sends a SET of
    # categories left.  The effect below is constant on {1, 4, 6} vs the rest: one subset split
    # captures it, one-hot splits need several.
    rng = np.random.default_rng(77)
    cat = rng.integers(0, 7, size=400)
    X = np.column_stack([cat, rng.integers(0, 5, size=400), rng.normal(size=400)]).astype(float)
    f = np.where(np.isin(cat, [1, 4, 6]), 2.0, -1.0)
    Y = f + rng.normal(0, 0.2, 400)
    op = BARTOp(X, Y, m=10, split_rules=["SubsetSplit", "SubsetSplit", "ContinuousSplit"])
    res = sample_chain(op, tune=100, draws=40, random_seed=3415, backend=oracle)
    assert np.sqrt(np.mean((res["mu"].mean(axis=0) - f) ** 2)) < 0.3
    vi = res["vi_counts"].sum(axis=0)
    assert vi[0] > vi[1] + vi[2]
    # set-valued split values: integer masks below 2^52, no empty child, and the history
    # evaluated on the training rows reproduces astep's output
    base, batches = res["history"]
    rules = np.array([2, 2, 0], np.int32)
    for ta in [base] + batches:
        sub = (ta.var == 0) | (ta.var == 1)
        assert np.all(ta.split[sub] == np.floor(ta.split[sub])) and np.all(ta.split[sub] < 2.0 ** 52)
        assert np.all(ta.split[sub] >= 1)
        off = np.repeat(ta.node_off[:-1], np.diff(ta.node_off))
        inner = ta.var >= 0
        assert np.all(ta.count[ta.left[inner] + off[inner]] > 0)
        assert np.all(ta.count[ta.right[inner] + off[inner]] > 0)
    ps = PosteriorSampler.from_history(batches, base, 10, 1, backend=oracle)
    pred = ps.sample_posterior(X, list(range(40)))
    np.testing.assert_allclose(pred[:, 0, :], res["mu"], rtol=0, atol=1e-9)
    ref = predict_numpy(ps.pool, ps.forest_idx[[7]], X)
    np.testing.assert_allclose(pred[7], ref[0], atol=1e-12)
    # invalid category codes are rejected up front
    Xbad = X.copy()
    Xbad[0, 0] = 60.0
    with pytest.raises(ValueError, match="SubsetSplit column 0"):
        PGBART([BARTOp(Xbad, Y, m=2, split_rules=["SubsetSplit"] * 2 + ["ContinuousSplit"])],
               backend=oracle)


@pytest.mark.parametrize("alpha,beta,P", [(0.95, 2.0, 10), (0.5, 1.0, 5), (0.95, 1.0, 20), (0.95, 2.0, 2)])
def test_flat_likelihood_reproduces_the_tree_prior(oracle, alpha, beta, P):
    """An oracle-INDEPENDENT check of the particle machinery (proposal, weights, resampling, final pick, batch
    cursor): with a likelihood that cannot tell trees apart (sigma = 1e6) the chain's trees must be draws from the
    prior of Chipman et al. (reference bart.py:108-113): a node at depth d splits with probability
    alpha (1 + d)^-beta.  Expected number of leaves of that Galton-Watson tree: L_d = (1 - p_d) + 2 p_d L_{d+1}."""
    L = 1.0
    for d in range(60, -1, -1):
        p_d = alpha * (1 + d) ** -beta
        L = (1 - p_d) + 2 * p_d * L
    rng = np.random.default_rng(0)
    n, p, m = 4000, 3, 40
    X, Y = rng.normal(size=(n, p)), rng.normal(size=n)
    st = PyBartSettings.from_data(X, Y, m=m, num_particles=P, seed=7, batch=(1.0, 1.0), alpha=alpha, beta=beta)
    s = PySampler(st, X, Y, np.zeros(p, np.int32), np.ones(p), backend=oracle)
    s.set_likelihood([1e6])
    leaves, stumps = [], []
    for it in range(120):
        s.step(False)
        if it >= 20 and it % 4 == 0:
            t = s.export_trees(1)
            off = np.append(np.asarray(t.node_off), t.total_nodes)
            for k in range(m):
                v = np.asarray(t.var)[off[k]:off[k + 1]]
                leaves.append(int((v < 0).sum()))
                stumps.append(len(v) == 1)
    se = np.std(leaves) / np.sqrt(len(leaves) / 4.0)           # (draws 4 steps apart are still correlated)
    assert abs(np.mean(leaves) - L) < 4 * se + 0.02, (np.mean(leaves), L)
    assert abs(np.mean(stumps) - (1 - alpha)) < 0.04, np.mean(stumps)


def test_flat_likelihood_reproduces_the_split_prior_and_uniform_split_rows(oracle):
    """Same device, one level down: under a flat likelihood the split VARIABLE of the root follows `split_prior`
    (reference bart.py:96-99) and its split VALUE is the value of a uniformly chosen row, i.e. its rank among the
    column's values is uniform."""
    rng = np.random.default_rng(1)
    n, m = 3000, 40
    X, Y = rng.normal(size=(n, 4)), rng.normal(size=n)
    prior = np.array([3.0, 1.0, 1.0, 0.5])
    st = PyBartSettings.from_data(X, Y, m=m, num_particles=8, seed=11, batch=(1.0, 1.0))
    s = PySampler(st, X, Y, np.zeros(4, np.int32), prior, backend=oracle)
    s.set_likelihood([1e6])
    ranks = np.sort(X, axis=0)
    var_counts, quantiles = np.zeros(4), []
    for it in range(140):
        s.step(False)
        if it >= 20 and it % 4 == 0:
            t = s.export_trees(1)
            for k in np.asarray(t.node_off)[:m]:
                j = int(np.asarray(t.var)[k])
                if j >= 0:
                    var_counts[j] += 1
                    quantiles.append(np.searchsorted(ranks[:, j], np.asarray(t.split)[k]) / n)
    freq = var_counts / var_counts.sum()
    assert np.max(np.abs(freq - prior / prior.sum())) < 0.04, freq
    q = np.asarray(quantiles)
    assert abs(q.mean() - 0.5) < 0.03 and abs(np.mean(q < 0.25) - 0.25) < 0.04 and abs(np.mean(q > 0.75) - 0.25) < 0.04


def test_a_chain_whose_first_updates_keep_the_stump_does_not_die(oracle):
    """Deviation 12.  leaf_sd is tuned to the running sd of the accepted trees' predictions from the third
    tree update on ([U] RunningSd); if the untouched stump wins the first three updates that sd is exactly 0,
    and a leaf value is mean(sum_trees) / m + N(0, 1) leaf_sd: with leaf_sd = 0 no leaf ever moves again.
    This key did exactly that (found as a 1-in-35 flake of the step-method test; 62 of 150 keys at m = 2, P = 3)."""
    rng = np.random.default_rng(5)
    n = 200
    X = rng.normal(size=(n, 2))
    z = rng.normal(size=n)
    f = np.where(X[:, 0] > 0, 1.0, -1.0)
    Y = f + 2.0 * z + rng.normal(0, 0.2, n)
    st = PyBartSettings.from_data(X, Y, m=8, num_particles=6, seed=3330035369473741674, batch=(1.0, 1.0))
    s = PySampler(st, X, Y, np.zeros(2, np.int32), np.ones(2), backend=oracle)
    s.set_likelihood([0.2])
    for _ in range(10):
        mu, _ = s.step(True)
    assert s.state()["leaf_sd"][0] > 0.0 and mu.std() > 0.5 and np.corrcoef(mu, Y)[0, 1] > 0.3
    dead = 0
    Xs, Ys = X[:120], Y[:120]
    for seed in range(60):                      # tiny forests, few particles: the stump wins often
        st = PyBartSettings.from_data(Xs, Ys, m=2, num_particles=3, seed=seed)
        s = PySampler(st, Xs, Ys, np.zeros(2, np.int32), np.ones(2), backend=oracle)
        s.set_likelihood([1.0])
        for _ in range(6):
            s.step(True)
        dead += s.state()["leaf_sd"][0] == 0.0
    assert dead == 0
    # K-vector leaves: no output's leaf_sd may collapse either
    Yc = (rng.random(n) < 0.5).astype(float) + (X[:, 0] > 0)
    for seed in range(20):
        st = PyBartSettings.from_data(X, Yc, m=2, num_particles=3, seed=seed, family="categorical", n_outputs=3)
        s = PySampler(st, X, Yc, np.zeros(2, np.int32), np.ones(2), backend=oracle)
        s.set_likelihood([])
        for _ in range(6):
            s.step(True)
        assert np.all(s.state()["leaf_sd"] > 0.0), seed


def _subset_codes_are_checked_by_the_library(backend):
    """At the ABI itself (PySampler, no PGBART in front): a SubsetSplit column must hold integer codes 0 .. 51 or
    NaN -- 52, a fraction or a negative code would be CLAMPED by the split rule, so pgb_set_data refuses them."""
    rng = np.random.default_rng(2)
    X = rng.normal(size=(300, 3))
    X[:, 1] = rng.integers(0, 52, 300)
    X[::7, 1] = np.nan
    Y = rng.normal(size=300)
    rules = np.array([0, 2, 0], np.int32)
    st = PyBartSettings.from_data(X, Y, m=3, num_particles=4)
    PySampler(st, X, Y, rules, np.ones(3), backend=backend).step(True)       # 0 .. 51 and NaN are fine
    for bad in (52.0, 3.5, -1.0, np.inf):
        Xb = X.copy()
        Xb[5, 1] = bad
        with pytest.raises(_abi.PGBError, match="SubsetSplit column 1"):
            PySampler(st, Xb, Y, rules, np.ones(3), backend=backend)
    Xc = X.copy()
    Xc[5, 0] = 99.5                                                           # other rules take any value
    PySampler(st, Xc, Y, rules, np.ones(3), backend=backend).step(True)


def test_subset_codes_are_checked_by_the_library(oracle):
    _subset_codes_are_checked_by_the_library(oracle)


def test_variable_importance_ranks_the_informative_covariates(oracle):
    # SURVEY.md 8f f4 / reference utils.py:868-1090: "VI" follows the inclusion counts, "backward"
    # the greedy elimination by squared correlation with the full prediction
    from pymc_bart_amd.importance import (compute_variable_importance, generate_sequences, hdi,
                                          inclusion_counts, pearsonr2)

    rng = np.random.default_rng(21)
    X = rng.normal(size=(300, 5))
    Y = 3.0 * X[:, 1] + 1.5 * np.sin(2 * X[:, 3]) + rng.normal(0, 0.2, 300)
    op = BARTOp(X, Y, m=20)
    res = sample_chain(op, tune=120, draws=60, random_seed=5, backend=oracle)
    stats = res["variable_inclusion"]
    counts = inclusion_counts(stats, 5)
    assert np.array_equal(counts, res["vi_counts"].sum(axis=0))
    assert np.array_equal(inclusion_counts({"sample_stats": {"variable_inclusion": np.array(stats)}}, 5), counts)

    out = compute_variable_importance(stats, op, X, method="VI", samples=20, random_seed=1, backend=oracle)
    assert set(out) == {"indices", "labels", "r2_mean", "r2_hdi", "preds", "preds_all"}
    assert list(out["indices"]) == list(np.argsort(counts, kind="stable")[::-1])
    assert set(out["indices"][:2]) == {1, 3}
    assert out["r2_mean"].shape == (5,) and out["r2_hdi"].shape == (5, 2)
    assert out["preds"].shape == (5, 20, 300) and out["preds_all"].shape == (20, 300)
    assert out["labels"][0] == str(out["indices"][0]) and out["labels"][1].startswith("+ ")
    assert out["r2_mean"][1] > 0.9 and out["r2_mean"][1] > out["r2_mean"][0]   # top-2 model explains the fit
    assert np.all(out["r2_hdi"][:, 0] <= out["r2_mean"] + 1e-12) and np.all(out["r2_mean"] <= out["r2_hdi"][:, 1] + 1e-12)

    back = compute_variable_importance(None, op, X, method="backward", samples=15, random_seed=2, backend=oracle)
    assert set(back["indices"][:2]) == {1, 3} and sorted(back["indices"]) == [0, 1, 2, 3, 4]
    assert back["r2_mean"].shape == (5,) and back["r2_mean"][-1] > 0.8
    assert np.all(np.diff(back["r2_mean"][:3]) > -0.05)  # adding variables back does not hurt

    bvi = compute_variable_importance(stats, op, X, method="backward_VI", fixed=2, samples=10,
                                      random_seed=3, backend=oracle)
    assert sorted(bvi["indices"]) == [0, 1, 2, 3, 4] and bvi["r2_mean"].shape == (5,)
    assert set(bvi["indices"][-2:]) == set(np.argsort(counts, kind="stable")[:2])  # the fixed, least used ones
    with pytest.raises(ValueError):
        compute_variable_importance(stats, op, X, method="nope", backend=oracle)
    with pytest.raises(ValueError):
        compute_variable_importance(stats, op, X, method="backward_VI", fixed=5, backend=oracle)

    # helpers (reference utils.py:1330-1346)
    assert generate_sequences(4, 0, []) == [()]
    assert generate_sequences(4, 2, [1]) == [(1, 0), (1, 2), (1, 3)]
    a = rng.normal(size=50)
    assert abs(pearsonr2(a, 2 * a + 1) - 1.0) < 1e-12 and pearsonr2(a, rng.normal(size=50)) < 0.3
    lo, hi = hdi(rng.normal(size=4000), 0.94)
    assert -2.2 < lo < -1.6 and 1.6 < hi < 2.2


def test_posterior_sampler_cache_is_not_fooled_by_reused_ids_or_growing_chains(oracle):
    # the reference caches on id(op) + number of chains (utils.py:110-130); ids are reused by later
    # objects and chains grow, so the cache here must notice both
    import gc

    rng = np.random.default_rng(8)
    X = rng.normal(size=(60, 2))
    seen = {}
    for k in range(40):  # many short-lived ops: some will reuse an earlier id
        Y = (k + 1) * 10.0 + X[:, 0]
        op = BARTOp(X, Y, m=3)
        sample_chain(op, tune=2, draws=2, random_seed=k, backend=oracle)
        s = _get_posterior_sampler(op, backend=oracle)
        pred = _sample_posterior(s, X=X, rng=np.random.default_rng(0), size=2)
        assert abs(pred.mean() - Y.mean()) < 5.0, "stale sampler of an earlier op"
        seen[id(op)] = seen.get(id(op), 0) + 1
        del op, s
        gc.collect()
    assert max(seen.values()) > 1, "the loop is meant to recycle ids"
    # a chain that grows invalidates the entry
    Y = X[:, 0]
    op = BARTOp(X, Y, m=3)
    step = PGBART([op], num_particles=4, likelihood=NormalLikelihood(1.0), backend=oracle)
    step.stop_tuning()
    step.astep(None, {})
    assert _get_posterior_sampler(op, backend=oracle).n_draws == 1
    step.astep(None, {})
    assert _get_posterior_sampler(op, backend=oracle).n_draws == 2


def test_tree_history_round_trips_through_a_file(oracle, tmp_path):
    # an on-disk format for (baseline_forest, batches) per chain
    from pymc_bart_amd.trees import load_history, save_history

    rng = np.random.default_rng(31)
    X = rng.normal(size=(150, 3))
    X[:, 2] = rng.integers(0, 4, 150)
    Y = X[:, 0] + (X[:, 2] == 1) + rng.normal(0, 0.2, 150)
    op = BARTOp(X, Y, m=6, split_rules=["ContinuousSplit", "ContinuousSplit", "OneHotSplit"])
    for c in range(2):
        sample_chain(op, tune=10, draws=7, random_seed=3, chain=c, backend=oracle)
    assert len(op.all_trees) == 2
    ref = _sample_posterior(_get_posterior_sampler(op, backend=oracle), X, np.random.default_rng(5), size=9)
    path = tmp_path / "history.npz"
    save_history(path, op.all_trees, m=6)
    all_trees, m = load_history(path)
    assert m == 6 and len(all_trees) == 2
    for (b0, bs0), (b1, bs1) in zip(op.all_trees, all_trees):
        assert len(bs0) == len(bs1) == 7
        for t0, t1 in zip([b0] + list(bs0), [b1] + list(bs1)):
            for f in ("tree_id", "node_off", "var", "split", "left", "right", "count", "value", "rule"):
                assert np.array_equal(getattr(t0, f), getattr(t1, f)), f
            assert np.array_equal(t1.rule, np.where(t1.var == 2, 1, 0))  # the one-hot column's splits say so
    op2 = BARTOp(X, Y, m=6, all_trees=all_trees)  # (no split rules on the op: the trees carry them)
    op2.n_outputs = 1
    again = _sample_posterior(_get_posterior_sampler(op2, backend=oracle), X, np.random.default_rng(5), size=9)
    assert np.array_equal(ref, again)
    with pytest.raises(ValueError):
        np.savez(tmp_path / "bad.npz", format=np.array("x"))
        load_history(tmp_path / "bad.npz")


@pytest.mark.parametrize("response", ["linear", "mix"])
def test_linear_response_fits_slopes_and_stays_consistent(oracle, response):
    # reference tests parametrise response=["constant", "linear"] (tests/test_bart.py:44-123); leaves
    # then predict value + slope * (x[split variable of the parent] - xbar)  ([U] fast_linear_fit)
    rng = np.random.default_rng(0)
    n = 600
    X = rng.uniform(-2, 2, size=(n, 3))
    f = np.where(X[:, 0] < 0, 2 * X[:, 0] + 1, -1.5 * X[:, 0] + 1) + 0.5 * X[:, 1]
    Y = f + rng.normal(0, 0.1, n)

    def run(resp):
        st = PyBartSettings.from_data(X, Y, m=10, num_particles=10, seed=3, response=resp)
        s = PySampler(st, X, Y, np.zeros(3, np.int32), np.ones(3), backend=oracle)
        s.set_likelihood([0.1])
        for it in range(120):
            mu, _ = s.step(it < 60)
        return s, mu

    s, mu = run(response)
    _, mu_const = run("constant")
    rmse = lambda a: float(np.sqrt(np.mean((a - f) ** 2)))  # noqa: E731
    assert rmse(mu) < 0.9 * rmse(mu_const)        # piecewise-linear target: slopes help
    forest = s.export_trees(1)
    leaves = forest.var < 0
    assert (forest.svar[leaves] >= 0).sum() > 5 and np.all(forest.svar[~leaves] == -1)
    assert np.all(forest.slope[forest.svar < 0] == 0.0)
    # the running sum is the sum of the stored trees evaluated WITH their linear parts ...
    rules = np.zeros(3, np.int32)
    pred = predict_numpy(forest, np.arange(10)[None, :], X)[0, 0]
    np.testing.assert_allclose(pred, mu, rtol=0, atol=1e-9)
    # ... and the prediction entry point agrees with the host restatement, incl. excluded / missing
    ps = PosteriorSampler(forest, np.arange(10, dtype=np.int32)[None, :], 10, 1, backend=oracle)
    np.testing.assert_allclose(ps.sample_posterior(X, [0])[0, 0], pred, atol=1e-12)
    Xn = X[:50].copy()
    Xn[::3, 0] = np.nan
    np.testing.assert_allclose(ps.sample_posterior(Xn, [0], excluded=[1])[0, 0],
                               predict_numpy(forest, np.arange(10)[None, :], Xn, excluded=[1])[0, 0],
                               atol=1e-12)
    assert s.counters.saturations == 0


@pytest.mark.filterwarnings("ignore:response=")
def test_linear_response_through_the_step_method_and_its_limits(oracle):
    rng = np.random.default_rng(5)
    X = rng.normal(size=(200, 2))
    Y = 1.5 * X[:, 0] + rng.normal(0, 0.2, 200)
    op = BARTOp(X, Y, m=5, response="linear")
    res = sample_chain(op, tune=40, draws=10, random_seed=2, backend=oracle)
    base, batches = res["history"]
    assert any((ta.svar >= 0).any() for ta in [base] + batches)
    ps = PosteriorSampler.from_history(batches, base, 5, 1, backend=oracle)
    np.testing.assert_allclose(ps.sample_posterior(X, list(range(10)))[:, 0, :], res["mu"], rtol=0, atol=1e-9)
    assert res["vi_counts"].sum(axis=0)[0] > res["vi_counts"].sum(axis=0)[1]
    # categorical rules next to linear leaves: a leaf regresses on whatever column its parent split on, as
    # upstream's fast_linear_fit does on X[idx, selected_predictor] (a one-hot left child has no spread: constant)
    Xc = X.copy()
    Xc[:, 0] = rng.integers(0, 5, 200)
    Yc = 0.5 * X[:, 1] + 1.0 * Xc[:, 0] + rng.normal(0, 0.2, 200)
    opc = BARTOp(Xc, Yc, m=5, response="linear", split_rules=["OneHotSplit", "ContinuousSplit"])
    resc = sample_chain(opc, tune=40, draws=10, random_seed=2, backend=oracle)
    basec, batchesc = resc["history"]
    rules_c = np.array([_abi.RULE_ONEHOT, _abi.RULE_CONTINUOUS], np.int32)
    psc = PosteriorSampler.from_history(batchesc, basec, 5, 1, backend=oracle)
    np.testing.assert_allclose(psc.sample_posterior(Xc, list(range(10)))[:, 0, :], resc["mu"], rtol=0, atol=1e-9)
    last = psc.pool
    onehot_leaf = (last.svar == 0)
    assert onehot_leaf.any()                           # leaves below a one-hot split keep that regressor
    # K-vector leaves carry one slope per output on the shared regressor (the reference's
    # test_shape[linear-response] case: shape=(2, n) observed through Normal(w[0], |w[1]|))
    st = PyBartSettings.from_data(X, Y, m=6, num_particles=8, family="normal_meanscale", n_outputs=2,
                                  response="linear")
    s2 = PySampler(st, X, Y, np.zeros(2, np.int32), np.ones(2), backend=oracle)
    s2.set_likelihood([])
    for _ in range(40):
        w, _ = s2.step(True)
    assert w.shape == (2, 200) and np.isfinite(w).all()
    forest = s2.export_trees(1)
    assert forest.slope.shape == (forest.total_nodes, 2)
    lin_leaves = forest.svar >= 0
    assert lin_leaves.any() and (forest.slope[lin_leaves, 0] != 0).any() and (forest.slope[lin_leaves, 1] != 0).any()
    assert np.all(forest.slope[~lin_leaves] == 0.0)
    pred = predict_numpy(forest, np.arange(6)[None, :], X)[0]
    np.testing.assert_allclose(pred, w, rtol=0, atol=1e-9)          # sum_trees == sum of per-tree predictions
    ps2 = PosteriorSampler(forest, np.arange(6, dtype=np.int32)[None, :], 6, 2, backend=oracle)
    np.testing.assert_allclose(ps2.sample_posterior(X, [0])[0], pred, atol=1e-12)
    assert s2.counters.saturations == 0
    # every single-output family takes them
    st = PyBartSettings.from_data(X, (Y > 0).astype(float), m=4, num_particles=6, family="bernoulli_probit",
                                  response="linear")
    s = PySampler(st, X, (Y > 0).astype(float), np.zeros(2, np.int32), np.ones(2), backend=oracle)
    s.set_likelihood([])
    for _ in range(30):
        s.step(True)
    assert (s.export_trees(1).svar >= 0).any()


def test_count_likelihoods_poisson_and_negative_binomial(oracle):
    """The count models of the PyMC-BART documentation: BART on log(counts), counts observed by a
    Poisson / NegativeBinomial likelihood with a log link."""
    from scipy.stats import nbinom, poisson

    from pymc_bart_amd import NegativeBinomialLikelihood, PoissonLikelihood

    # the per-row values are log-pmfs up to a term that does not depend on mu
    f = oracle.lib.lib.pgbo_loglik
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    f.restype = None
    y = np.array([0.0, 1.0, 3.0, 10.0, 50.0] * 2)
    mu = np.concatenate([np.full(5, 0.7), np.full(5, 2.1)])
    out = np.empty(10)
    f(_abi.FAMILIES["poisson_log"], y.ctypes.data, mu.ctypes.data, 10, out.ctypes.data)
    np.testing.assert_allclose(out[:5] - out[5:], poisson.logpmf(y[:5], np.exp(0.7)) - poisson.logpmf(y[:5], np.exp(2.1)),
                               rtol=0, atol=1e-10)
    assert np.all(out <= 0.0)

    rng = np.random.default_rng(9)
    X = rng.normal(size=(800, 3))
    lograte = 0.9 * X[:, 0] + 0.8
    for lik, counts, point in (
        (PoissonLikelihood(), rng.poisson(np.exp(lograte)).astype(float), {}),
        (NegativeBinomialLikelihood("alpha"), rng.negative_binomial(3.0, 3.0 / (3.0 + np.exp(lograte))).astype(float),
         {"alpha": 3.0}),
    ):
        op = BARTOp(X, np.log(counts + 0.5), m=20)
        step = PGBART([op], num_particles=10, likelihood=lik, observed=counts, random_seed=4, backend=oracle)
        vi = np.zeros(3, np.int64)
        for it in range(120):
            if it == 80:
                step.stop_tuning()
            mu_hat, stats = step.astep(None, point)
            if it >= 80:
                vi += np.array(_decode_vi(stats[0]["variable_inclusion"], 3))
        assert np.corrcoef(mu_hat, lograte)[0, 1] > 0.85
        assert vi[0] > vi[1] + vi[2]
        assert step.counters["saturations"] == 0
    # log-lik differences of the negative binomial follow scipy's as well (alpha enters as the parameter)
    st = PyBartSettings.from_data(X, np.log(counts + 0.5), m=2, num_particles=4, family="negbin_log")
    s = PySampler(st, X, counts, np.zeros(3, np.int32), np.ones(3), backend=oracle)
    with pytest.raises(_abi.PGBError, match="alpha"):
        s.set_likelihood([])
    s.set_likelihood([3.0])
    assert nbinom.logpmf(2, 3.0, 0.5) < 0  # (scipy available: the family is checked end to end above)


def test_quantile_and_robust_regression_families(oracle):
    """AsymmetricLaplace(b, q) makes BART the q-quantile of y (the quantile-regression example of the
    PyMC-BART documentation); StudentT(nu, sigma) is the outlier-robust regression."""
    from pymc_bart_amd import AsymmetricLaplaceLikelihood, StudentTLikelihood

    rng = np.random.default_rng(17)
    X = rng.uniform(-2, 2, size=(1500, 2))
    f = np.sin(2 * X[:, 0])
    Y = f + rng.normal(0, 0.2 + 0.3 * (X[:, 0] > 0), 1500)
    cover = {}
    for q in (0.1, 0.9):
        step = PGBART([BARTOp(X, Y, m=20)], num_particles=10, likelihood=AsymmetricLaplaceLikelihood(q=q, b="b"),
                      random_seed=2, backend=oracle)
        for it in range(150):
            mu, _ = step.astep(None, {"b": 0.2})
        cover[q] = float(np.mean(Y <= mu))
        assert step.counters["saturations"] == 0
    assert 0.03 < cover[0.1] < 0.2 and 0.8 < cover[0.9] < 0.97   # BART tracks the requested quantile
    # heavy-tailed noise with gross outliers: the Student-t fit stays near f
    Yo = f + 0.1 * rng.standard_t(2, 1500)
    Yo[::50] += 25.0
    step = PGBART([BARTOp(X, np.clip(Yo, -3, 3), m=20)], num_particles=10, observed=Yo,
                  likelihood=StudentTLikelihood(nu=3.0, sigma="s"), random_seed=3, backend=oracle)
    for it in range(150):
        mu, _ = step.astep(None, {"s": 0.15})
    assert np.sqrt(np.mean((mu - f) ** 2)) < 0.35
    st = PyBartSettings.from_data(X, Y, m=2, num_particles=4, family="asymmetric_laplace")
    s = PySampler(st, X, Y, np.zeros(2, np.int32), np.ones(2), backend=oracle)
    with pytest.raises(_abi.PGBError, match="0 < q < 1"):
        s.set_likelihood([0.2, 1.5])


def test_offset_of_the_linear_predictor_for_count_models(oracle):
    """A log-exposure (or any other additive term of the model) enters the per-row families as an
    offset of the linear predictor: BART then fits the rate, not rate x exposure."""
    from pymc_bart_amd import PoissonLikelihood

    rng = np.random.default_rng(23)
    X = rng.normal(size=(1500, 2))
    expo = rng.uniform(0.2, 6.0, 1500)
    lograte = 0.8 * X[:, 0] + 0.3
    counts = rng.poisson(expo * np.exp(lograte)).astype(float)
    op = BARTOp(X, np.log((counts + 0.5) / expo), m=20)
    step = PGBART([op], num_particles=10, likelihood=PoissonLikelihood(), observed=counts, random_seed=6,
                  backend=oracle)
    for it in range(150):
        mu, _ = step.astep(None, {}, offset=np.log(expo))
    assert np.sqrt(np.mean((mu - lograte) ** 2)) < 0.3
    assert abs(float(np.mean(mu - lograte))) < 0.1            # no exposure left in the fit
    # the Normal family has no predictor offset in the ABI (the caller subtracts from the response)
    st = PyBartSettings.from_data(X, lograte, m=2, num_particles=4)
    s = PySampler(st, X, lograte, np.zeros(2, np.int32), np.ones(2), backend=oracle)
    with pytest.raises(_abi.PGBError, match="per-row families"):
        s.set_offset(np.zeros(1500))


def test_out_of_sample_fit_is_competitive_with_gradient_boosting(oracle):
    """A reference-free quality pin: on Friedman's function (BASELINE.json configs[0] shape) the
    posterior-mean prediction on held-out rows is at least as good as a tuned-by-default gradient
    boosting regressor, up to 15 %."""
    from sklearn.ensemble import GradientBoostingRegressor

    rng = np.random.default_rng(3415)
    X = rng.uniform(0, 1, (700, 5))
    f = 10 * np.sin(np.pi * X[:, 0] * X[:, 1]) + 20 * (X[:, 2] - 0.5) ** 2 + 10 * X[:, 3] + 5 * X[:, 4]
    Y = f + rng.normal(0, 1, 700)
    tr, te = slice(0, 500), slice(500, 700)
    op = BARTOp(X[tr], Y[tr], m=50)
    sample_chain(op, tune=300, draws=200, random_seed=1, backend=oracle)
    pred = _sample_posterior(_get_posterior_sampler(op, backend=oracle), X[te], np.random.default_rng(0), size=150)
    bart_rmse = float(np.sqrt(np.mean((pred.mean(axis=0)[:, 0] - f[te]) ** 2)))
    gbm = GradientBoostingRegressor(random_state=0).fit(X[tr], Y[tr])
    gbm_rmse = float(np.sqrt(np.mean((gbm.predict(X[te]) - f[te]) ** 2)))
    assert bart_rmse < 1.15 * gbm_rmse, (bart_rmse, gbm_rmse)
    assert bart_rmse < 1.6   # in absolute terms: noise sd is 1, f has sd ~4.9


def test_partial_dependence_ice_and_inclusion_helpers(oracle):
    # the numbers behind the reference's plot_pdp / plot_ice (utils.py:168-487), its
    # get_variable_inclusion (utils.py:747-806; use: tests/test_bart.py:205-208) and vi_to_kulprit
    # (utils.py:1093-1108; use: tests/test_utils.py:84-86)
    from pymc_bart_amd import (compute_variable_importance, get_variable_inclusion,
                               individual_conditional_expectation, partial_dependence, vi_to_kulprit)
    from pymc_bart_amd.partial import pdp_grid

    rng = np.random.default_rng(8)
    X = rng.uniform(-1, 1, size=(300, 3))
    Y = 2.0 * X[:, 0] + (X[:, 2] > 0) + rng.normal(0, 0.1, 300)
    op = BARTOp(X, Y, m=20)
    res = sample_chain(op, tune=150, draws=60, random_seed=11, backend=oracle)

    # grids: quantiles (default), linear, insample
    assert pdp_grid(X).shape == (9, 3) and pdp_grid(X, "linear").shape == (10, 3)
    np.testing.assert_allclose(pdp_grid(X, "linear", 5)[[0, -1]], [X.min(0), X.max(0)])
    np.testing.assert_allclose(pdp_grid(X, "quantiles", [0.5])[0], np.median(X, axis=0))
    assert pdp_grid(X, "insample") is not None and pdp_grid(X, "insample").shape == X.shape
    with pytest.raises(ValueError):
        pdp_grid(X, "nope")

    pd_ = partial_dependence(op, X, xs_interval="linear", xs_values=9, samples=40, random_seed=3, backend=oracle)
    assert set(pd_["pd"]) == {0, 1, 2} and pd_["pd"][0].shape == (40, 9, 1) and pd_["labels"][1] == "X_1"
    m0, m1 = pd_["pd"][0].mean(axis=0)[:, 0], pd_["pd"][1].mean(axis=0)[:, 0]
    slope = np.polyfit(pd_["x"][0], m0, 1)[0]
    assert 1.5 < slope < 2.3                        # the linear effect of x0 is recovered ...
    assert np.ptp(m1) < 0.25 * np.ptp(m0)           # ... x1 has none
    step = pd_["pd"][2].mean(axis=0)[:, 0]
    assert 0.7 < step[-1] - step[0] < 1.3           # the unit jump along x2
    assert abs(pd_["reference"] - Y.mean()) < 0.3
    # same generator, same call pattern -> same numbers; var_idx restricts the sweep
    again = partial_dependence(op, X, var_idx=[0], xs_interval="linear", xs_values=9, samples=40, random_seed=3,
                               backend=oracle)
    assert np.array_equal(again["pd"][0], pd_["pd"][0]) and set(again["pd"]) == {0}
    doubled = partial_dependence(op, X, var_idx=[0], xs_interval="linear", xs_values=9, samples=40, random_seed=3,
                                 func=lambda a: 2 * a, backend=oracle)
    assert np.array_equal(doubled["pd"][0], 2 * pd_["pd"][0])

    ice = individual_conditional_expectation(op, X, var_idx=[0, 1], instances=6, samples=15, random_seed=4,
                                             backend=oracle)
    assert ice["ice"][0].shape == (6, 300, 1) and len(ice["instances"]) == 6
    assert np.all(ice["ice"][0][:, 0, 0] == 0.0)    # centred on the first row
    order = np.argsort(ice["x"][0])
    rise = ice["ice"][0][:, order[-20:], 0].mean() - ice["ice"][0][:, order[:20], 0].mean()
    assert 2.5 < rise < 4.5                          # ~ 2 * (0.9 - (-0.9))
    raw = individual_conditional_expectation(op, X, var_idx=[0], instances=6, samples=15, centered=False,
                                             random_seed=4, backend=oracle)
    np.testing.assert_allclose(raw["ice"][0] - raw["ice"][0][:, :1, :], ice["ice"][0], atol=1e-12)

    share, labels = get_variable_inclusion(res["variable_inclusion"], X)
    assert share.shape == (3,) and abs(share.sum() - 1) < 1e-12 and np.all(np.diff(share) <= 0)
    assert labels[0] in ("0", "2") and labels[-1] == "1" and all(isinstance(s, str) for s in labels)
    path = get_variable_inclusion(res["variable_inclusion"], X, to_kulprit=True)
    assert path[0] == [] and path[-1] == labels and len(path) == 4
    vi = compute_variable_importance(res["variable_inclusion"], op, X, samples=10, random_seed=1, backend=oracle)
    terms = vi_to_kulprit(vi)
    assert len(terms) == 3 and terms[0] == [] and all("+" not in t for ts in terms[1:] for t in ts)
