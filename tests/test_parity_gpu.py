"""Parity of the gfx950 HIP backend with the CPU oracle -- through the C ABI, bit for bit.

Integer decisions (split variables, rows, resampling indices, counts) AND floating-point outputs
(sum_trees, leaf values, leaf_sd) must be identical: the numeric contract (include/pgbart_spec.h)
makes every result independent of execution order, so the tolerance is 0.
"""
import json
import os

import numpy as np
import pytest

from _cases import CASES, digest, make_case, random_case, run_case
from pymc_bart_amd import workloads
from pymc_bart_amd.chains import sample_chain
from pymc_bart_amd.pgbart import BARTOp
from pymc_bart_amd.sampler import PyBartSettings, PySampler
from pymc_bart_amd.trees import PosteriorSampler, predict_numpy
from pymc_bart_amd.utils import _get_posterior_sampler, _sample_posterior

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_runs.json")))


def _assert_same(a, b):
    assert np.array_equal(a["sum_trees"], b["sum_trees"])
    assert np.array_equal(a["vi"], b["vi"])
    assert len(a["trees"]) == len(b["trees"])
    for x, y in zip(a["trees"], b["trees"]):
        assert np.array_equal(x, y)
    assert a["counters"] == b["counters"]
    assert np.array_equal(a["split_weights"], b["split_weights"])
    assert np.array_equal(a["state"]["leaf_sd"], b["state"]["leaf_sd"])
    assert a["state"]["iter"] == b["state"]["iter"] and a["state"]["lower"] == b["state"]["lower"]


@pytest.mark.parametrize("name", CASES)
def test_hip_equals_oracle_and_golden(hip, oracle, name):
    c = make_case(name)
    g = run_case(c, hip)
    o = run_case(c, oracle)
    _assert_same(g, o)
    assert digest(g) == GOLD[name]          # committed fixture
    assert g["counters"]["saturations"] == 0


@pytest.mark.parametrize("name", ["cfg1_friedman", "nan_onehot_prior", "max_particles", "one_tree_two_particles", "probit_cfg4_small",
                                  "categorical_k3_reference", "categorical_k12", "linear_response", "subset_rule"])
def test_two_particles_per_lane_build_gives_the_same_chain(hip, name):
    """`libpgbart_hip_p128.so` (the same source with two particles per lane of the control kernel) run on cases of
    <= 64 particles reproduces the committed fingerprints of the default build: the build parameter is a capacity,
    not a numeric choice (`pgb_weights_scan` adds an all-zero second block)."""
    from pymc_bart_amd import _abi
    from pymc_bart_amd.sampler import Backend

    big = Backend(lib=_abi.load_hip_library(128), mem=hip.mem)
    assert big.lib.max_particles == 128 and hip.lib.max_particles == 64
    c = make_case(name)
    assert c["P"] <= 64
    assert digest(run_case(c, big)) == GOLD[name]


@pytest.mark.parametrize("name", ["cfg1_friedman", "nan_onehot_prior", "duplicates", "onehot_fail_nan",
                                  "probit_cfg4_small", "logit_nan_onehot", "categorical_k3_reference",
                                  "categorical_k4_cfg5_small", "meanscale_k2_reference"])
def test_float32_shadow_of_the_split_columns_changes_nothing(hip, monkeypatch, name):
    """A design matrix larger than the Infinity Cache is partitioned on 16-bit order keys of its columns
    (k_rows / k_rows_mk <F32>: key(x) = number of the column's equi-depth boundaries <= x decides, equal keys
    fetch the float64 value; round 3: a float32 shadow).  Forced on at test sizes (PGB_X32_MIN_MB=0, read when
    the data are set): missing values, one-hot columns, heavy ties (many rows per key), every row-pass instance
    that has the variant -- the committed fingerprints still hold."""
    monkeypatch.setenv("PGB_X32_MIN_MB", "0")
    g = run_case(make_case(name), hip)
    assert digest(g) == GOLD[name]
    assert g["counters"]["saturations"] == 0


def test_hip_is_deterministic_across_runs(hip):
    c = make_case("nan_onehot_prior")
    assert digest(run_case(c, hip)) == digest(run_case(c, hip))


def test_step_async_equals_stepwise(hip):
    c = make_case("cfg1_friedman")
    X, Y = c["X"], c["Y"]
    st = PyBartSettings.from_data(X, Y, m=c["m"], num_particles=c["P"], seed=7)
    mk = lambda: PySampler(st, X, Y, np.zeros(5, np.int32), np.ones(5), backend=hip)  # noqa: E731
    a, b = mk(), mk()
    for s in (a, b):
        s.set_likelihood([1.0])
    for _ in range(6):
        sa, _ = a.step(True)
    b.step_async(True, 5)
    sb, _ = b.step(True)
    assert np.array_equal(sa, sb)
    ca, cb = a.sync(), b.sync()
    for k in ("particle_steps", "tree_updates", "rows_touched", "rounds"):
        assert ca[k] == cb[k]


@pytest.mark.parametrize("name", ["nan_onehot_prior", "categorical_k4_cfg5_small", "mix_response"])
def test_host_output_step_equals_the_device_output_step(hip, name):
    """pgb_step_host (PGBART.astep's return path: mapped block + one DMA) against pgb_step (device
    buffer + per-item copies): same sum_trees, same vi, same exported trees, same counters."""
    c = make_case(name)
    X, Y = c["X"], c["Y"]
    p = X.shape[1]
    fam = c.get("family", "normal")
    st = PyBartSettings.from_data(X, c.get("bart_Y", Y), m=c["m"], num_particles=c["P"], seed=11, family=fam,
                                  n_outputs=c.get("K", 1), response=c.get("response", "constant"))
    rules = np.zeros(p, np.int32) if c["rules"] is None else c["rules"]
    prior = np.ones(p) if c["prior"] is None else c["prior"]
    a = PySampler(st, X, Y, rules, prior, backend=hip)
    b = PySampler(st, X, Y, rules, prior, backend=hip)
    K, n = st.n_outputs, st.n
    for it in range(14):
        for s in (a, b):
            s.set_likelihood([0.7] if fam == "normal" else c.get("lik_params", []))
        sa, va = a.step(it < 7)                    # host path
        _, vb = b.step(it < 7, fetch=False)        # device path
        sb = hip.mem.to_host(b.sum_trees_device())
        sb = sb.reshape(K, n) if K > 1 else sb
        assert np.array_equal(sa, sb) and np.array_equal(va, vb)
        ta, tb = a.export_trees(0), b.export_trees(0)   # a: served from the mapped block; b: fetched
        for f in ("tree_id", "node_off", "var", "left", "right", "count", "split", "value", "slope", "xbar", "svar"):
            assert np.array_equal(getattr(ta, f), getattr(tb, f)), f
        assert a.counters.as_dict() == b.counters.as_dict()


def test_step_async_returns_before_the_work_is_done_and_errors_surface_in_sync(hip):
    """pgb_step_async hands the steps to the handle's worker thread and returns; pgb_sync waits."""
    import time

    w = workloads.cfg2(seed=1, n=50_000, p=10, m=50, num_particles=20)
    st = PyBartSettings.from_data(w["X"], w["Y"], m=50, num_particles=20, seed=3)
    s = PySampler(st, w["X"], w["Y"], np.zeros(10, np.int32), np.ones(10), backend=hip)
    s.set_likelihood([1.0])
    s.step_async(False, 2)
    s.sync()
    t0 = time.perf_counter()
    s.step_async(False, 200)
    t_call = time.perf_counter() - t0
    c = s.sync()
    t_all = time.perf_counter() - t0
    assert c["tree_updates"] == 202 * 5
    assert t_call < 0.25 * t_all, (t_call, t_all)
    # a call that needs the handle while a job runs waits for it (no interleaving)
    s.step_async(False, 20)
    st1, _ = s.step(False)
    assert s.counters.tree_updates == 223 * 5 and st1.shape == (50_000,)


def test_predict_kernel_matches_oracle_and_numpy(hip, oracle):
    rng = np.random.default_rng(5)
    X = rng.normal(size=(700, 4))
    X[:, 3] = rng.integers(0, 3, 700)
    Y = 3 * X[:, 0] + (X[:, 3] == 1) + rng.normal(0, 0.1, 700)
    rules = ["ContinuousSplit"] * 3 + ["OneHotSplit"]
    res_g = sample_chain(BARTOp(X, Y, m=8, split_rules=rules), 20, 10, random_seed=1, backend=hip)
    res_o = sample_chain(BARTOp(X, Y, m=8, split_rules=rules), 20, 10, random_seed=1, backend=oracle)
    assert np.array_equal(res_g["mu"], res_o["mu"])
    rid = np.array([0, 0, 0, 1], np.int32)
    base, batches = res_g["history"]
    pg = PosteriorSampler.from_history(batches, base, 8, 1, backend=hip)
    po = PosteriorSampler.from_history(batches, base, 8, 1, backend=oracle)
    Xn = rng.normal(size=(333, 4))
    Xn[:, 3] = rng.integers(0, 3, 333)
    Xn[::7, 0] = np.nan
    for excl in (None, [0], [1, 3], [0, 1, 2, 3]):
        a = pg.sample_posterior(Xn, [0, 3, 9, 9], excl)
        b = po.sample_posterior(Xn, [0, 3, 9, 9], excl)
        assert a.shape == (4, 1, 333)
        assert np.array_equal(a, b)
    ref = predict_numpy(pg.pool, pg.forest_idx[[2]], Xn[:40], excluded=[1])
    np.testing.assert_allclose(pg.sample_posterior(Xn[:40], [2], [1]), ref, atol=1e-12)
    # every stored draw evaluated on the training X is the sampled sum_trees
    np.testing.assert_allclose(pg.sample_posterior(X, list(range(10)))[:, 0, :], res_g["mu"], atol=1e-9)


def test_sample_posterior_row_subset_consistency_gpu(hip):
    # reference tests/test_utils.py:24-32 on the HIP path
    rng0 = np.random.default_rng(3415)
    X = np.hstack([rng0.normal(0, 1, size=(50, 2)), rng0.binomial(1, 0.5, size=(50, 1))])
    Y = rng0.normal(0, 1, size=50)
    op = BARTOp(X, Y, m=10)
    sample_chain(op, tune=40, draws=40, random_seed=3415, backend=hip)
    sampler = _get_posterior_sampler(op, backend=hip)
    pred_all = _sample_posterior(sampler, X=X, rng=np.random.default_rng(3), size=2)
    pred_first = _sample_posterior(sampler, X=X[:10], rng=np.random.default_rng(3))
    np.testing.assert_almost_equal(pred_first, pred_all[0, :10], decimal=4)
    assert pred_all.shape == (2, 50, 1) and pred_first.shape == (10, 1)


def test_full_size_cfg2_parity_and_invariants(hip, oracle):
    """BASELINE.json configs[1] at full size: n=100k, p=50, m=200, 40 particles."""
    w = workloads.cfg2(seed=3415)
    X, Y = w["X"], w["Y"]
    n, p = X.shape
    st = PyBartSettings.from_data(X, Y, m=200, num_particles=40, seed=3415)
    g = PySampler(st, X, Y, np.zeros(p, np.int32), np.ones(p), backend=hip)
    o = PySampler(st, X, Y, np.zeros(p, np.int32), np.ones(p), backend=oracle)
    for s in (g, o):
        s.set_likelihood([1.0])
    # (1) exact agreement with the oracle on a bounded number of steps (oracle: ~1 s per step)
    for it in range(4):
        a, va = g.step(tune=it < 2)
        b, vb = o.step(tune=it < 2)
        assert np.array_equal(a, b) and np.array_equal(va, vb)
    cg, co = g.counters.as_dict(), o.counters.as_dict()
    for k in ("particle_steps", "tree_updates", "rows_touched", "rounds", "saturations"):
        assert cg[k] == co[k]
    # (2) size-independent properties after a longer GPU-only run
    g.step_async(False, 10)
    st_dev, _ = g.step(False)
    forest = g.export_trees(1)
    assert forest.n_trees == 200
    roots = forest.node_off[:-1]
    assert np.all(forest.count[roots] == n)            # every tree still owns every row
    inner = np.flatnonzero(forest.var >= 0)
    base = np.repeat(forest.node_off[:-1], np.diff(forest.node_off))
    assert np.all(forest.count[base[inner] + forest.left[inner]] + forest.count[base[inner] + forest.right[inner]]
                  == forest.count[inner])              # children partition their parent (no NaN here)
    ps = PosteriorSampler(forest, np.arange(200, dtype=np.int32)[None, :], 200, 1, backend=hip)
    pred = ps.sample_posterior(X, [0])[0, 0]
    np.testing.assert_allclose(pred, st_dev, rtol=0, atol=1e-8)  # sum_trees == sum of its trees
    assert g.counters.saturations == 0
    rmse = float(np.sqrt(np.mean((st_dev - w["f"]) ** 2)))
    assert rmse < np.std(w["f"])                        # it is actually learning the signal


def test_full_size_cfg4_probit_parity_and_invariants(hip, oracle):
    """BASELINE.json configs[3] at full size: Bernoulli-probit, n=1M, p=100, m=200, 40 particles.
    The oracle needs seconds per tree at this size, so exact agreement is checked on 2 steps of
    2 trees; size-independent properties on a longer GPU-only run."""
    w = workloads.cfg4(seed=3415)
    X, Y = w["X"], w["Y"]
    n, p = X.shape
    st = PyBartSettings.from_data(X, Y, m=200, num_particles=40, seed=3415, family="bernoulli_probit",
                                  batch=(2, 2))
    g = PySampler(st, X, Y, np.zeros(p, np.int32), np.ones(p), backend=hip)
    o = PySampler(st, X, Y, np.zeros(p, np.int32), np.ones(p), backend=oracle)
    for s in (g, o):
        s.set_likelihood([])
    for it in range(2):
        a, va = g.step(tune=it < 1)
        b, vb = o.step(tune=it < 1)
        assert np.array_equal(a, b) and np.array_equal(va, vb)
    cg, co = g.counters.as_dict(), o.counters.as_dict()
    for k in ("particle_steps", "tree_updates", "rows_touched", "rounds", "saturations"):
        assert cg[k] == co[k]
    del o
    g.step_async(False, 20)
    st_dev, _ = g.step(False)
    forest = g.export_trees(1)
    roots = forest.node_off[:-1]
    assert np.all(forest.count[roots] == n)
    inner = np.flatnonzero(forest.var >= 0)
    base = np.repeat(forest.node_off[:-1], np.diff(forest.node_off))
    assert np.all(forest.count[base[inner] + forest.left[inner]] + forest.count[base[inner] + forest.right[inner]]
                  == forest.count[inner])
    sub = np.arange(0, n, 97)
    ps = PosteriorSampler(forest, np.arange(200, dtype=np.int32)[None, :], 200, 1, backend=hip)
    pred = ps.sample_posterior(X[sub], [0])[0, 0]
    np.testing.assert_allclose(pred, st_dev[sub], rtol=0, atol=1e-8)
    assert g.counters.saturations == 0
    assert np.corrcoef(st_dev, w["f"])[0, 1] > 0.3  # the latent signal is being picked up


def test_full_size_cfg5_categorical_parity_and_invariants(hip, oracle):
    """BASELINE.json configs[4] at full size: shape=(4, n) BART, n=250k, p=200, m=100, softmax."""
    w = workloads.cfg5(seed=3415)
    X, Y = w["X"], w["Y"]
    n, p = X.shape
    st = PyBartSettings.from_data(X, Y, m=100, num_particles=40, seed=3415, family="categorical",
                                  n_outputs=4, batch=(1, 1))
    g = PySampler(st, X, Y, np.zeros(p, np.int32), np.ones(p), backend=hip)
    o = PySampler(st, X, Y, np.zeros(p, np.int32), np.ones(p), backend=oracle)
    for s in (g, o):
        s.set_likelihood([])
    for it in range(2):
        a, va = g.step(tune=it < 1)
        b, vb = o.step(tune=it < 1)
        assert a.shape == (4, n)
        assert np.array_equal(a, b) and np.array_equal(va, vb)
    cg, co = g.counters.as_dict(), o.counters.as_dict()
    for k in ("particle_steps", "tree_updates", "rows_touched", "rounds", "saturations"):
        assert cg[k] == co[k]
    assert np.array_equal(g.state()["leaf_sd"], o.state()["leaf_sd"])
    del o
    g.step_async(False, 30)
    st_dev, _ = g.step(False)
    forest = g.export_trees(1)
    assert forest.n_outputs == 4 and forest.value.shape[1] == 4
    sub = np.arange(0, n, 53)
    ps = PosteriorSampler(forest, np.arange(100, dtype=np.int32)[None, :], 100, 4, backend=hip)
    pred = ps.sample_posterior(X[sub], [0])[0]
    np.testing.assert_allclose(pred, st_dev[:, sub], rtol=0, atol=1e-8)
    assert g.counters.saturations == 0


def test_concurrent_chains_on_one_gpu_equal_the_chains_run_alone(hip):
    """pm.sample(chains=4) on one GPU: every chain on its own HIP stream and host thread.  Chains
    share nothing, so each must reproduce, bit for bit, what it draws when it runs alone."""
    from pymc_bart_amd.chains import sample_chains

    w = workloads.cfg2(seed=5, n=20_000, p=10, m=20, num_particles=10)
    kw = dict(num_particles=10, random_seed=11, backend=hip)
    together = sample_chains(BARTOp(w["X"], w["Y"], m=20), chains=4, tune=3, draws=3, **kw)
    assert [r["chain"] for r in together] == [0, 1, 2, 3]
    for c in (0, 3):
        alone = sample_chain(BARTOp(w["X"], w["Y"], m=20), 3, 3, chain=c, **kw)
        assert np.array_equal(alone["mu"], together[c]["mu"])
        assert np.array_equal(alone["sigma"], together[c]["sigma"])
        assert alone["variable_inclusion"] == together[c]["variable_inclusion"]
    assert not np.array_equal(together[0]["mu"], together[1]["mu"])


@pytest.mark.parametrize("name", ["nan_onehot_prior", "ragged_1025", "max_particles",
                                  "categorical_k4_cfg5_small", "probit_cfg4_small", "linear_response",
                                  "categorical_k3_mix"])
def test_checkpoint_resume_does_not_change_the_chain_gpu(hip, name):
    """A chain resumed from pgb_checkpoint_load on a fresh handle reproduces the committed
    fingerprint of the uninterrupted chain (cuts in tuning, at the boundary and in the draws)."""
    c = make_case(name)
    cuts = (1, c["steps"] // 2 - 1, c["steps"] // 2 + 2)
    assert digest(run_case(c, hip, checkpoint_at=cuts)) == GOLD[name]


def test_pgbart_pickle_round_trip_on_gpu(hip):
    import pickle

    from pymc_bart_amd.pgbart import PGBART, NormalLikelihood

    w = workloads.cfg2(seed=8, n=30_000, p=8, m=20, num_particles=12)

    def run(pickle_at):
        step = PGBART([BARTOp(w["X"], w["Y"], m=20)], num_particles=12,
                      likelihood=NormalLikelihood("sigma"), random_seed=3)
        out = []
        for it in range(8):
            if it == 4:
                step.stop_tuning()
            if it in pickle_at:
                step = pickle.loads(pickle.dumps(step))
            out.append(step.astep(None, {"sigma": 1.1})[0])
        return np.array(out)

    assert np.array_equal(run(()), run((2, 5)))


def test_variable_importance_on_gpu_matches_the_oracle(hip, oracle):
    """the O(p) / O(p^2) prediction sweeps with excluded covariates run in
    k_predict; the chain, its history and every sweep agree with the CPU oracle."""
    from pymc_bart_amd.importance import compute_variable_importance

    rng = np.random.default_rng(21)
    X = rng.normal(size=(2000, 6))
    Y = 3.0 * X[:, 1] + 1.5 * np.sin(2 * X[:, 3]) + rng.normal(0, 0.2, 2000)
    outs = []
    for be in (hip, oracle):
        op = BARTOp(X, Y, m=20)
        res = sample_chain(op, tune=40, draws=20, random_seed=5, backend=be)
        for method, kw in (("VI", {}), ("backward", {}), ("backward_VI", {"fixed": 2})):
            outs.append(compute_variable_importance(res["variable_inclusion"], op, X, method=method,
                                                    samples=8, random_seed=1, backend=be, **kw))
    for g, o in zip(outs[:3], outs[3:]):
        assert np.array_equal(g["indices"], o["indices"])
        np.testing.assert_allclose(g["preds_all"], o["preds_all"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(g["r2_mean"], o["r2_mean"], rtol=0, atol=1e-9)
    assert set(outs[0]["indices"][:2]) == {1, 3}


def test_full_size_cfg2_recovers_the_regression_function(hip):
    """Not parity but purpose: at the headline size (n=100k, p=50, m=200, P=40) a short run must
    fit the synthetic regression function.  f has standard deviation ~4.9, the noise is N(0, 1);
    ten sweeps over the trees bring the posterior mean inside the noise level and most of the
    sampler's splits onto the five informative columns."""
    w = workloads.cfg2(seed=3415)
    X, Y, f = w["X"], w["Y"], w["f"]
    st = PyBartSettings.from_data(X, Y, m=w["m"], num_particles=w["num_particles"], seed=11)
    s = PySampler(st, X, Y, np.zeros(50, np.int32), np.ones(50), backend=hip)
    s.set_likelihood([1.0])
    s.step_async(True, 100)   # 10 sweeps while tuning
    s.sync()
    draws, vi = [], np.zeros(50, np.int64)
    for _ in range(30):
        mu, v = s.step(False)
        draws.append(mu)
        vi += v
    post = np.mean(draws, axis=0)
    rmse = float(np.sqrt(np.mean((post - f) ** 2)))
    assert rmse < float(np.std(Y - f)), rmse                         # inside the noise level (measured: 0.71)
    assert rmse < 0.2 * float(np.std(f))                             # and far from a constant fit
    assert vi[:5].sum() > 0.5 * vi.sum()                             # 5 of 50 columns draw most splits (0.68)
    assert s.counters.saturations == 0


def test_fuzz_parity_over_random_configurations(hip, oracle):
    """120 random configurations (sizes around the chunk / wave boundaries, every family and split
    rule, NaNs, ties, priors, batch sizes, tree priors): the two backends agree bit for bit on
    every one.  (The same generator ran about 50 000 configurations on MI355X in this round: 0 mismatches.)"""
    for seed in range(5000, 5120):
        c = random_case(seed)
        g, o = digest(run_case(c, hip)), digest(run_case(c, oracle))
        assert g == o, (seed, c["family"], c["X"].shape, c["m"], c["P"], c["K"], c["rules"].tolist())


def test_fuzz_parity_under_the_upstream_semantics_switches(hip, oracle):
    """The same generator with pgb_settings.compat = 1, 2, 3 in turn (fresh particles at log-weight 0 until they grow;
    empty right leaves of one-hot splits; both): bit for bit."""
    for seed in range(7000, 7075):
        c = random_case(seed, compat=1 + seed % 3)
        g, o = digest(run_case(c, hip)), digest(run_case(c, oracle))
        assert g == o, (seed, c["compat"], c["family"], c["X"].shape, c["m"], c["P"], c["K"], c["rules"].tolist())


@pytest.mark.filterwarnings("ignore:response=")
def test_linear_leaves_predict_the_same_on_gpu_and_host(hip, oracle):
    """response="linear": the chain, its exported slopes and the prediction kernel (with excluded and
    missing regressors) agree with the oracle / the host restatement."""
    rng = np.random.default_rng(12)
    X = rng.uniform(-2, 2, size=(5000, 4))
    X[rng.random(5000) < 0.05, 2] = np.nan
    Y = np.where(X[:, 0] < 0, 2 * X[:, 0], -X[:, 0]) + 0.3 * X[:, 1] + rng.normal(0, 0.1, 5000)
    res = {}
    for name, be in (("hip", hip), ("oracle", oracle)):
        op = BARTOp(X, Y, m=12, response="linear")
        r = sample_chain(op, tune=20, draws=6, num_particles=12, random_seed=4, sigma=0.2, backend=be)
        s = _get_posterior_sampler(op, backend=be)
        Xn = X[:300].copy()
        Xn[::5, 0] = np.nan
        res[name] = (r["mu"], _sample_posterior(s, Xn, np.random.default_rng(1), size=4, excluded=[1]), r["history"])
    assert np.array_equal(res["hip"][0], res["oracle"][0])
    np.testing.assert_allclose(res["hip"][1], res["oracle"][1], rtol=0, atol=1e-12)
    base, batches = res["hip"][2]
    assert any((ta.svar >= 0).any() for ta in [base] + batches)
    for a, b in zip([base] + batches, [res["oracle"][2][0]] + res["oracle"][2][1]):
        assert np.array_equal(a.slope, b.slope) and np.array_equal(a.xbar, b.xbar) and np.array_equal(a.svar, b.svar)


def test_k_vector_linear_leaves_on_gpu(hip, oracle):
    """The reference's test_shape[linear-response] model at a larger size: K = 2 leaves with one slope
    per output on the shared regressor.  Chain, exported slopes and the prediction kernel (excluded
    and missing regressors included) agree with the oracle / the host restatement."""
    rng = np.random.default_rng(21)
    n, m = 6000, 10
    X = rng.uniform(-2, 2, size=(n, 3))
    X[rng.random(n) < 0.05, 1] = np.nan
    Y = np.where(X[:, 0] < 0, 1.5 * X[:, 0], -X[:, 0]) + rng.normal(0, 1, n) * (0.3 + 0.3 * (X[:, 2] > 0))
    rules = np.zeros(3, np.int32)
    out = {}
    for name, be in (("hip", hip), ("oracle", oracle)):
        st = PyBartSettings.from_data(X, Y, m=m, num_particles=12, family="normal_meanscale", n_outputs=2,
                                      response="linear", seed=9)
        s = PySampler(st, X, Y, rules, np.ones(3), backend=be)
        s.set_likelihood([])
        for it in range(30):
            w, _ = s.step(it < 20)
        forest = s.export_trees(1)
        ps = PosteriorSampler(forest, np.arange(m, dtype=np.int32)[None, :], m, 2, backend=be)
        Xn = X[:400].copy()
        Xn[::4, 0] = np.nan
        out[name] = (w, forest, ps.sample_posterior(X[:400], [0]), ps.sample_posterior(Xn, [0], excluded=[2]))
        assert s.counters.saturations == 0
    (wh, fh, ph, pxh), (wo, fo, po, pxo) = out["hip"], out["oracle"]
    assert np.array_equal(wh, wo)
    for f in ("var", "split", "left", "right", "count", "value", "slope", "xbar", "svar"):
        assert np.array_equal(getattr(fh, f), getattr(fo, f)), f
    lin = fh.svar >= 0
    assert fh.slope.shape == (fh.total_nodes, 2) and (fh.slope[lin, 0] != 0).any() and (fh.slope[lin, 1] != 0).any()
    np.testing.assert_allclose(ph, po, rtol=0, atol=1e-12)
    np.testing.assert_allclose(pxh, pxo, rtol=0, atol=1e-12)
    full = ~np.isnan(X[:400]).any(axis=1)  # (training rows with a missing split value leave their tree)
    np.testing.assert_allclose(ph[0][:, full], wh[:, :400][:, full], rtol=0, atol=1e-9)  # sum_trees == predictions
    host = predict_numpy(fh, np.arange(m)[None, :], Xn[:60], excluded=[2])
    np.testing.assert_allclose(pxh[0][:, :60], host[0], rtol=0, atol=1e-12)


@pytest.mark.parametrize("name", ["nan_onehot_prior", "probit_cfg4_small", "categorical_k3_mix", "linear_response"])
def test_results_do_not_depend_on_launch_geometry(hip, name, monkeypatch):
    """All row reductions are integer sums, so the draws cannot depend on how the rows are cut into
    work items or how many workgroups run them: odd grids and item targets (read at pgb_create)
    reproduce the committed fingerprint."""
    c = make_case(name)
    for env in ({"PGB_ROWS_GRID": "7", "PGB_ROWS_TARGET": "3", "PGB_ROWS_TARGET_INIT": "5", "PGB_LL_GRID": "5",
                 "PGB_LL_TARGET": "2"},
                {"PGB_ROWS_GRID": "333", "PGB_ROWS_TARGET": "100000", "PGB_ROWS_TARGET_INIT": "1",
                 "PGB_LL_GRID": "1000", "PGB_LL_TARGET": "99999"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        assert digest(run_case(c, hip)) == GOLD[name], env


def test_host_callback_likelihood_on_gpu_reproduces_the_builtin_family(hip):
    """Family "callback" (pgb_set_loglik_callback): the device hands the host every re-labelled row's
    linear predictor, the callback evaluates them, the sums go back.  With the check loss spelt out in
    exact IEEE arithmetic it reproduces the built-in asymmetric-Laplace fingerprint bit for bit, and
    the step method built on it runs through checkpoints."""
    from test_callback_family import callback_case, run_callback_case

    g = run_callback_case(callback_case(), hip)
    assert digest(g) == GOLD["quantile_asymlaplace"]
    assert g["counters"]["saturations"] == 0


def test_host_callback_with_offset_matches_the_oracle(hip, oracle):
    from test_callback_family import run_callback_case

    c = dict(make_case("poisson_exposure"))
    c["family"] = "callback"
    c["callback"] = lambda y, mu: y * mu - 2.0 * mu * mu       # any elementwise function of exact arithmetic
    _assert_same(run_callback_case(c, hip), run_callback_case(c, oracle))


def test_partial_dependence_sweep_on_gpu_matches_the_oracle(hip, oracle):
    """The PDP / ICE sweeps (all-but-one covariate excluded, k_predict) give the oracle's numbers."""
    from pymc_bart_amd import individual_conditional_expectation, partial_dependence

    rng = np.random.default_rng(31)
    X = rng.uniform(-1, 1, size=(2000, 4))
    X[rng.random(2000) < 0.05, 3] = np.nan
    Y = 2.0 * X[:, 0] - X[:, 1] ** 2 + rng.normal(0, 0.1, 2000)
    got = {}
    for name, be in (("hip", hip), ("oracle", oracle)):
        op = BARTOp(X, Y, m=15)
        sample_chain(op, tune=20, draws=10, num_particles=10, random_seed=6, sigma=0.2, backend=be)
        Xc = np.nan_to_num(X)
        pd_ = partial_dependence(op, Xc, xs_interval="linear", xs_values=12, samples=20, random_seed=2, backend=be)
        ice = individual_conditional_expectation(op, Xc[:200], var_idx=[0, 1], instances=4, samples=8,
                                                 random_seed=2, backend=be)
        got[name] = (pd_, ice)
    for j in range(4):
        np.testing.assert_allclose(got["hip"][0]["pd"][j], got["oracle"][0]["pd"][j], rtol=0, atol=1e-12)
    for j in (0, 1):
        np.testing.assert_allclose(got["hip"][1]["ice"][j], got["oracle"][1]["ice"][j], rtol=0, atol=1e-12)
    assert got["hip"][0]["reference"] == pytest.approx(got["oracle"][0]["reference"], abs=1e-12)


def test_prediction_kernel_paths_agree_with_the_oracle(hip, oracle):
    """k_predict has a fixed-length walk (rows without missing values, trees without excluded
    variables) and a general walk, rows staged in LDS (p <= 126) or read from HBM (wider): every
    combination gives the oracle's sums."""
    rng = np.random.default_rng(17)
    for p, rules_kind in ((5, "cont"), (150, "cont"), (6, "mixed")):
        n = 1500
        X = rng.normal(size=(n, p))
        rules = np.zeros(p, np.int32)
        if rules_kind == "mixed":
            X[:, 1] = rng.integers(0, 4, n)
            rules[1] = 1
        Y = X[:, 0] - 2 * (X[:, p - 1] > 0) + rng.normal(0, 0.3, n)
        st = PyBartSettings.from_data(X, Y, m=12, num_particles=8, seed=5)
        out = {}
        for name, be in (("hip", hip), ("oracle", oracle)):
            s = PySampler(st, X, Y, rules, np.ones(p), backend=be)
            s.set_likelihood([0.5])
            for it in range(12):
                s.step(it < 8)
            forest = s.export_trees(1)
            ps = PosteriorSampler(forest, np.arange(12, dtype=np.int32)[None, :], 12, 1, backend=be)
            Xn = X[:700].copy()
            Xn[100:140:3, 0] = np.nan                      # some waves carry missing values, others do not
            out[name] = (ps.sample_posterior(X, [0]), ps.sample_posterior(Xn, [0]),
                         ps.sample_posterior(Xn, [0], excluded=[0]), ps.sample_posterior(X[:65], [0], excluded=[p - 1]))
        for a, b in zip(out["hip"], out["oracle"]):
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)


def test_prediction_rejects_a_matrix_narrower_than_the_trees(hip):
    """A tree that splits on column j cannot be evaluated on a matrix with fewer than j + 1 columns:
    the library says so instead of reading past the rows."""
    from pymc_bart_amd import _abi

    rng = np.random.default_rng(3)
    X = rng.normal(size=(400, 4))
    Y = 3.0 * X[:, 3] + rng.normal(0, 0.1, 400)
    op = BARTOp(X, Y, m=5)
    sample_chain(op, tune=10, draws=3, num_particles=8, random_seed=1, sigma=0.3, backend=hip)
    s = _get_posterior_sampler(op, backend=hip)
    assert _sample_posterior(s, X, np.random.default_rng(0), size=2).shape == (2, 400, 1)
    with pytest.raises(_abi.PGBError, match="column"):
        _sample_posterior(s, X[:, :2], np.random.default_rng(0), size=2)


def test_normal_chain_is_equivariant_under_rescaling_of_the_response_on_gpu(hip):
    """A property of the HIP backend by itself (no oracle involved), at a size with many chunks per pass."""
    from test_oracle_behaviour import _scale_equivariance

    _scale_equivariance(hip, n=40_000)


@pytest.mark.parametrize("x32", ["default", "float32 shadow forced"])
def test_chain_is_invariant_under_monotone_transforms_of_the_covariates_on_gpu(hip, monkeypatch, x32):
    from test_oracle_behaviour import _monotone_invariance

    if x32 != "default":
        monkeypatch.setenv("PGB_X32_MIN_MB", "0")
    _monotone_invariance(hip, n=30_000)


def test_thirty_million_rows_stay_consistent(hip):
    """Size edge (include/pgbart.h: n < 2^31 - 1024; memory sized for 288 GB): n = 30 M rows -- 29 297 chunks per
    pass, label rings of 15 GB, element offsets beyond 2^32 -- with the properties that need no oracle: every row
    is in exactly one leaf of every tree, sum_trees is the sum of the stored trees' predictions (checked on a
    sample of rows through the prediction kernel), the chain is finite and has moved."""
    rng = np.random.default_rng(30)
    n, p, m = 30_000_000, 4, 5
    X = rng.standard_normal((n, p))
    X[::1000, 2] = np.nan
    Y = np.sin(X[:, 0]) + 0.5 * (X[:, 1] > 0) + 0.1 * rng.standard_normal(n)
    st = PyBartSettings.from_data(X, Y, m=m, num_particles=10, seed=7, batch=(1.0, 1.0))
    s = PySampler(st, X, Y, np.zeros(p, np.int32), np.ones(p), backend=hip)
    s.set_likelihood([0.1])
    for _ in range(3):
        mu, _ = s.step(True)
    assert mu.shape == (n,) and np.all(np.isfinite(mu)) and mu.std() > 0.05
    forest = s.export_trees(1)
    off = np.asarray(forest.node_off)
    count, var, left, right = (np.asarray(getattr(forest, f)) for f in ("count", "var", "left", "right"))
    for k in range(m):
        assert count[off[k]] == n                                            # the root holds every row
        nodes = np.arange(off[k], off[k + 1])
        inner = nodes[var[nodes] >= 0]
        kids = count[off[k] + left[inner]] + count[off[k] + right[inner]]
        assert np.all(kids <= count[inner]) and np.all(count[inner] - kids <= np.isnan(X[:, 2]).sum())   # only NaN rows drop out
    rows = rng.integers(0, n, 50_000)
    rows[:3] = [0, n - 1, 2 ** 24 + 1]
    ps = PosteriorSampler(forest, np.arange(m, dtype=np.int32)[None, :], m, 1, backend=hip)
    pred = ps.sample_posterior(X[rows], [0])[0, 0]
    ok = ~np.isnan(X[rows, 2])                                                # (a dropped row is predicted by its parent's mixture)
    np.testing.assert_allclose(pred[ok], mu[rows][ok], rtol=0, atol=1e-9)
    assert s.counters.saturations == 0


def test_subset_codes_are_checked_by_the_library_on_gpu(hip):
    from test_oracle_behaviour import _subset_codes_are_checked_by_the_library

    _subset_codes_are_checked_by_the_library(hip)


def test_packed_tree_record_on_gpu_equals_the_array_export(hip, oracle):
    """``pgb_export_trees_packed`` on the HIP backend: the record served from the mapped block of the last
    ``pgb_step_host`` and the one fetched from the device (after ``pgb_step``) both equal the array export,
    and equal the oracle's record byte for byte."""
    import ctypes as C

    from pymc_bart_amd import _abi
    from pymc_bart_amd.trees import TreeArrays

    c = make_case("cfg1_friedman")
    st = PyBartSettings.from_data(c["X"], c["Y"], m=c["m"], num_particles=c["P"], seed=c["seed"])
    rules, prior = np.zeros(c["X"].shape[1], np.int32), np.ones(c["X"].shape[1])
    g = PySampler(st, c["X"], c["Y"], rules, prior, backend=hip)
    o = PySampler(st, c["X"], c["Y"], rules, prior, backend=oracle)
    for s_ in (g, o):
        s_.set_likelihood([1.0])

    def arrays(s_, which):
        lib = s_.backend.lib
        cc = _abi.TreeArraysC()
        lib.check(lib.lib.pgb_export_trees(s_._h, which, C.byref(cc)), "size")
        ta = TreeArrays.empty(cc.n_trees, cc.total_nodes, cc.n_outputs)
        c2 = ta.as_c()
        lib.check(lib.lib.pgb_export_trees(s_._h, which, C.byref(c2)), "fill")
        return ta

    for it in range(6):
        fetch = it % 2 == 0           # host-output step (mapped block) / device-output step (fetched)
        g.step(it < 3, fetch=fetch)
        o.step(it < 3, fetch=fetch)
        for which in (0, 1):
            pg, po, ag = g.export_trees(which), o.export_trees(which), arrays(g, which)
            assert pg.raw == po.raw
            for f in ("tree_id", "node_off", "var", "split", "left", "right", "count", "value"):
                assert np.array_equal(getattr(pg, f), getattr(ag, f)), (it, which, f)


def test_failed_callback_poisons_the_gpu_handle_until_a_checkpoint_is_loaded(hip):
    """include/pgbart.h: a log-likelihood callback that fails abandons the astep half-way; every later step /
    export on that handle is refused (PGB_E_STATE) until an idle image is restored, after which the chain is the
    one that never saw the failure."""
    from pymc_bart_amd import _abi

    rng = np.random.default_rng(3)
    X = rng.normal(size=(3000, 3))
    Y = X[:, 0] + rng.normal(0, 0.3, 3000)
    st = PyBartSettings.from_data(X, Y, m=5, num_particles=6, seed=2, family="callback")
    good = lambda y, mu: -0.5 * ((y - mu) / 0.3) ** 2  # noqa: E731
    s = PySampler(st, X, Y, np.zeros(3, np.int32), np.ones(3), backend=hip)
    t = PySampler(st, X, Y, np.zeros(3, np.int32), np.ones(3), backend=hip)
    for q in (s, t):
        q.set_loglik_callback(good)
        q.step(True)
    image = s.checkpoint()

    def bad(y, mu):
        raise FloatingPointError("logp overflowed")

    s.set_loglik_callback(bad)
    with pytest.raises(_abi.PGBError, match="logp overflowed"):
        s.step(True)
    s.set_loglik_callback(good)
    for call in (lambda: s.step(True), lambda: s.step_async(True, 1), lambda: s.export_trees(0)):
        with pytest.raises(_abi.PGBError, match="abandoned half-way"):
            call()
    s.restore(image)
    for _ in range(3):
        a, _ = s.step(True)
        b, _ = t.step(True)
        assert np.array_equal(a, b)


def test_pageable_and_pinned_output_buffers_receive_the_same_sum_trees(hip):
    """`pgb_step_host` writes sum_trees straight into device-accessible pinned memory (what PySampler hands in) and
    goes through a densify + DMA for a pageable buffer; with and without a second output stream.  Same chain, same
    bits, whichever way the results leave the device."""
    import ctypes as C

    c = make_case("cfg1_friedman")
    st = PyBartSettings.from_data(c["X"], c["Y"], m=c["m"], num_particles=c["P"], seed=c["seed"])
    rules, prior = np.zeros(c["X"].shape[1], np.int32), np.ones(c["X"].shape[1])
    a = PySampler(st, c["X"], c["Y"], rules, prior, backend=hip)      # pinned buffers, shared output stream
    b = PySampler(st, c["X"], c["Y"], rules, prior, backend=hip)      # pageable buffer, results on the sampler's stream
    lib = hip.lib
    lib.check(lib.lib.pgb_set_output_stream(b._h, None), "pgb_set_output_stream")
    for s_ in (a, b):
        s_.set_likelihood([1.0])
    n = c["X"].shape[0]
    vi = np.zeros(c["X"].shape[1], np.int32)
    for it in range(8):
        ra, via = a.step(it < 4)
        rb = np.empty(n)                                               # ordinary (pageable) memory
        lib.check(lib.lib.pgb_step_host(b._h, int(it < 4), rb.ctypes.data, vi.ctypes.data, C.byref(b.counters)),
                  "pgb_step_host")
        assert np.array_equal(ra, rb) and np.array_equal(via, vi)
        assert a.export_trees(0).raw == b.export_trees(0).raw


def test_soak_of_every_stepping_api_in_random_alternation(hip, oracle):
    """`tools/soak_parity.py` for a few seconds inside the suite: host-output asteps, device-output steps, asynchronous
    batches, tune flips, sigma changes, checkpoint -> fresh sampler -> restore and exports in random alternation on
    one chain, HIP and oracle compared after every call (the round's run of it: 49 174 asteps, 1 820 restores, 0
    mismatches -- profiles/r03_experiments.md)."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import soak_parity

    calls, steps, kinds = soak_parity.run(6.0, 11, backs={"hip": hip, "oracle": oracle})
    assert calls > 200 and steps > 200 and len(kinds) >= 6


def test_offsets_and_responses_the_likelihood_tables_cannot_address_are_refused_on_gpu(hip):
    """k_nonfinite with its bound (PGB_MAX_OFFSET): the HIP library refuses what the oracle refuses."""
    from test_host_logic import _refusals

    _refusals(hip)


def test_softmax_slow_children_take_the_unfactorised_form_on_both_backends(hip):
    """The factorised softmax of constant leaves (pgb_loglik_cat_f) sends a child whose leaf values lie more than
    PGB_CAT_DMAX = 300 apart down the unfactorised form -- unreachable with sane leaf values.  TEST builds of both
    backends with PGB_CAT_DMAX = 0.001 (``__graft_entry__.build``: build/variants/) make nearly every child a slow
    one: the kernel's deferred fallback (rows marked in the passes, evaluated after them) must reproduce the oracle's
    bit for bit, for K = 3, 4 and the run-time-K instance, with and without missing values and offsets."""
    import __graft_entry__ as g
    from _oracle import NumpyMemory
    from pymc_bart_amd import _abi
    from pymc_bart_amd.sampler import Backend

    if not (os.path.exists(g.HIP_SO_CATSLOW) and os.path.exists(g.ORACLE_SO_CATSLOW)):
        pytest.skip("the PGB_CAT_DMAX test builds are missing (python -c 'import __graft_entry__ as g; g.build()')")
    hip_slow = Backend(lib=_abi.PGBLibrary(g.HIP_SO_CATSLOW), mem=hip.mem)
    orc_slow = Backend(lib=_abi.PGBLibrary(g.ORACLE_SO_CATSLOW), mem=NumpyMemory())
    assert hip_slow.lib.backend_name == "hip-gfx950" and orc_slow.lib.backend_name == "oracle-cpu"
    for name in ("categorical_k3_reference", "categorical_k4_cfg5_small", "categorical_k6_generic", "categorical_k12",
                 "categorical_k3_offset"):
        c = make_case(name)
        a, b = digest(run_case(c, hip_slow)), digest(run_case(c, orc_slow))
        assert a == b, name
        assert a["counters"]["particle_steps"] > 0
