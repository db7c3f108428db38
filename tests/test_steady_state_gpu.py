"""Parity where the bench measures: chains that have left their stumps behind, at BASELINE.json's full sizes.

The full-size cases of tests/test_parity_gpu.py compare the first tree updates of a chain: every one of them
replaces a stump.  Here (round-5 VERDICT, "next" #1):
  (a) full n and p with FEW trees, so that every tree is revisited several times inside the oracle's budget;
  (b) the chain image (include/pgbart_image.h): the GPU runs the bench's burn-in -- 2 000 tree updates while
      tuning: grown trees, tuned split weights and leaf_sd -- writes its image, the CPU oracle loads it, and both
      continue: bit for bit, with m = 200 / 100 trees, at the state bench.py times.
Also here: the HIP library writes, byte for byte, the image the oracle writes at the same point of a chain; chains
migrate between the backends and between the 64- and 128-particle builds without a trace.
"""
import json
import os

import numpy as np
import pytest

from _cases import digest, make_case, run_case
from pymc_bart_amd import workloads
from pymc_bart_amd.image import ChainImage, differing_fields
from pymc_bart_amd.sampler import PyBartSettings, PySampler

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_runs.json")))

MIGRATE = ["cfg1_friedman", "nan_onehot_prior", "ragged_1025", "tiny_n3", "one_tree_two_particles", "max_particles",
           "duplicates", "deep_trees", "onehot_fail_nan", "probit_cfg4_small", "logit_nan_onehot",
           "categorical_k3_reference", "categorical_k4_cfg5_small", "meanscale_k2_reference", "subset_rule",
           "categorical_k6_generic", "categorical_k12", "categorical_k16_linear", "linear_response", "mix_response",
           "negbin_counts", "quantile_asymlaplace", "poisson_exposure", "linear_poisson", "meanscale_k2_linear",
           "categorical_k3_mix", "categorical_k3_offset", "linear_mixed_rules", "categorical_k3_linear_mixed_rules",
           "stump_first_categorical", "upstream/nan_onehot_prior", "upstream/categorical_k3_reference",
           "particles_128", "particles_100_probit", "categorical_k4_particles_100"]


@pytest.mark.parametrize("name", MIGRATE)
def test_a_chain_migrates_between_the_backends_without_a_trace(hip, oracle, name):
    """GPU -> oracle -> GPU -> oracle through pgb_checkpoint_save / _load, cuts while tuning, at the boundary and in
    the draws: the committed fingerprint of the chain that never moved."""
    c = make_case(name)
    h = c["steps"] // 2
    cuts = {1: oracle, h - 1: hip, h: oracle, h + 2: hip, c["steps"] - 1: oracle}
    assert digest(run_case(c, hip, checkpoint_at=cuts)) == GOLD[name]


@pytest.mark.parametrize("name", ["nan_onehot_prior", "probit_cfg4_small", "categorical_k4_cfg5_small", "categorical_k12",
                                  "linear_response", "categorical_k3_mix", "subset_rule", "onehot_fail_nan",
                                  "upstream/logit_nan_onehot", "particles_128"])
def test_the_hip_image_is_the_oracle_image(hip, oracle, name):
    """Same chain, same point: the two backends write the same record -- every section, the row labels of every
    tree included -- except for who wrote it and the backend-specific slot counter."""
    c = make_case(name)
    g = run_case(c, hip)["sampler"]
    o = run_case(c, oracle)["sampler"]
    ig, io = ChainImage.parse(g.checkpoint()), ChainImage.parse(o.checkpoint())
    assert ig.writer == "hip-gfx950" and io.writer == "oracle-cpu"
    assert differing_fields(ig, io) == []


@pytest.mark.parametrize("name", ["nan_onehot_prior", "probit_cfg4_small", "categorical_k12", "linear_response"])
def test_the_two_builds_of_the_library_exchange_images(hip, name):
    from pymc_bart_amd import _abi
    from pymc_bart_amd.sampler import Backend

    big = Backend(lib=_abi.load_hip_library(128), mem=hip.mem)
    c = make_case(name)
    h = c["steps"] // 2
    assert digest(run_case(c, hip, checkpoint_at={2: big, h: hip, h + 3: big})) == GOLD[name]


def test_a_poisoned_handle_is_restored_by_an_image_of_another_backend(hip, oracle):
    """include/pgbart.h: a callback error abandons the astep half-way and poisons the handle until an idle image is
    loaded.  The image may come from anywhere: here the oracle wrote it."""
    from pymc_bart_amd import _abi

    rng = np.random.default_rng(4)
    X = rng.normal(size=(1500, 3))
    Y = (rng.random(1500) < 1 / (1 + np.exp(-1.5 * X[:, 0]))).astype(float)
    st = PyBartSettings.from_data(X, Y, m=5, num_particles=8, seed=9, family="callback")

    def ll(y, mu):
        return y * mu - np.logaddexp(0.0, mu)

    def mk(be, fn):
        s = PySampler(st, X, Y, np.zeros(3, np.int32), np.ones(3), backend=be)
        s.set_loglik_callback(fn)
        s.set_likelihood([])
        return s

    o = mk(oracle, ll)
    for it in range(5):
        o.step(it < 3)
    blob = o.checkpoint()
    calls = {"n": 0}

    def flaky(y, mu):
        calls["n"] += 1
        if calls["n"] == 3:
            raise RuntimeError("boom")
        return ll(y, mu)

    g = mk(hip, flaky)
    with pytest.raises(_abi.PGBError, match="boom"):
        for _ in range(4):
            g.step(True)
    with pytest.raises(_abi.PGBError, match="abandoned"):
        g.step(True)
    with pytest.raises(_abi.PGBError, match="abandoned"):
        g.checkpoint()
    g.restore(blob)
    a, va = g.step(False)
    b, vb = o.step(False)
    assert np.array_equal(a, b) and np.array_equal(va, vb)


# ---------------------------------------------------------------- (a) full n and p, few trees: every tree is revisited
def _same_step(g, o, tune):
    a, va = g.step(tune)
    b, vb = o.step(tune)
    assert np.array_equal(a, b) and np.array_equal(va, vb)
    ta, tb = g.export_trees(0), o.export_trees(0)
    for f in ("tree_id", "node_off", "var", "left", "right", "count", "split", "value", "rule"):
        assert np.array_equal(getattr(ta, f), getattr(tb, f)), f


def _same_chain_state(g, o):
    cg, co = g.counters.as_dict(), o.counters.as_dict()
    for k in ("particle_steps", "tree_updates", "rows_touched", "rounds", "saturations", "partitions"):
        assert cg[k] == co[k], (k, cg[k], co[k])
    assert cg["saturations"] == 0
    sg, so = g.state(), o.state()
    assert np.array_equal(sg["leaf_sd"], so["leaf_sd"]) and sg["iter"] == so["iter"] and sg["lower"] == so["lower"]
    assert np.array_equal(g.split_weights(), o.split_weights())


def _pair(w, hip, oracle, **kw):
    X, Y = w["X"], w["Y"]
    p = X.shape[1]
    st = PyBartSettings.from_data(X, Y, seed=3415, **kw)
    g = PySampler(st, X, Y, np.zeros(p, np.int32), np.ones(p), backend=hip)
    o = PySampler(st, X, Y, np.zeros(p, np.int32), np.ones(p), backend=oracle)
    for s in (g, o):
        s.set_likelihood([1.0] if kw.get("family", "normal") == "normal" else [])
    return g, o


def test_full_size_cfg2_shape_few_trees_revisited(hip, oracle):
    """n = 100k, p = 50, P = 40, m = 8, every tree every step: 2 tuning + 3 draw steps = each tree updated 5 times
    (grown reference particles, FINAL / INIT on grown old trees, split weights and leaf_sd tuned in between)."""
    g, o = _pair(workloads.cfg2(seed=3415), hip, oracle, m=8, num_particles=40, batch=(1.0, 1.0))
    for it in range(5):
        _same_step(g, o, it < 2)
    _same_chain_state(g, o)
    assert ChainImage.parse(g.checkpoint()).node_off[-1] > 3 * 8  # the trees did grow
    assert differing_fields(ChainImage.parse(g.checkpoint()), ChainImage.parse(o.checkpoint())) == []


def test_full_size_cfg4_shape_few_trees_revisited(hip, oracle):
    """n = 1M, p = 100, Bernoulli-probit, P = 40, m = 4 (16-bit order keys by size, not forced): 4 steps of 4 trees."""
    g, o = _pair(workloads.cfg4(seed=3415), hip, oracle, m=4, num_particles=40, batch=(1.0, 1.0), family="bernoulli_probit")
    for it in range(4):
        _same_step(g, o, it < 2)
    _same_chain_state(g, o)
    assert differing_fields(ChainImage.parse(g.checkpoint()), ChainImage.parse(o.checkpoint())) == []


def test_full_size_cfg5_shape_few_trees_revisited(hip, oracle):
    """n = 250k, p = 200, K = 4 softmax, P = 40, m = 4: 4 steps of 4 trees."""
    g, o = _pair(workloads.cfg5(seed=3415), hip, oracle, m=4, num_particles=40, batch=(1.0, 1.0), family="categorical",
                 n_outputs=4)
    for it in range(4):
        _same_step(g, o, it < 2)
    _same_chain_state(g, o)
    assert differing_fields(ChainImage.parse(g.checkpoint()), ChainImage.parse(o.checkpoint())) == []


# ---------------------------------------------------------------- (b) the oracle resumes the GPU chain after the bench's burn-in
def _burn_in_then_compare(g, o, tune_steps, compare):
    """g: `tune_steps` asteps while tuning on the GPU alone; its image goes to the oracle; then `compare` =
    [(tune, n_steps), ...] on both."""
    g.step_async(True, tune_steps)
    g.sync()
    blob = g.checkpoint()
    img = ChainImage.parse(blob)
    o.restore(blob)
    assert differing_fields(ChainImage.parse(o.checkpoint()), img) == []  # the oracle holds the same chain now
    for tune, k in compare:
        for _ in range(k):
            _same_step(g, o, tune)
    _same_chain_state(g, o)
    end_g, end_o = ChainImage.parse(g.checkpoint()), ChainImage.parse(o.checkpoint())
    assert differing_fields(end_g, end_o) == []
    return img


def test_full_size_cfg2_parity_at_steady_state(hip, oracle):
    """BASELINE.json configs[1] exactly as bench.py runs it (m = 200, P = 40, 20 trees per astep): 100 tuning asteps
    on the GPU, then the oracle takes the image and both run one more tuning astep and one draw astep -- 40 tree
    updates on trees that have been re-sampled ten times."""
    w = workloads.cfg2(seed=3415)
    g, o = _pair(w, hip, oracle, m=200, num_particles=40)
    img = _burn_in_then_compare(g, o, 100, [(True, 1), (False, 1)])
    assert img.header.iter == 2000 and img.header.rs_count == 2000
    assert img.node_off[-1] > 3 * 200 and img.alpha.max() > img.alpha.min()  # grown trees, tuned split weights
    assert not np.array_equal(img.leaf_sd, [g.settings.init_leaf_sd])


def test_full_size_cfg4_parity_at_steady_state(hip, oracle):
    """BASELINE.json configs[3]: n = 1M, p = 100, probit, m = 200, P = 40; 8 trees per astep so that the oracle's
    share (8 tree updates at ~2.5 s each) stays bounded: 250 tuning asteps (2 000 tree updates) on the GPU, then
    one tuning and one draw astep on both."""
    w = workloads.cfg4(seed=3415)
    g, o = _pair(w, hip, oracle, m=200, num_particles=40, batch=(8, 8), family="bernoulli_probit")
    img = _burn_in_then_compare(g, o, 250, [(True, 1), (False, 1)])  # (tune=0 is what bench.py times)
    assert img.header.iter == 2000 and img.node_off[-1] > 3 * 200


def test_full_size_cfg5_parity_at_steady_state(hip, oracle):
    """BASELINE.json configs[4]: K = 4 softmax, n = 250k, p = 200, m = 100, P = 40; 8 trees per astep (a sweep
    is 12 asteps of 8 trees and one of 4): 130 tuning asteps = ten sweeps = 1 000 tree updates on the GPU, then one
    tuning and one draw astep on both."""
    w = workloads.cfg5(seed=3415)
    g, o = _pair(w, hip, oracle, m=100, num_particles=40, batch=(8, 8), family="categorical", n_outputs=4)
    img = _burn_in_then_compare(g, o, 130, [(True, 1), (False, 1)])  # (tune=0 is what bench.py times)
    assert img.header.iter == 1000 and img.node_off[-1] > 3 * 100
