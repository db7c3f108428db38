"""Host-side rules of the step method that do not need a GPU (round-1 ADVICE items): batch sizes as
upstream computes them, fixed-point range sizing and saturation reporting, history publication,
malformed histories rejected by the predictor, offsets surviving a pickle round trip."""
import pickle

import numpy as np
import pytest

from pymc_bart_amd import _abi
from pymc_bart_amd.pgbart import PGBART, BARTOp, NormalLikelihood
from pymc_bart_amd.sampler import PyBartSettings, PySampler
from pymc_bart_amd.trees import PosteriorSampler


def _data(n=300, p=3, seed=5):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, p))
    return X, X[:, 0] - X[:, 1] + rng.normal(0, 0.2, n)


def test_batch_fractions_follow_upstream_and_ints_are_counts():
    """[U] batch = (max(1, int(m*b)), ...): batch=(1.0, 1.0) re-samples EVERY tree each step."""
    X, Y = _data()
    st = PyBartSettings.from_data(X, Y, m=20, batch=(1.0, 1.0))
    assert st.batch_sizes() == (20, 20)
    assert PyBartSettings.from_data(X, Y, m=20, batch=(0.1, 0.26)).batch_sizes() == (2, 5)
    assert PyBartSettings.from_data(X, Y, m=5, batch=(0.1, 0.1)).batch_sizes() == (1, 1)
    assert PyBartSettings.from_data(X, Y, m=20, batch=(3, 2)).batch_sizes() == (3, 2)   # Python ints: counts
    assert PyBartSettings.from_data(X, Y, m=20, batch=(1, 1)).batch_sizes() == (1, 1)
    assert PyBartSettings.from_data(X, Y, m=20, batch=(50, 1.0)).batch_sizes() == (20, 20)


def test_all_trees_per_step_when_batch_is_one(oracle):
    X, Y = _data()
    st = PyBartSettings.from_data(X, Y, m=20, num_particles=5, batch=(1.0, 1.0), seed=1)
    s = PySampler(st, X, Y, np.zeros(3, np.int32), np.ones(3), backend=oracle)
    s.set_likelihood([1.0])
    s.step(True)
    assert s.counters.tree_updates == 20


def test_range_covers_the_observed_response_and_saturation_is_reported_once(oracle):
    X, Y = _data()
    # the likelihood's response is far outside Y's own range: the range must follow it
    st = PyBartSettings.from_data(X, Y, m=10, y_obs=Y + 500.0)
    assert st.range_exp >= 12 and PyBartSettings.from_data(X, Y, m=10).range_exp < 8
    assert PyBartSettings.from_data(X, Y, m=10, range_exp=20).range_exp == 20
    # an offset that leaves the range: that step raises (and names itself), later steps do not
    step = PGBART([BARTOp(X, Y, m=10)], num_particles=5, likelihood=NormalLikelihood(1.0), random_seed=3,
                  backend=oracle)
    step.astep(None)
    with pytest.raises(_abi.PGBError, match="by astep 2"):
        step.astep(None, offset=np.full(X.shape[0], 1.0e4))
    step.astep(None, offset=np.zeros(X.shape[0]))      # the chain is usable again
    step.astep(None)
    wide = PGBART([BARTOp(X, Y, m=10)], num_particles=5, likelihood=NormalLikelihood(1.0), random_seed=3,
                  backend=oracle, range_exp=18)
    wide.astep(None, offset=np.full(X.shape[0], 1.0e4))  # sized for it: no error


def test_history_is_published_once_and_grows_in_place(oracle):
    X, Y = _data()
    op = BARTOp(X, Y, m=6)
    a = PGBART([op], num_particles=4, random_seed=1, chain=0, backend=oracle)
    b = PGBART([op], num_particles=4, random_seed=1, chain=1, backend=oracle)
    for s in (a, b):
        s.stop_tuning()
    for _ in range(5):
        a.astep(None)
        b.astep(None)
    assert len(op.all_trees) == 2 and a._slot != b._slot
    assert len(op.all_trees[a._slot][1]) == 5 and op.all_trees[a._slot][1] is a._batches


def test_list_like_history_is_current_after_every_draw(oracle):
    """A history container that is neither a list nor a manager proxy: the entry is re-assigned on every
    draw, so a reader never sees fewer batches than draws (round-2 ADVICE: no flush call exists in PyMC)."""
    class Proxy:  # pickles what it is given, like multiprocessing.Manager().list()
        def __init__(self):
            self.items, self.sets = [], 0

        def append(self, x):
            self.items.append(pickle.loads(pickle.dumps(x)))

        def __len__(self):
            return len(self.items)

        def __setitem__(self, i, x):
            self.sets += 1
            self.items[i] = pickle.loads(pickle.dumps(x))

        def __getitem__(self, i):
            return self.items[i]

    X, Y = _data(n=120)
    op = BARTOp(X, Y, m=4, all_trees=Proxy())
    s = PGBART([op], num_particles=4, random_seed=1, backend=oracle)
    s.stop_tuning()
    for d in range(20):
        s.astep(None)
        assert len(op.all_trees[0][1]) == d + 1
    s.flush_history()                        # harmless, nothing left to send
    assert len(op.all_trees) == 1 and len(op.all_trees[0][1]) == 20


@pytest.mark.parametrize("start_method", ["fork", "spawn"])
def test_manager_list_history_is_complete_when_the_worker_just_exits(oracle, start_method):
    """Reference ``bart.py:134-135``: ``all_trees`` is a ``Manager().list()``; PyMC worker processes run
    their draws and exit without telling the step method (``tests/test_bart.py:84-104``: 100 draws x 2
    chains, then predictions from the parent).  Every draw must be in the parent's history."""
    import multiprocessing as mp

    from _workers import run_chain_and_exit
    from pymc_bart_amd.utils import _get_posterior_sampler, _sample_posterior

    X, Y = _data(n=150)
    with mp.Manager() as manager:
        op = BARTOp(X, Y, m=5, all_trees=manager.list())
        ctx = mp.get_context(start_method)
        procs = [ctx.Process(target=run_chain_and_exit, args=(op, c, 7, 100)) for c in range(2)]
        for pr in procs:
            pr.start()
        for pr in procs:
            pr.join(120)
        assert [pr.exitcode for pr in procs] == [0, 0]
        assert len(op.all_trees) == 2                      # one entry per chain, as utils.py:124-127 reads it
        assert [len(batches) for _, batches in op.all_trees] == [100, 100]
        op.n_outputs = 1                                   # (set by the step method in the worker's copy)
        ps = _get_posterior_sampler(op, backend=oracle)
        assert ps.n_draws == 200
        pred = _sample_posterior(ps, X[:7], np.random.default_rng(0), size=(2, 3))
        assert pred.shape == (2, 3, 7, 1) and np.all(np.isfinite(pred))
        # ... and in the process that owns the manager (sequential sampling): same mailbox, same rule
        s = PGBART([op], num_particles=4, random_seed=1, chain=2, backend=oracle)
        s.stop_tuning()
        for _ in range(3):
            s.astep(None)
        assert [len(batches) for _, batches in op.all_trees] == [100, 100, 3]


def test_predictor_rejects_malformed_histories(oracle):
    X, Y = _data()
    step = PGBART([BARTOp(X, Y, m=5)], num_particles=5, random_seed=2, backend=oracle)
    step.stop_tuning()
    for _ in range(3):
        step.astep(None)
    ps = PosteriorSampler.from_history(step._batches, step._baseline, 5, 1, backend=oracle)
    good = ps.sample_posterior(X[:4], [0, 1], [])
    assert good.shape == (2, 1, 4)
    # a forest index outside the tree list (mismatched m / truncated file)
    with pytest.raises(_abi.PGBError, match="forest_tree_idx"):
        bad = PosteriorSampler.from_history(step._batches, step._baseline, 5, 1, backend=oracle)
        bad.forest_idx = bad.forest_idx.copy()
        bad.forest_idx[0, 0] = 10_000
        bad.sample_posterior(X[:4], [0], [])
    with pytest.raises(_abi.PGBError, match="inconsistent"):
        bad = PosteriorSampler.from_history(step._batches, step._baseline, 5, 1, backend=oracle)
        k = int(np.argmax(bad.pool.var >= 0))
        assert bad.pool.var[k] >= 0
        bad.pool.left = bad.pool.left.copy()
        bad.pool.left[k] = 300
        bad.sample_posterior(X[:4], [0], [])


def test_offset_survives_pickling(oracle, monkeypatch):
    import pymc_bart_amd.sampler as sm

    monkeypatch.setattr(sm, "_DEFAULT_BACKEND", oracle)   # unpickling builds on the default backend
    X, Y = _data()
    off = np.linspace(-1, 1, X.shape[0])
    a = PGBART([BARTOp(X, Y + off, m=6)], num_particles=5, likelihood=NormalLikelihood(1.0), random_seed=9,
               backend=oracle)
    b = PGBART([BARTOp(X, Y + off, m=6)], num_particles=5, likelihood=NormalLikelihood(1.0), random_seed=9,
               backend=oracle)
    for s in (a, b):
        s.astep(None, offset=off)
    b2 = pickle.loads(pickle.dumps(b))
    for _ in range(4):                       # the caller does NOT repeat the offset
        ra, _ = a.astep(None)
        rb, _ = b2.astep(None)
        assert np.array_equal(ra, rb)


def test_gather_without_dense_draws():
    from pymc_bart_amd.chains import gather_chains

    res = {"chain": 0, "mu": None, "sigma": np.ones(3), "variable_inclusion": [], "vi_counts": np.zeros((3, 2)),
           "history": (None, []), "counters": {}, "step": object()}
    out = gather_chains(res)
    assert out[0]["mu"] is None and "step" not in out[0]


def test_multi_output_offsets(oracle):
    """Offsets of the linear predictors for K-vector leaves ([K][n]): zeros change nothing (bit for bit),
    a real offset moves the chain, and the step method checks the shape."""
    from pymc_bart_amd.pgbart import CategoricalLikelihood

    rng = np.random.default_rng(4)
    n, K = 300, 3
    X = rng.normal(size=(n, 3))
    Z = rng.normal(size=n)
    off = np.stack([1.5 * Z, -1.5 * Z, np.zeros(n)])
    F = np.stack([X[:, 0], -X[:, 0], 0 * X[:, 0]]) + off
    pr = np.exp(F) / np.exp(F).sum(0)
    Y = (rng.random(n)[None, :] > np.cumsum(pr, axis=0)).sum(0).clip(0, K - 1).astype(float)

    def run(offset, steps=30):
        st = PGBART([BARTOp(X, Y, m=6)], num_particles=6, likelihood=CategoricalLikelihood(K), random_seed=3,
                    backend=oracle)
        outs = []
        for it in range(steps):
            if it == steps // 2:
                st.stop_tuning()
            outs.append(st.astep(None, offset=offset if it == 0 else None)[0])
        return np.array(outs)

    base = run(None)
    assert np.array_equal(base, run(np.zeros((K, n))))
    with_off = run(off)
    assert not np.array_equal(base, with_off)
    # with the Z-driven part handled by the offset, the trees need less of the class-0-vs-1 contrast
    assert np.abs(with_off[-10:, 0] - with_off[-10:, 1]).mean() < np.abs(base[-10:, 0] - base[-10:, 1]).mean() + 0.5
    with pytest.raises(ValueError, match="shape"):
        PGBART([BARTOp(X, Y, m=6)], num_particles=6, likelihood=CategoricalLikelihood(K), random_seed=3,
               backend=oracle).astep(None, offset=np.zeros(n))


def _export_unpacked(s, which):
    """The two-call array export of the ABI (``pgb_export_trees``): what ``export_trees`` used before."""
    import ctypes as C

    from pymc_bart_amd.trees import TreeArrays

    lib = s.backend.lib
    c = _abi.TreeArraysC()
    lib.check(lib.lib.pgb_export_trees(s._h, which, C.byref(c)), "size")
    ta = TreeArrays.empty(c.n_trees, c.total_nodes, c.n_outputs)
    c2 = ta.as_c()
    lib.check(lib.lib.pgb_export_trees(s._h, which, C.byref(c2)), "fill")
    return ta


TREE_FIELDS = ("tree_id", "node_off", "var", "split", "left", "right", "count", "value", "slope", "xbar", "svar",
               "rule")


@pytest.mark.parametrize("response,K", [("constant", 1), ("linear", 1), ("constant", 3)])
def test_packed_tree_record_equals_the_array_export(oracle, response, K):
    """``pgb_export_trees_packed`` (one call, one record, decoded lazily) carries exactly what the array export
    does, for batches and baseline forests, constant and linear leaves, K-vector leaves; it survives pickling as
    its raw bytes; a buffer that is too small reports the size it needs."""
    import ctypes as C
    import warnings

    from pymc_bart_amd.pgbart import CategoricalLikelihood
    from pymc_bart_amd.trees import PackedTrees

    rng = np.random.default_rng(6)
    X = rng.normal(size=(400, 3))
    Y = X[:, 0] + rng.normal(0, 0.3, 400) if K == 1 else rng.integers(0, K, 400).astype(float)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        op = BARTOp(X, Y, m=7, response=response)
    lik = NormalLikelihood(1.0) if K == 1 else CategoricalLikelihood(K)
    st = PGBART([op], num_particles=6, likelihood=lik, observed=Y, random_seed=2, backend=oracle, batch=(3, 3))
    for it in range(8):
        if it == 4:
            st.stop_tuning()
        st.astep(None)
    s = st.sampler
    for which in (0, 1):
        packed, plain = s.export_trees(which), _export_unpacked(s, which)
        assert isinstance(packed, PackedTrees) and packed._ta is None          # not decoded yet
        assert packed.n_trees == plain.n_trees == (3 if which == 0 else 7) and packed.n_outputs == K
        for f in TREE_FIELDS:
            assert np.array_equal(getattr(packed, f), getattr(plain, f)), f
        twin = pickle.loads(pickle.dumps(packed))
        assert twin.raw == packed.raw and np.array_equal(twin.value, plain.value)
        assert len(packed.raw) < 0.75 * sum(getattr(plain, f).nbytes for f in TREE_FIELDS) or response != "constant"
    if response == "linear":
        assert (s.export_trees(1).svar >= 0).any()
    nb = C.c_int64()
    rc = s.backend.lib.lib.pgb_export_trees_packed(s._h, 1, None, 0, C.byref(nb))
    assert rc == _abi.PGB_E_NOMEM and nb.value == len(s.export_trees(1).raw)
    # the history a step publishes is made of these records and the predictor reads them
    ps = PosteriorSampler.from_history(st._batches, st._baseline, 7, K, backend=oracle)
    assert ps.n_draws == 4 and ps.sample_posterior(X[:5], [0, 3], []).shape == (2, K, 5)


def test_offsets_and_responses_the_likelihood_tables_cannot_address_are_refused(oracle):
    """``pgb_set_offset`` refuses non-finite values and values beyond +-PGB_MAX_OFFSET = 1e6 (the table-driven exp /
    log-Phi take their index from the bits of the linear predictor without a clamp: exact saturation only below
    4.6e7 -- round-4 ADVICE); the offset is reset and the chain stays usable.  ``pgb_set_response`` refuses
    non-finite values.  (The same assertions run against the HIP library in tests/test_parity_gpu.py.)"""
    _refusals(oracle)


def _refusals(backend):
    from pymc_bart_amd.sampler import PyBartSettings, PySampler

    rng = np.random.default_rng(1)
    X = rng.normal(size=(300, 2))
    Y = (rng.random(300) < 0.5).astype(float)
    st = PyBartSettings.from_data(X, Y, m=3, num_particles=4, family="bernoulli_logit", seed=1)
    s = PySampler(st, X, Y, np.zeros(2, np.int32), np.ones(2), backend=backend)
    s.set_likelihood([])
    ok = rng.normal(0, 1, 300)
    s.set_offset(ok)
    ref, _ = s.step(True)
    for bad in (np.inf, -np.inf, np.nan, 1.5e8, -1.0e6 * 1.0000001):
        off = ok.copy()
        off[17] = bad
        with pytest.raises(_abi.PGBError, match="offset has non-finite values or values beyond"):
            s.set_offset(off)
    s.set_offset(np.full(300, 1.0e6))      # the bound itself is inside
    s.set_offset(ok)                        # ... and after a refusal the chain goes on
    again, _ = s.step(True)
    assert np.isfinite(again).all() and again.shape == ref.shape
    yb = Y.copy()
    yb[3] = np.nan
    with pytest.raises(_abi.PGBError, match="response has non-finite"):
        s.set_response(yb)


def test_tree_arrays_pickled_before_a_field_existed_still_load(tmp_path):
    """Round-5 ADVICE: `rule` became a mandatory array of TreeArrays; a history or step method pickled before that is
    restored without __init__ and must not raise on the first concat / predict.  And a format-1 history file whose
    per-column rules array is shorter than a split column is an error with a message, not an IndexError."""
    from pymc_bart_amd.trees import TreeArrays, load_history

    t = TreeArrays.empty(2, 5, 1)
    t.node_off[:] = [0, 2, 5]
    old = {k: v for k, v in t.__dict__.items() if k not in ("rule", "svar")}
    u = TreeArrays.__new__(TreeArrays)
    u.__setstate__(old)
    assert np.array_equal(u.rule, np.zeros(5, np.int32)) and np.array_equal(u.svar, np.full(5, -1))
    assert TreeArrays.concat([u, pickle.loads(pickle.dumps(t))]).total_nodes == 10
    arrs = dict(format=np.array("pgbart-history-1"), n_chains=np.array(1), m=np.array(1), rules=np.array([0, 1], np.int32),
                c0_n_outputs=np.array(1), c0_sizes=np.array([1]), c0_tree_id=np.zeros(1, np.int32),
                c0_node_off=np.array([0, 3], np.int32), c0_var=np.array([4, -1, -1], np.int32), c0_split=np.zeros(3),
                c0_left=np.array([1, -1, -1], np.int32), c0_right=np.array([2, -1, -1], np.int32),
                c0_count=np.array([3, 1, 2]), c0_value=np.zeros((3, 1)))
    np.savez(tmp_path / "h.npz", **arrs)
    with pytest.raises(ValueError, match="column 4"):
        load_history(tmp_path / "h.npz")


@pytest.mark.parametrize("container", ["manager_list", "box"])
def test_four_workers_registering_at_the_same_instant_get_their_own_entries(oracle, container):
    """Round-5 VERDICT, smaller #9: the history slot of a chain is found on the list itself (owner token on the
    baseline forest), not taken as ``len(trees) - 1`` after the append -- four spawned worker processes released by
    one barrier right before their first draw; chain c makes 3 + c draws.  The parent sees four distinct entries with
    3, 4, 5, 6 batches: on the reference's ``Manager().list()`` (``bart.py:134-135``) and on a list-like that is not
    one (every draw RE-ASSIGNS the chain's entry there, so a wrong index would overwrite a neighbour's)."""
    import multiprocessing as mp

    from _workers import box_manager, register_at_barrier

    X, Y = _data(n=90)
    ctx = mp.get_context("spawn")
    mgr = mp.Manager() if container == "manager_list" else box_manager()
    if container == "box":
        mgr.start()
    try:
        trees = mgr.list() if container == "manager_list" else mgr.Box()
        op = BARTOp(X, Y, m=3, all_trees=trees)
        barrier = ctx.Barrier(4)
        procs = [ctx.Process(target=register_at_barrier, args=(op, c, 3 + c, barrier)) for c in range(4)]
        for pr in procs:
            pr.start()
        for pr in procs:
            pr.join(180)
        assert [pr.exitcode for pr in procs] == [0, 0, 0, 0]
        assert len(trees) == 4
        entries = [trees[i] for i in range(4)]
        assert sorted(len(b) for _, b in entries) == [3, 4, 5, 6]
        owners = [base.owner for base, _ in entries]
        assert len(set(owners)) == 4 and all(o and o.count(":") == 2 for o in owners)
        assert len({o.split(":")[0] for o in owners}) == 4      # four worker pids
    finally:
        mgr.shutdown()
