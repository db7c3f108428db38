"""The host-callback likelihood (family "callback", ``pgb_set_loglik_callback``): the slow fallback for
models outside the closed family (upstream evaluates ``model.datalogp`` for every
particle).  The pin: a callback that spells out the asymmetric-Laplace check loss with exactly the
arithmetic of the built-in family (IEEE +, -, *, / only) must reproduce that family's committed
fingerprint BIT FOR BIT -- same draws, same trees, same counters -- because the per-row values are the
same doubles and everything downstream is the same fixed-point algebra."""
import json
import os
import pickle

import numpy as np
import pytest

from _cases import digest, make_case, run_case
from pymc_bart_amd import _abi
from pymc_bart_amd.pgbart import PGBART, BARTOp, CallbackLikelihood
from pymc_bart_amd.sampler import PyBartSettings, PySampler

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_runs.json")))


def check_loss(b, q):
    def logp(y, mu):
        u = (y - mu) / b
        return -(u * np.where(u < 0.0, q - 1.0, q))
    return logp


def callback_case(name="quantile_asymlaplace"):
    c = dict(make_case(name))
    b, q = c.pop("lik_params")
    c["family"] = "callback"
    c["callback"] = check_loss(b, q)
    return c


def run_callback_case(c, backend):
    """run_case with the callback installed (run_case builds the sampler itself: patch its class)."""
    orig = PySampler.__init__

    def init(self, *a, **k):
        orig(self, *a, **k)
        if self.settings.family == "callback":
            self.set_loglik_callback(c["callback"])

    PySampler.__init__ = init
    try:
        return run_case(c, backend)
    finally:
        PySampler.__init__ = orig


def test_callback_reproduces_the_builtin_family_bit_for_bit(oracle):
    got = digest(run_callback_case(callback_case(), oracle))
    assert got == GOLD["quantile_asymlaplace"]


def test_callback_with_offset_reproduces_the_poisson_exposure_structure(oracle):
    """The linear predictor a callback sees includes the offset (log-exposure)."""
    seen = {}

    def logp(y, mu):
        seen["n"] = seen.get("n", 0) + y.size
        seen["max_mu"] = max(seen.get("max_mu", -1e9), float(mu.max()))
        return y * mu - np.exp(np.minimum(mu, 30.0))

    rng = np.random.default_rng(0)
    X = rng.normal(size=(400, 3))
    expo = rng.uniform(0.5, 3.0, 400)
    Y = rng.poisson(expo * np.exp(0.5 * X[:, 0])).astype(float)
    st = PyBartSettings.from_data(X, np.log((Y + 0.5) / expo), m=8, num_particles=6, seed=1, family="callback")
    s = PySampler(st, X, Y, np.zeros(3, np.int32), np.ones(3), backend=oracle)
    s.set_loglik_callback(logp)
    s.set_offset(np.log(expo) + 7.0)
    for it in range(6):
        s.set_likelihood([])
        s.step(it < 3)
    assert seen["n"] > 400 and seen["max_mu"] > 6.0      # the offset is part of what the callback sees


def test_a_raising_callback_surfaces_as_an_error(oracle):
    rng = np.random.default_rng(1)
    X = rng.normal(size=(100, 2))
    Y = rng.normal(size=100)

    def bad(y, mu):
        raise ZeroDivisionError("model logp blew up")

    st = PyBartSettings.from_data(X, Y, m=4, num_particles=4, seed=1, family="callback")
    s = PySampler(st, X, Y, np.zeros(2, np.int32), np.ones(2), backend=oracle)
    with pytest.raises(_abi.PGBError, match="pgb_set_loglik_callback first"):
        s.step(True)
    good = s.checkpoint()
    s.set_loglik_callback(bad)
    with pytest.raises(_abi.PGBError, match="model logp blew up"):
        s.step(True)
    # the step was abandoned half-way: the handle refuses to go on (include/pgbart.h) ...
    s.set_loglik_callback(lambda y, mu: -0.5 * (y - mu) ** 2)
    for call in (lambda: s.step(True), lambda: s.step_async(True, 1), lambda: s.export_trees(0)):
        with pytest.raises(_abi.PGBError, match="abandoned half-way"):
            call()
    # ... until an idle image is restored; the chain then equals one that never saw the bad callback
    s.restore(good)
    twin = PySampler(st, X, Y, np.zeros(2, np.int32), np.ones(2), backend=oracle)
    twin.set_loglik_callback(lambda y, mu: -0.5 * (y - mu) ** 2)
    for _ in range(3):
        a, _ = s.step(True)
        b, _ = twin.step(True)
        assert np.array_equal(a, b)
    with pytest.raises(ValueError, match="n_outputs \\* n"):
        s.set_offset(np.zeros(7))
    with pytest.raises(_abi.PGBError, match="callback family"):
        PySampler(PyBartSettings.from_data(X, Y, m=4, num_particles=4), X, Y, np.zeros(2, np.int32), np.ones(2),
                  backend=oracle).set_loglik_callback(bad)


def _module_level_logp(y, mu):          # picklable
    z = (y - mu) / 0.5
    return -0.5 * z * z


def test_step_method_with_a_callback_likelihood_fits_and_pickles(oracle, monkeypatch):
    import pymc_bart_amd.sampler as sm

    monkeypatch.setattr(sm, "_DEFAULT_BACKEND", oracle)
    rng = np.random.default_rng(2)
    X = rng.uniform(-1, 1, size=(300, 2))
    f = np.sin(3 * X[:, 0])
    Y = f + rng.normal(0, 0.5, 300)
    step = PGBART([BARTOp(X, Y, m=10)], num_particles=8, likelihood=CallbackLikelihood(_module_level_logp),
                  random_seed=4, backend=oracle)
    draws = []
    for it in range(120):
        if it == 60:
            step.stop_tuning()
        mu, stats = step.astep(None)
        if it >= 60:
            draws.append(mu)
    assert np.corrcoef(np.mean(draws, axis=0), f)[0, 1] > 0.8
    twin = pickle.loads(pickle.dumps(step))
    a, _ = step.astep(None)
    b, _ = twin.astep(None)
    assert np.array_equal(a, b)


def test_full_vector_model_logp_as_a_callback(oracle):
    """What the PyMC bridge does for a model outside the closed family: the observed variable's
    elementwise logp, a function of the WHOLE BART vector (with per-row terms the sampler knows nothing
    about), wrapped as a (y, mu, rows) callback.  Against a direct per-row callback: same chain."""
    from pymc_bart_amd._pymc_bridge import FullVectorLogp

    rng = np.random.default_rng(5)
    n = 250
    X = rng.uniform(-1, 1, size=(n, 2))
    scale = rng.uniform(0.25, 1.0, n)                   # heteroscedastic: a per-row term of the likelihood
    scale = 2.0 ** np.round(np.log2(scale))             # powers of two: exact arithmetic on both routes
    Y = np.sin(2 * X[:, 0]) + rng.normal(0, 1, n) * scale
    calls = {"full": 0}

    def model_logp(bart_value):                         # stands in for the compiled PyTensor function
        calls["full"] += 1
        z = (Y - bart_value) / scale
        return -0.5 * z * z

    def per_row(y, mu, rows):
        z = (y - mu) / scale[rows]
        return -0.5 * z * z

    def run(logp):
        st = PyBartSettings.from_data(X, Y, m=6, num_particles=6, seed=9, family="callback")
        s = PySampler(st, X, Y, np.zeros(2, np.int32), np.ones(2), backend=oracle)
        s.set_loglik_callback(logp)
        outs = []
        for it in range(10):
            s.set_likelihood([])
            outs.append(s.step(it < 5)[0])
        return np.array(outs)

    a = run(per_row)
    b = run(FullVectorLogp(model_logp, np.full(n, Y.mean())))
    assert np.array_equal(a, b) and calls["full"] > 10
