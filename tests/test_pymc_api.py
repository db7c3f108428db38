"""PGBART against the PyMC step-method API -- with an in-test double of
``pymc.step_methods`` because PyMC cannot be installed on the build box.

The double reproduces what ``pm.sample`` does with a step method [P]:
``ArrayStepShared.step(point)`` copies the OTHER variables' current values into shared variables, ravels
the step's own variable into ``q``, calls ``astep(q)`` with that single argument and puts the result
back into the point; ``pm.STEP_METHODS`` is the registry ``import pymc_bart`` extends
(reference ``pymc_bart/__init__.py:15-18``); ``competence`` decides the assignment
(reference ``tests/test_bart.py:167-208``); the manual form is ``PGBART([mu1], num_particles=5)`` handed to
``pm.sample(step=[...])`` (``tests/test_bart.py:231-235``).
"""
import enum
import importlib
import pickle
import sys
import types

import numpy as np
import pytest


class _Shared:
    def __init__(self, value):
        self.value = np.asarray(value, float)

    def set_value(self, v, borrow=False):
        self.value = np.asarray(v, float)

    def get_value(self):
        return self.value


class _Competence(enum.IntEnum):
    INCOMPATIBLE = 0
    COMPATIBLE = 1
    PREFERRED = 2
    IDEAL = 3


class _ArrayStepShared:
    """[P] pymc.step_methods.arraystep.ArrayStepShared, reduced to what a single-variable step uses."""

    def __init__(self, vars, shared, blocked=True, rng=None):  # noqa: A002
        self.vars = vars
        self.var_names = tuple(v.name for v in vars)
        self.shared = dict(shared)
        self.blocked = blocked

    def step(self, point):
        for name, shared_var in self.shared.items():
            shared_var.set_value(point[name], borrow=True)
        q = np.concatenate([np.ravel(point[n]) for n in self.var_names])
        apoint, stats = self.astep(q)                       # ONE argument, as PyMC calls it
        new_point = dict(point)
        (name,) = self.var_names
        new_point[name] = np.asarray(apoint).reshape(np.shape(point[name]))
        return new_point, stats


@pytest.fixture()
def fake_pymc(monkeypatch, oracle):
    pm = types.ModuleType("pymc")
    pm.STEP_METHODS = ["NUTS", "Metropolis"]
    sm = types.ModuleType("pymc.step_methods")
    arr = types.ModuleType("pymc.step_methods.arraystep")
    arr.ArrayStepShared = _ArrayStepShared
    comp = types.ModuleType("pymc.step_methods.compound")
    comp.Competence = _Competence
    for name, mod in (("pymc", pm), ("pymc.step_methods", sm), ("pymc.step_methods.arraystep", arr),
                      ("pymc.step_methods.compound", comp)):
        monkeypatch.setitem(sys.modules, name, mod)
    import pymc_bart_amd
    import pymc_bart_amd.pgbart as pgb
    import pymc_bart_amd.sampler as smp

    # (the module is re-executed under the double and its ORIGINAL namespace put back afterwards: test modules
    #  that imported PGBART at collection keep the very class object `pymc_bart_amd.pgbart.PGBART` names, which
    #  pickling checks -- whatever order the test files run in)
    saved = {m: dict(m.__dict__) for m in (pgb, pymc_bart_amd)}
    importlib.reload(pgb)
    importlib.reload(pymc_bart_amd)
    monkeypatch.setattr(smp, "_DEFAULT_BACKEND", oracle)
    # older PyMC seeds NumPy's global generator per chain before the first step, and a step method nobody keyed
    # mixes one draw from it into its key (PGBART._key_for_this_process): without this line every run of the suite
    # samples a different chain (that is how deviation 12's dead chain surfaced as a 1-in-35 flake)
    np.random.seed(20261002)
    yield pm, pgb
    for name in ("pymc", "pymc.step_methods", "pymc.step_methods.arraystep", "pymc.step_methods.compound"):
        sys.modules.pop(name, None)
    for m, ns in saved.items():
        m.__dict__.clear()
        m.__dict__.update(ns)


def _two_term_data(seed=3415, n=30):
    rng = np.random.default_rng(seed)
    X1, X2 = rng.normal(size=(n, 3)), rng.normal(size=(n, 2))
    Y1 = X1[:, 0] + rng.normal(0, 0.1, n)
    Y2 = X2[:, 1] + rng.normal(0, 0.1, n)
    return X1, X2, Y1, Y2, Y1 + Y2 + rng.normal(0, 0.1, n)


def test_import_registers_the_step_method_and_competence_is_ideal(fake_pymc):
    pm, pgb = fake_pymc
    assert pgb.PGBART in pm.STEP_METHODS and pm.STEP_METHODS[:2] == ["NUTS", "Metropolis"]
    assert issubclass(pgb.PGBART, _ArrayStepShared)
    X1, _, Y1, _, _ = _two_term_data()
    op = pgb.BARTOp(X1, Y1, m=3, name="mu1")
    assert pgb.PGBART.competence(op, has_grad=False) is _Competence.IDEAL
    assert pgb.PGBART.competence(object(), has_grad=True) is _Competence.INCOMPATIBLE
    assert pgb.PGBART.name == "pgbart" and pgb.PGBART.generates_stats and not pgb.PGBART.default_blocked
    assert pgb.PGBART.stats_dtypes_shapes == {"variable_inclusion": (object, []), "tune": (bool, [])}


def test_manual_steps_for_two_bart_variables_run_through_step_point(fake_pymc, oracle):
    """The reference's only direct use of the boundary (tests/test_bart.py:211-241): two BART terms in one
    Normal likelihood, one PGBART each, sigma owned by another sampler.  Everything reaches the step
    methods the way PyMC delivers it: through ``step(point)`` -> shared variables -> ``astep(q)``."""
    _, pgb = fake_pymc
    X1, X2, Y1, Y2, Y = _two_term_data()
    sigma_sh = {1: _Shared(1.0), 2: _Shared(1.0)}
    other_sh = {1: _Shared(np.zeros(30)), 2: _Shared(np.zeros(30))}
    mu1, mu2 = pgb.BARTOp(X1, Y1, m=3, name="mu1"), pgb.BARTOp(X2, Y2, m=3, name="mu2")

    def make(op, k, other_name):
        # y ~ Normal(mu_k + mu_other, sigma): the other term is this step's offset, read from ITS shared copy
        st = pgb.PGBART([op], num_particles=5, likelihood=pgb.NormalLikelihood(sigma_sh[k]), observed=Y,
                        shared={"sigma": sigma_sh[k], other_name: other_sh[k]}, random_seed=3415, backend=oracle)
        st.offset_from = other_sh[k]
        return st

    step1, step2 = make(mu1, 1, "mu2"), make(mu2, 2, "mu1")
    for st in (step1, step2):  # additive model: the offset comes from the shared copy of the other term
        orig = st.astep
        st.astep = (lambda q, _o=orig, _s=st: _o(q, offset=_s.offset_from.get_value()))
    point = {"mu1": np.full(30, Y1.mean()), "mu2": np.full(30, Y2.mean()), "sigma": 0.5}
    rng = np.random.default_rng(1)
    draws = {"mu1": [], "mu2": []}
    for it in range(40):
        if it == 20:
            step1.stop_tuning()
            step2.stop_tuning()
        for st in (step1, step2):
            point, stats = st.step(point)
            assert set(stats[0]) == {"variable_inclusion", "tune"} and stats[0]["tune"] == (it < 20)
        res = Y - point["mu1"] - point["mu2"]
        point["sigma"] = float(np.sqrt((1.0 + 0.5 * res @ res) / rng.gamma(1.0 + 15.0)))
        if it >= 20:
            draws["mu1"].append(point["mu1"])
            draws["mu2"].append(point["mu2"])
    assert np.array(draws["mu1"]).shape == (20, 30) and np.array(draws["mu2"]).shape == (20, 30)
    assert step1.sampler.settings.n == 30 and sigma_sh[1].get_value() == pytest.approx(point["sigma"], rel=0.5)
    fit = np.mean(draws["mu1"], axis=0) + np.mean(draws["mu2"], axis=0)
    assert np.corrcoef(fit, Y)[0, 1] > 0.8
    assert len(mu1.all_trees) == 1 and len(mu2.all_trees) == 1      # separate histories per BART variable
    assert len(mu1.all_trees[0][1]) == 20


def test_sigma_arrives_through_the_shared_variable_not_through_a_point_argument(fake_pymc, oracle):
    _, pgb = fake_pymc
    rng = np.random.default_rng(2)
    X = rng.normal(size=(200, 3))
    Y = X[:, 0] + rng.normal(0, 0.3, 200)
    sh = _Shared(1.0)

    def run(sigmas):
        st = pgb.PGBART([pgb.BARTOp(X, Y, m=5)], num_particles=6, batch=(1.0, 1.0), likelihood=pgb.NormalLikelihood(sh),
                        shared={"sigma": sh}, random_seed=7, backend=oracle)
        point = {"mu": np.full(200, Y.mean()), "sigma": 1.0}
        outs = []
        for s in sigmas:
            point["sigma"] = s
            point, _ = st.step(point)
            outs.append(point["mu"].copy())
        return np.array(outs)

    a = run([1.0, 0.05, 0.2, 3.0])
    b = run([1.0, 0.05, 0.2, 3.0])
    c = run([1.0, 50.0, 0.2, 3.0])
    assert np.array_equal(a, b)
    assert np.array_equal(a[0], c[0]) and not np.array_equal(a[1], c[1])   # the second step saw another sigma
    with pytest.raises(KeyError, match="shared"):                          # a name needs a point; PyMC passes none
        pgb.PGBART([pgb.BARTOp(X, Y, m=5)], num_particles=6, likelihood=pgb.NormalLikelihood("sigma"),
                   random_seed=7, backend=oracle).astep(np.zeros(200))


def test_pickled_step_method_keeps_the_pymc_surface(fake_pymc, oracle):
    _, pgb = fake_pymc
    rng = np.random.default_rng(3)
    X = rng.normal(size=(100, 2))
    Y = X[:, 1] + rng.normal(0, 0.2, 100)
    sh = _Shared(0.5)
    st = pgb.PGBART([pgb.BARTOp(X, Y, m=4)], num_particles=5, likelihood=pgb.NormalLikelihood(0.5),
                    shared={"sigma": sh}, random_seed=5, backend=oracle)
    point = {"mu": np.full(100, Y.mean()), "sigma": 0.5}
    point, _ = st.step(point)
    twin = pickle.loads(pickle.dumps(st))
    a, _ = st.step(point)
    b, _ = twin.step(point)
    assert np.array_equal(a["mu"], b["mu"]) and twin.var_names == ("mu",)


def test_device_choice_of_a_worker_process(monkeypatch):
    import torch

    from pymc_bart_amd import pgbart as pgb

    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    picked = []
    monkeypatch.setattr(torch.cuda, "set_device", lambda i: picked.append(i))
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert pgb._pick_device() == 5
    monkeypatch.delenv("LOCAL_RANK")
    monkeypatch.setenv("PGBART_DEVICE", "11")
    assert pgb._pick_device() == 3          # modulo the visible devices
    monkeypatch.delenv("PGBART_DEVICE")
    import multiprocessing as mp

    monkeypatch.setattr(mp.current_process(), "_identity", (3,), raising=False)
    assert pgb._pick_device() == 2          # third worker of the pool -> GPU 2
    assert picked == [5, 3, 2]
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 0)
    assert pgb._pick_device() is None


# ---- the family a model uses, decided from numeric probes (pymc_bart_amd/_pymc_bridge.py)
def _probe_factory(kind, n=40, K=1, offset=None, **par):
    from scipy.special import expit, ndtr

    off = np.zeros(n) if offset is None else offset

    def probe(x):
        x = np.asarray(x, float)
        if kind == "normal":
            return "normal", [x + off, np.full(n, par.get("sigma", 0.7))]
        if kind == "meanscale":
            return "normal", [x[0], np.abs(x[1])]
        if kind == "probit":
            return "bernoulli", [ndtr(x + off)]
        if kind == "logit":
            return "bernoulli", [expit(x + off)]
        if kind == "cloglog":
            return "bernoulli", [1 - np.exp(-np.exp(x))]
        if kind == "softmax":
            e = np.exp(x - x.max(axis=0))
            return "categorical", [(e / e.sum(axis=0)).T]       # (n, K) as the reference model writes it
        if kind == "poisson":
            return "poisson", [np.exp(x + off)]
        if kind == "negbin":
            return "negative_binomial", [np.exp(x + off), np.full(n, par.get("alpha", 2.0))]
        if kind == "negbin_np":     # [P] the nbinom variable's own parameters: n = alpha, p = alpha / (mu + alpha)
            a = par.get("alpha", 2.0)
            return "nbinom", [np.full(n, a), a / (np.exp(x + off) + a)]
        if kind == "hetero":
            return "normal", [x, np.exp(0.1 * x)]
        if kind == "student_t":     # [P] StudentT: (nu, mu, sigma)
            return "studentt", [np.full(n, par.get("nu", 4.0)), x + off, np.full(n, par.get("sigma", 0.6))]
        if kind == "ald":           # [P] AsymmetricLaplace: (b, kappa, mu)
            return "asymmetriclaplace", [np.full(n, par.get("b", 2.0)), np.full(n, par.get("kappa", 3.0)), x + off]
        if kind == "gamma_scale":   # Gamma(alpha, scale = mean / alpha)
            a = par.get("alpha", 3.0)
            return "gamma", [np.full(n, a), np.exp(x + off) / a]
        if kind == "gamma_rate":    # older PyTensor: Gamma(alpha, rate = alpha / mean)
            a = par.get("alpha", 3.0)
            return "gamma", [np.full(n, a), a / np.exp(x + off)]
        if kind == "gamma_musigma":  # Gamma(mu = exp(BART), sigma fixed): alpha moves with BART -- not in the family
            m = np.exp(x)
            return "gamma", [m * m / 0.25, 0.25 / m]
        return "weibull", [x]

    return probe


def test_likelihood_family_is_identified_numerically():
    from pymc_bart_amd._pymc_bridge import identify

    n = 40
    off = np.linspace(-1, 1, n)
    b = identify(_probe_factory("normal", sigma=0.7), (n,))
    params, o = b.current()
    assert b.likelihood.family == "normal" and params == [0.7] and o.shape == (n,) and not o.any()
    b = identify(_probe_factory("normal", offset=off, sigma=1.3), (n,))
    params, o = b.current()
    assert params == [1.3] and np.allclose(o, off)                 # the second BART term / a fixed effect
    assert identify(_probe_factory("meanscale"), (2, n)).likelihood.family == "normal_meanscale"
    b = identify(_probe_factory("probit", offset=off), (n,))
    assert b.likelihood.family == "bernoulli_probit" and np.allclose(b.current()[1], off, atol=1e-9)
    assert identify(_probe_factory("logit"), (n,)).likelihood.family == "bernoulli_logit"
    b = identify(_probe_factory("softmax", K=3), (3, n))
    assert b.likelihood.family == "categorical" and b.likelihood.n_outputs == 3
    b = identify(_probe_factory("poisson", offset=np.log(np.linspace(0.5, 4, n))), (n,))
    assert b.likelihood.family == "poisson_log" and b.current()[1] is not None
    b = identify(_probe_factory("negbin", alpha=2.5), (n,))
    assert b.likelihood.family == "negbin_log" and b.current()[0] == [2.5]
    b = identify(_probe_factory("negbin_np", offset=off, alpha=1.5), (n,))
    params, o = b.current()
    assert b.likelihood.family == "negbin_log" and params == [1.5] and np.allclose(o, off)
    b = identify(_probe_factory("student_t", offset=off, nu=5.0, sigma=0.4), (n,))
    params, o = b.current()
    assert b.likelihood.family == "student_t" and params == [0.4, 5.0] and np.allclose(o, off)   # kernel order: sigma, nu
    b = identify(_probe_factory("ald", offset=off, b=2.0, kappa=3.0), (n,))
    params, o = b.current()
    assert b.likelihood.family == "asymmetric_laplace" and np.allclose(o, off)
    assert params[1] == pytest.approx(0.9) and params[0] == pytest.approx(np.sqrt(0.9 * 0.1) / 2.0)   # q = k^2 / (1 + k^2)
    for kind in ("gamma_scale", "gamma_rate"):
        b = identify(_probe_factory(kind, offset=off, alpha=3.0), (n,))
        params, o = b.current()
        assert b.likelihood.family == "gamma_log" and params == [3.0] and np.allclose(o, off)
    for bad in ("cloglog", "hetero", "gamma_musigma", "weibull"):
        with pytest.raises(NotImplementedError):
            identify(_probe_factory(bad), (n,))


def test_asymmetric_laplace_parameters_map_pymc_density_onto_the_kernel_family(oracle):
    """[P] PyMC: logp = log(b / (kappa + 1/kappa)) - b kappa (y - mu) for y >= mu, - (b / kappa)(mu - y) below
    (AsymmetricLaplace.logp).  With (s, q) from the bridge the kernel family must give the same DIFFERENCES in mu."""
    import ctypes as C

    from pymc_bart_amd import _abi
    from pymc_bart_amd._pymc_bridge import _ald_params

    f = oracle.lib.lib.pgbo_loglikq
    f.restype, f.argtypes = None, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_void_p]
    rng = np.random.default_rng(12)
    y, mu_a, mu_b = rng.normal(0, 2, 2000), rng.normal(0, 1, 2000), rng.normal(0, 1, 2000)

    def pymc_logp(mu, b, kappa):
        v = y - mu
        return np.log(b / (kappa + 1 / kappa)) - np.where(v >= 0, b * kappa * v, -(b / kappa) * v)

    for b_, kappa in ((2.0, 3.0), (0.7, 1.0), (5.0, 0.4)):
        s_, q_ = _ald_params(b_, kappa)
        la, lb = np.zeros(2000), np.zeros(2000)
        f(_abi.FAMILIES["asymmetric_laplace"], y.ctypes.data, mu_a.ctypes.data, 2000, s_, q_, la.ctypes.data)
        f(_abi.FAMILIES["asymmetric_laplace"], y.ctypes.data, mu_b.ctypes.data, 2000, s_, q_, lb.ctypes.data)
        ok = (la > -2047) & (lb > -2047)
        np.testing.assert_allclose((la - lb)[ok], (pymc_logp(mu_a, b_, kappa) - pymc_logp(mu_b, b_, kappa))[ok], rtol=1e-10, atol=1e-10)


def test_an_offset_that_is_zero_at_bind_time_is_still_followed(fake_pymc, oracle):
    """Round-2 ADVICE (high): ``y ~ Normal(BART + b * x, sigma)`` with ``b ~ Normal`` starts at b = 0, so the
    probe at the model's initial point sees no offset; NUTS then moves ``b``.  The binding must keep reporting
    the other terms and the step method must fit ``y - b * x``, not the raw response."""
    from pymc_bart_amd._pymc_bridge import identify

    _, pgb = fake_pymc
    rng = np.random.default_rng(8)
    n = 200
    X = rng.normal(size=(n, 2))
    z = rng.normal(size=n)
    f = np.where(X[:, 0] > 0, 1.0, -1.0)
    Y = f + 3.0 * z + rng.normal(0, 0.1, n)
    b_now = {"b": 0.0}

    def probe(x):
        return "normal", [np.asarray(x, float) + b_now["b"] * z, np.full(n, 0.3)]

    binding = identify(probe, (n,))                        # bound while b == 0
    assert binding.has_offset and not binding.current()[1].any()

    def make():
        st = pgb.PGBART([pgb.BARTOp(X, Y, m=8)], num_particles=6, batch=(1.0, 1.0), likelihood=binding.likelihood,
                        observed=Y, random_seed=4, backend=oracle)
        st._binding = binding
        return st

    st = make()
    uploads = []
    orig = st._apply_offset
    st._apply_offset = lambda o: (uploads.append(np.array(o)), orig(o))[1]
    st.astep(None)
    assert uploads == []                                    # zero offset, nothing held: nothing to upload
    b_now["b"] = 3.0                                        # the other sampler moved b
    for it in range(30):
        if it == 15:
            st.stop_tuning()
        mu, _ = st.astep(None)
    assert len(uploads) == 1 and np.allclose(uploads[0], 3.0 * z)   # uploaded once, when it changed
    assert np.corrcoef(mu, f)[0, 1] > 0.9                   # BART explains f, not f + 3 z
    assert abs(np.corrcoef(mu, z)[0, 1]) < 0.3
    # the same chain with the offset handed in explicitly: bit-identical
    ref = make()
    ref._binding = None
    ref.astep(None)
    for it in range(30):
        if it == 15:
            ref.stop_tuning()
        mu_ref, _ = ref.astep(None, offset=3.0 * z)
    assert np.array_equal(mu, mu_ref)


class _Owner:
    def __init__(self, op):
        self.op = op


class _ModelVar:  # a model variable (not duck-typed): has an owner whose op carries the BART attributes
    def __init__(self, op):
        self.owner = _Owner(op)
        self.name = "mu"


def test_every_chain_copy_of_a_step_method_gets_its_own_key(fake_pymc, oracle, monkeypatch):
    """PyMC builds ONE step method and copies it into every chain.  The copies must not draw the same forests:
    `set_rng` ([P] PyMC >= 5.17 calls it per chain) keys a copy reproducibly; without it, the first astep of a copy
    keys itself from NumPy's global generator (which older PyMC seeds per chain) and the worker's ordinal."""
    import multiprocessing as mp

    _, pgb = fake_pymc
    rng = np.random.default_rng(12)
    X = rng.normal(size=(150, 2))
    Y = X[:, 0] + rng.normal(0, 0.3, 150)

    RV = _ModelVar
    def build():
        st = pgb.PGBART([RV(pgb.BARTOp(X, Y, m=5))], num_particles=5, likelihood=pgb.NormalLikelihood(1.0),
                        random_seed=7, backend=oracle)
        assert not st._keyed                       # under PyMC an explicit seed is the BASE of the chains' keys
        return st

    def run(st, k=4):
        return np.array([st.astep(None)[0] for _ in range(k)])

    # set_rng: same generator state -> same chain, different -> different; reproducible across copies
    a, b, c = build(), build(), build()
    a.set_rng(np.random.default_rng(100))
    b.set_rng(np.random.default_rng(100))
    c.set_rng(np.random.default_rng(101))
    ra, rb, rc = run(a), run(b), run(c)
    assert np.array_equal(ra, rb) and not np.array_equal(ra, rc)
    key = a.settings.seed
    a.set_rng(np.random.default_rng(5))            # after the first step the key stays (the forest is not restarted)
    assert a.settings.seed == key
    # no set_rng: two copies of ONE step method in two "worker processes" of an older PyMC
    parent = build()
    blob = pickle.dumps(parent)
    outs = []
    for ident, seed in (((1,), 555), ((2,), 555), ((1,), 556), ((1,), 555)):
        monkeypatch.setattr(mp.current_process(), "_identity", ident, raising=False)
        np.random.seed(seed)                       # what pm.sample does in the process that runs the chain
        outs.append(run(pickle.loads(blob)))
    assert not np.array_equal(outs[0], outs[1])    # other worker
    assert not np.array_equal(outs[0], outs[2])    # other chain seed
    assert np.array_equal(outs[0], outs[3])        # same worker ordinal, same chain seed: reproducible
    # without PyMC semantics (duck-typed op, explicit seed) nothing is re-keyed: the chain is a function of the seed
    d1 = pgb.PGBART([pgb.BARTOp(X, Y, m=5)], num_particles=5, random_seed=7, backend=oracle)
    d2 = pgb.PGBART([pgb.BARTOp(X, Y, m=5)], num_particles=5, random_seed=7, backend=oracle)
    assert d1._keyed and np.array_equal(run(d1), run(d2))
