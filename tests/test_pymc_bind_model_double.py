"""`_pymc_bridge.bind_model` EXECUTED -- against an in-test double of the slice of PyMC / PyTensor it calls.

PyMC cannot be installed on the build box (round-2 VERDICT: "bind_model has never executed").  The double below
reproduces the public calls `bind_model` makes [P: PyMC >= 5 API] -- `modelcontext`, `Model.initial_point /
value_vars / rvs_to_values / values_to_rvs / observed_RVs / replace_rvs_by_values / logp`,
`pytensorf.inputvars / make_shared_replacements / join_nonshared_inputs / compile`, `graph.traversal.ancestors`,
`rv.owner.op.dist_params` -- with plain Python closures standing in for PyTensor graphs.  It cannot prove the
real API has these shapes; it does prove that the glue runs end to end: the reference call convention
`PGBART([mu], num_particles=5)` inside a model (`tests/test_bart.py:231-235`), family / sigma / offset read from
the model, `step(point)` driven the way `pm.sample` drives it, an additive term that starts at zero, and the
elementwise-logp fallback for a likelihood outside the closed family.
"""
import sys
import types

import numpy as np
import pytest

from test_pymc_api import _ArrayStepShared, _Shared, fake_pymc  # noqa: F401  (fixture)


class Var:
    """A 'tensor variable': a name, an optional evaluator over an environment {value-var name: array}, an owner."""

    def __init__(self, name, fn=None, owner=None, parents=(), data=None, dtype="float64"):
        self.name, self.fn, self.owner, self.parents, self.dtype = name, fn, owner, tuple(parents), dtype
        if data is not None:
            self.data = np.asarray(data)

    def eval_in(self, env):
        return np.asarray(self.fn(env) if self.fn is not None else env[self.name], dtype=float)


class Owner:
    def __init__(self, op, inputs):
        self.op, self.inputs = op, list(inputs)


class DistOp:
    def __init__(self, name):
        self.name = name

    def dist_params(self, node):  # PyMC's RandomVariable ops: inputs = [rng, size, *params]
        return node.inputs[2:]


class FakeModel:
    def __init__(self, free, observed, logp_of):
        self.free_rvs = [rv for rv, _ in free]
        self.value_vars = [v for _, v in free]
        self.rvs_to_values = {rv: v for rv, v in free}
        self.rvs_to_values[observed[0]] = observed[1]
        self.values_to_rvs = {v: rv for rv, v in free}
        self.observed_RVs = [observed[0]]
        self._init = {}
        self._logp_of = logp_of

    def initial_point(self):
        return dict(self._init)

    def replace_rvs_by_values(self, exprs):  # the doubles are written over value-variable names already
        return list(exprs)

    def logp(self, vars=None, sum=True):  # noqa: A002
        assert sum is False and len(vars) == 1
        return [self._logp_of]


def _install_fake_pymc_graph_api(monkeypatch, model):
    pm_model = types.ModuleType("pymc.model")
    pm_model.modelcontext = lambda m: m if m is not None else model
    ptf = types.ModuleType("pymc.pytensorf")
    ptf.inputvars = lambda vs: list(vs)

    def make_shared_replacements(point, vars, mdl):  # noqa: A002
        return {v: _Shared(point[v.name]) for v in mdl.value_vars if v not in vars}

    def join_nonshared_inputs(point, outs, inputs, shared):
        (inp,) = inputs
        shape = np.shape(point[inp.name])

        def env_of(x):
            env = {inp.name: np.asarray(x, float).reshape(shape)}
            env.update({v.name: s.get_value() for v, s in shared.items()})
            return env

        wrapped = [Var(o.name, fn=(lambda x, _o=o: _o.eval_in(env_of(x)))) for o in outs]
        return wrapped, Var("inarray", dtype="float64")

    def compile(inputs, outputs, **kwargs):  # noqa: A001
        if isinstance(outputs, (list, tuple)):
            return _Fn(lambda x: [o.fn(x) for o in outputs])
        return _Fn(lambda x: outputs.fn(x))

    ptf.make_shared_replacements, ptf.join_nonshared_inputs, ptf.compile = make_shared_replacements, join_nonshared_inputs, compile
    trav = types.ModuleType("pytensor.graph.traversal")

    def ancestors(vs):
        seen, todo = [], list(vs)
        while todo:
            v = todo.pop()
            if v in seen:
                continue
            seen.append(v)
            todo += list(getattr(v, "parents", ()))
            if getattr(v, "owner", None) is not None:
                todo += [i for i in v.owner.inputs if isinstance(i, Var)]
        return seen

    trav.ancestors = ancestors
    for name, mod in (("pymc.model", pm_model), ("pymc.pytensorf", ptf), ("pytensor", types.ModuleType("pytensor")),
                      ("pytensor.graph", types.ModuleType("pytensor.graph")), ("pytensor.graph.traversal", trav)):
        monkeypatch.setitem(sys.modules, name, mod)


class _Fn:
    """A compiled-function stand-in (callable object: `fn.trust_input = True` must be settable)."""

    def __init__(self, f):
        self._f = f

    def __call__(self, x):
        return self._f(x)


def _build_model(kind, n=200, seed=5):
    """y ~ Dist(BART + b * z, ...) with b, sigma owned by other samplers; returns (model, pieces)."""
    from pymc_bart_amd.pgbart import BARTOp

    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, 2))
    z = rng.normal(size=n)
    f = np.where(X[:, 0] > 0, 1.0, -1.0)
    if kind == "normal":
        Y = f + 2.0 * z + rng.normal(0, 0.2, n)
        bart_Y = Y
    elif kind == "studentt":  # heavy tails: StudentT(nu = 4, mu = BART + b z, sigma)
        Y = f + 2.0 * z + 0.2 * rng.standard_t(4, n)
        bart_Y = Y
    else:  # an observed distribution outside the closed family: Laplace noise, scale 0.2
        Y = f + 2.0 * z + rng.laplace(0, 0.2, n)
        bart_Y = Y
    op = BARTOp(X, bart_Y, m=8, name="BART")
    mu_rv = Var("mu", owner=Owner(op, []))
    mu_val, sig_val, b_val = Var("mu"), Var("sigma"), Var("b")
    sig_rv, b_rv = Var("sigma_rv"), Var("b_rv")
    mean = Var("mean", fn=lambda env: env["mu"] + env["b"] * z, parents=(mu_rv, b_rv))
    sigma = Var("sigma_expr", fn=lambda env: np.full(n, float(env["sigma"])), parents=(sig_rv,))
    dist = DistOp({"normal": "normal", "studentt": "studentt"}.get(kind, "laplace"))
    params = [mean, sigma]
    if kind == "studentt":  # [P] StudentT's parameters: (nu, mu, sigma)
        params = [Var("nu_expr", fn=lambda env: np.full(n, 4.0)), mean, sigma]
    y_rv = Var("y", owner=Owner(dist, [Var("rng"), Var("size")] + params))
    y_val = Var("y_obs", data=Y)
    logp_el = Var("logp", fn=lambda env: -np.abs(Y - (env["mu"] + env["b"] * z)) / float(env["sigma"])
                  - np.log(2 * float(env["sigma"])))
    model = FakeModel([(mu_rv, mu_val), (sig_rv, sig_val), (b_rv, b_val)], (y_rv, y_val), logp_el)
    model._init = {"mu": np.full(n, bart_Y.mean()), "sigma": np.array(0.2), "b": np.array(0.0)}  # b starts at ZERO
    return model, dict(X=X, Y=Y, z=z, f=f, mu_rv=mu_rv, op=op)


@pytest.fixture()
def graph_api(fake_pymc, monkeypatch):  # noqa: F811
    pm, pgb = fake_pymc
    # the double of ArrayStepShared keys `shared` by NAME; PyMC's own maps variables to names itself
    orig_init = _ArrayStepShared.__init__

    def init(self, vars, shared, blocked=True, rng=None):  # noqa: A002
        orig_init(self, vars, {getattr(k, "name", k): v for k, v in shared.items()}, blocked, rng)

    monkeypatch.setattr(_ArrayStepShared, "__init__", init)
    monkeypatch.setattr(pgb, "_HAVE_PYMC", True)
    return pm, pgb


def test_bind_model_runs_the_reference_call_convention_on_a_normal_model(graph_api, oracle, monkeypatch):
    _, pgb = graph_api
    model, P = _build_model("normal")
    _install_fake_pymc_graph_api(monkeypatch, model)
    step = pgb.PGBART([P["mu_rv"]], num_particles=6, batch=(1.0, 1.0), model=model, backend=oracle, random_seed=3)
    b = step._binding
    assert b.kind == "normal" and b.likelihood.family == "normal" and step.bart is P["op"]
    assert step.var_names == ("mu",) and set(step.shared) == {"sigma", "b"}
    assert P["op"].n_outputs == 1 and np.array_equal(step._y_obs, P["Y"])
    point = model.initial_point()
    rng = np.random.default_rng(0)
    mus = []
    for it in range(40):
        if it == 20:
            step.stop_tuning()
        point, stats = step.step(point)                      # shared <- point, astep(q), mu back into the point
        assert stats[0]["tune"] == (it < 20)
        res = P["Y"] - point["mu"]
        point["b"] = np.array(float(res @ P["z"] / (P["z"] @ P["z"])) + rng.normal(0, 0.01))   # "NUTS" moves b ...
        r2 = res - float(point["b"]) * P["z"]
        point["sigma"] = np.array(float(np.sqrt((1.0 + 0.5 * r2 @ r2) / rng.gamma(1.0 + 0.5 * len(r2)))))  # ... and sigma
        if it >= 20:
            mus.append(point["mu"])
    fit = np.mean(mus, axis=0)
    assert abs(float(point["b"]) - 2.0) < 0.2                    # the coefficient was found from b = 0
    assert np.corrcoef(fit, P["f"])[0, 1] > 0.9                  # BART explains f, the offset explains 2 z
    assert abs(np.corrcoef(fit, P["z"])[0, 1]) < 0.3
    assert len(P["op"].all_trees) == 1 and len(P["op"].all_trees[0][1]) == 20


def test_bind_model_falls_back_to_the_models_elementwise_logp(graph_api, oracle, monkeypatch):
    _, pgb = graph_api
    model, P = _build_model("laplace")
    _install_fake_pymc_graph_api(monkeypatch, model)
    step = pgb.PGBART([P["mu_rv"]], num_particles=5, batch=(1.0, 1.0), model=model, backend=oracle, random_seed=4)
    assert step._binding.kind == "callback" and step.likelihood.family == "callback"
    point = model.initial_point()
    point["b"] = np.array(2.0)                                   # the other sampler's current value
    for it in range(12):
        if it == 6:
            step.stop_tuning()
        step.likelihood.logp.set_base(point["mu"])                # (the bridge's callback scatters into the current value)
        point, _ = step.step(point)
    assert np.corrcoef(point["mu"], P["f"])[0, 1] > 0.7          # fitted through the model's own logp, offset included


def test_bind_model_puts_a_student_t_model_on_the_device_family(graph_api, oracle, monkeypatch):
    """StudentT(nu, BART + b z, sigma) is one of the kernel families: no callback, sigma and nu read from the
    model every step in the kernel's order, the other term followed as an offset of the linear predictor."""
    _, pgb = graph_api
    model, P = _build_model("studentt")
    _install_fake_pymc_graph_api(monkeypatch, model)
    step = pgb.PGBART([P["mu_rv"]], num_particles=6, batch=(1.0, 1.0), model=model, backend=oracle, random_seed=5)
    assert step._binding.kind == "student_t" and step.likelihood.family == "student_t"
    point = model.initial_point()
    point["b"] = np.array(2.0)
    point["sigma"] = np.array(0.25)
    for it in range(60):  # (one key, one chain: long enough that the fit does not hang on the draw of the key)
        if it == 30:
            step.stop_tuning()
        point, _ = step.step(point)
    params, off = step._binding.current()
    assert params == [0.25, 4.0] and np.allclose(off, 2.0 * P["z"])
    assert np.corrcoef(point["mu"], P["f"])[0, 1] > 0.9          # BART explains f, the offset carries 2 z
    assert abs(np.corrcoef(point["mu"], P["z"])[0, 1]) < 0.3


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["normal", "studentt"])
def test_step_method_bound_to_a_model_gives_the_same_chain_on_gpu_and_oracle(graph_api, hip, oracle, monkeypatch, kind):
    """Row f3 on the GPU, against something other than itself: the step method built the reference way
    (``PGBART([mu_rv], ...)`` inside a model; family, sigma / nu and the other additive term read from the model
    every step) driven through ``step(point)`` with another sampler moving ``b`` and ``sigma`` in between -- the
    HIP backend and the oracle must produce the same points, statistics and tree history, bit for bit."""
    _, pgb = graph_api

    def run(backend):
        np.random.seed(20261002)   # older PyMC seeds NumPy's global generator per chain; the step method keys itself from it
        model, P = _build_model(kind)
        _install_fake_pymc_graph_api(monkeypatch, model)
        step = pgb.PGBART([P["mu_rv"]], num_particles=8, batch=(0.5, 0.5), model=model, backend=backend, random_seed=9)
        point = model.initial_point()
        rng = np.random.default_rng(1)
        out = []
        for it in range(24):
            if it == 12:
                step.stop_tuning()
            point, stats = step.step(point)
            res = P["Y"] - point["mu"]
            point["b"] = np.array(float(res @ P["z"] / (P["z"] @ P["z"])) + rng.normal(0, 0.01))       # "NUTS" moves b
            r2 = res - float(point["b"]) * P["z"]
            point["sigma"] = np.array(float(np.sqrt((1.0 + 0.5 * r2 @ r2) / rng.gamma(1.0 + 0.5 * len(r2)))))
            out.append((point["mu"].copy(), bytes(np.ascontiguousarray(stats[0]["variable_inclusion"]).tobytes()),
                        float(point["b"]), float(point["sigma"])))
        base, batches = P["op"].all_trees[0]
        return step, out, base, list(batches)

    sg, og, bg, tg = run(hip)
    so, oo, bo, to = run(oracle)
    assert sg.sampler.backend.lib.backend_name == "hip-gfx950" and sg._binding.kind == so._binding.kind
    for (mg, vg, b_g, s_g), (mo, vo, b_o, s_o) in zip(og, oo):
        assert np.array_equal(mg, mo) and vg == vo and b_g == b_o and s_g == s_o
    assert len(tg) == len(to) == 12
    for a, b in zip([bg] + tg, [bo] + to):
        for f in ("var", "split", "value", "count", "left", "right"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
