"""What the step method needs from a PyMC model, and how it gets it.

Upstream's PGBART evaluates the model's whole data log-likelihood through PyTensor for every particle
(``SURVEY.md`` Appendix A: ``update_weight``).  A device kernel cannot call PyTensor, so this step
method works with a closed family of likelihoods (``DESIGN.md`` deviation 9) and has to find out
WHICH member a model uses, and -- every step -- the current value of its other parameters.

It does so numerically rather than by pattern-matching graphs: ``probe(x)`` evaluates the parameters
of the observed variable's distribution with the BART variable set to ``x`` and every other variable
at its current (shared) value; a handful of probes decide the family and the link, and ONE probe per
step (``x = 0``) yields the offset (everything added to the BART term: a second BART variable,
reference ``tests/test_bart.py:211-241``; a log-exposure) and the scalar parameters (``sigma``, ...).

``identify`` / ``Binding`` are plain NumPy and tested without PyMC.  ``bind_model`` builds ``probe``
from a real model with PyTensor; PyMC is not installable on the build box, so that function follows the
public PyMC API from memory (marked [P]).  A single-output model outside the closed family is not refused:
it runs through the host-callback family on the model's own elementwise logp (``FullVectorLogp``, tested
here with a NumPy stand-in for the compiled function) -- slowly, like upstream, but unchanged.
"""

from __future__ import annotations

import numpy as np
from scipy.special import expit, ndtr, ndtri

from .pgbart import (AsymmetricLaplaceLikelihood, BernoulliLikelihood, CallbackLikelihood, CategoricalLikelihood,
                     GammaLikelihood, NegativeBinomialLikelihood, NormalLikelihood, NormalMeanScaleLikelihood,
                     PoissonLikelihood, StudentTLikelihood)

_TOL = 1e-8


class Binding:
    """The likelihood of a model as the sampler sees it: a family object plus, per step, the scalar
    parameters and the offset of the linear predictor, both read from ``probe`` at ``x = 0``."""

    #: kinds whose linear predictor may carry other additive terms of the model
    OFFSET_KINDS = ("normal", "bernoulli_probit", "bernoulli_logit", "poisson", "negbin", "negbin_np", "student_t",
                    "asymmetric_laplace", "gamma_scale", "gamma_rate")

    def __init__(self, likelihood, probe, shape, kind, has_offset=None):
        self.likelihood = likelihood
        self.probe = probe
        self.shape = shape
        self.kind = kind
        # Whether the model HAS other additive terms is never inferred from a value: `b * x` with
        # b ~ Normal starts at b = 0, so a zero offset at bind time says nothing about later steps
        # (round-2 ADVICE).  Every kind that can carry an offset reports it on every step; the step
        # method uploads it only when it changed.  (`has_offset` is accepted for older callers.)
        self.has_offset = kind in self.OFFSET_KINDS

    def current(self):
        """(params, offset) at the current values of the other variables; ``offset`` is an array for
        every kind in ``OFFSET_KINDS`` (zeros when the model has no other terms), else ``None``."""
        _, p = self.probe(np.zeros(self.shape))
        k = self.kind
        if k == "normal":
            return [_scalar(p[1], "sigma")], np.broadcast_to(np.asarray(p[0], float), self.shape)
        if k == "bernoulli_probit":
            return [], np.broadcast_to(ndtri(np.clip(p[0], 1e-300, 1 - 1e-16)), self.shape)
        if k == "bernoulli_logit":
            pc = np.clip(p[0], 1e-300, 1 - 1e-16)
            return [], np.broadcast_to(np.log(pc) - np.log1p(-pc), self.shape)
        if k == "poisson":
            return [], np.broadcast_to(np.log(p[0]), self.shape)
        if k == "negbin":
            return [_scalar(p[1], "alpha")], np.broadcast_to(np.log(p[0]), self.shape)
        if k == "negbin_np":  # (n, p) of the underlying nbinom variable: alpha = n, mean = n (1 - p) / p
            return [_scalar(p[0], "alpha")], np.broadcast_to(np.log(_nbinom_mean(p)), self.shape)
        if k == "student_t":  # [P] dist params (nu, mu, sigma); the kernel family takes (sigma, nu)
            return [_scalar(p[2], "sigma"), _scalar(p[0], "nu")], np.broadcast_to(np.asarray(p[1], float), self.shape)
        if k == "asymmetric_laplace":  # [P] dist params (b, kappa, mu), see _ald_params
            return _ald_params(p[0], p[1]), np.broadcast_to(np.asarray(p[2], float), self.shape)
        if k in ("gamma_scale", "gamma_rate"):
            return [_scalar(p[0], "alpha")], np.broadcast_to(np.log(_gamma_mean(p, k)), self.shape)
        return [], None  # categorical / mean-scale: no free parameters, no offsets


def _ald_params(b, kappa):
    """PyMC's AsymmetricLaplace(b, kappa, mu) has log-density -b kappa (y - mu) above mu and -(b / kappa) (mu - y)
    below; the kernel family is -rho_q((y - mu) / s) = -(q / s)(y - mu) above, -((1 - q) / s)(mu - y) below.
    Hence q = kappa^2 / (1 + kappa^2) (PyMC's own `q` argument: kappa = sqrt(q / (1 - q))) and
    s = sqrt(q (1 - q)) / b.  Returned in the kernel's order (s, q)."""
    kappa, b = _scalar(kappa, "kappa"), _scalar(b, "b")
    q = kappa * kappa / (1.0 + kappa * kappa)
    return [float(np.sqrt(q * (1.0 - q)) / b), float(q)]


def _nbinom_mean(p):
    n_, pr_ = np.asarray(p[0], float), np.asarray(p[1], float)
    return n_ * (1.0 - pr_) / pr_


def _gamma_mean(p, kind):
    a, second = np.asarray(p[0], float), np.asarray(p[1], float)
    return a * second if kind == "gamma_scale" else a / second


def _scalar(a, what):
    a = np.asarray(a, float)
    if a.size > 1 and not np.allclose(a, a.flat[0], rtol=1e-12, atol=0):
        raise NotImplementedError(f"{what} varies across observations: not in the closed likelihood family")
    return float(a.flat[0])


def _close(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return a.shape == b.shape and np.allclose(a, b, rtol=1e-7, atol=_TOL)


def identify(probe, shape, seed=0) -> Binding:
    """Decide the likelihood family from numeric probes.

    ``probe(x) -> (dist_name, [parameter arrays])`` with ``x`` shaped like the BART variable (``(n,)`` or
    ``(K, n)``).  Raises ``NotImplementedError`` (naming what it saw) outside the closed family."""
    rng = np.random.default_rng(seed)
    shape = tuple(shape)
    x0, xr = np.zeros(shape), rng.normal(0.0, 0.7, size=shape)
    name, p0 = probe(x0)
    _, pr = probe(xr)
    name = str(name).lower()
    if name in ("normal", "normal_rv"):
        if len(shape) == 1:
            if not _close(np.asarray(pr[0]) - np.asarray(p0[0]), xr):
                raise NotImplementedError("Normal likelihood whose mean is not (BART + other terms)")
            if not _close(pr[1], p0[1]):
                raise NotImplementedError("Normal likelihood whose sigma depends on the BART variable")
            return Binding(NormalLikelihood(_scalar(p0[1], "sigma")), probe, shape, "normal")
        if shape[0] == 2 and _close(pr[0], xr[0]) and _close(pr[1], np.abs(xr[1])):
            return Binding(NormalMeanScaleLikelihood(), probe, shape, "meanscale")
        raise NotImplementedError("multi-output Normal likelihood other than Normal(BART[0], |BART[1]|)")
    if name in ("bernoulli", "bernoulli_rv"):
        p_0, p_r = np.asarray(p0[0], float), np.asarray(pr[0], float)
        o = ndtri(np.clip(p_0, 1e-300, 1 - 1e-16))
        if _close(p_r, ndtr(xr + o)):
            return Binding(BernoulliLikelihood("probit"), probe, shape, "bernoulli_probit")
        o = np.log(p_0) - np.log1p(-p_0)
        if _close(p_r, expit(xr + o)):
            return Binding(BernoulliLikelihood("logit"), probe, shape, "bernoulli_logit")
        raise NotImplementedError("Bernoulli likelihood whose link is neither probit nor logit")
    if name in ("categorical", "categorical_rv"):
        K = shape[0]
        pm_ = np.asarray(pr[0], float)
        if pm_.shape == (shape[1], K):  # (n, K): the reference model transposes (tests/test_bart.py:156)
            pm_ = pm_.T
        e = np.exp(xr - xr.max(axis=0))
        if pm_.shape == shape and _close(pm_, e / e.sum(axis=0)):
            return Binding(CategoricalLikelihood(K), probe, shape, "categorical")
        raise NotImplementedError("Categorical likelihood whose probabilities are not softmax(BART)")
    if name in ("poisson", "poisson_rv"):
        if _close(np.log(pr[0]) - np.log(p0[0]), xr):
            return Binding(PoissonLikelihood(), probe, shape, "poisson")
        raise NotImplementedError("Poisson likelihood whose rate is not exp(BART + other terms)")
    if name in ("negative_binomial", "negativebinomial", "nbinom", "negative_binomial_rv"):
        if _close(np.log(pr[0]) - np.log(p0[0]), xr) and _close(pr[1], p0[1]):
            return Binding(NegativeBinomialLikelihood(_scalar(p0[1], "alpha")), probe, shape, "negbin")
        # [P] PyMC hands its (mu, alpha) to the nbinom variable as n = alpha, p = alpha / (mu + alpha)
        if (_close(pr[0], p0[0]) and np.all((np.asarray(p0[1]) > 0) & (np.asarray(p0[1]) < 1))
                and np.all((np.asarray(pr[1]) > 0) & (np.asarray(pr[1]) < 1))
                and _close(np.log(_nbinom_mean(pr)) - np.log(_nbinom_mean(p0)), xr)):
            return Binding(NegativeBinomialLikelihood(_scalar(p0[0], "alpha")), probe, shape, "negbin_np")
        raise NotImplementedError("NegativeBinomial likelihood outside mu = exp(BART + other terms), alpha free")
    if name in ("studentt", "student_t", "studentt_rv", "t"):  # [P] (nu, mu, sigma)
        if len(p0) == 3 and _close(np.asarray(pr[1]) - np.asarray(p0[1]), xr) and _close(pr[0], p0[0]) and _close(pr[2], p0[2]):
            return Binding(StudentTLikelihood(_scalar(p0[0], "nu"), _scalar(p0[2], "sigma")), probe, shape, "student_t")
        raise NotImplementedError("StudentT likelihood outside mu = BART + other terms with nu, sigma free of BART")
    if name in ("asymmetriclaplace", "asymmetric_laplace", "asymmetriclaplace_rv"):  # [P] (b, kappa, mu)
        if len(p0) == 3 and _close(np.asarray(pr[2]) - np.asarray(p0[2]), xr) and _close(pr[0], p0[0]) and _close(pr[1], p0[1]):
            s_, q_ = _ald_params(p0[0], p0[1])
            return Binding(AsymmetricLaplaceLikelihood(q=q_, b=s_), probe, shape, "asymmetric_laplace")
        raise NotImplementedError("AsymmetricLaplace likelihood outside mu = BART + other terms with b, kappa free of BART")
    if name in ("gamma", "gamma_rv"):
        # shape alpha free of BART, mean = exp(BART + other terms); the second parameter is a scale in recent
        # PyTensor (alpha, 1 / beta) and a rate in older releases (alpha, beta): whichever makes the mean log-linear
        if len(p0) == 2 and _close(pr[0], p0[0]):
            for kind in ("gamma_scale", "gamma_rate"):
                m0, mr = _gamma_mean(p0, kind), _gamma_mean(pr, kind)
                if np.all(m0 > 0) and np.all(mr > 0) and _close(np.log(mr) - np.log(m0), xr):
                    return Binding(GammaLikelihood(_scalar(p0[0], "alpha")), probe, shape, kind)
        raise NotImplementedError("Gamma likelihood outside alpha free of BART, mean = exp(BART + other terms)")
    raise NotImplementedError(
        f"observed distribution {name!r} is not in this sampler's closed likelihood family (Normal, "
        "Bernoulli probit/logit, Categorical softmax, Normal mean/scale, Poisson, NegativeBinomial, StudentT, "
        "AsymmetricLaplace, Gamma with a log link); pass likelihood= explicitly if it is one of them in disguise")


class FullVectorLogp:
    """A callback for family "callback" built from a function of the WHOLE BART vector:
    ``full_logp(bart_value) -> per-observation log-likelihood`` (what a compiled PyMC model gives: the
    observed variable's elementwise logp with every other variable at its current shared value).

    The sampler asks for (row, y, mu) triples, one particle after the other with ascending rows inside a
    particle; each particle's rows are scattered into a copy of the current BART value, the model function
    is evaluated on that vector -- exactly what upstream's PGBART does per particle -- and the requested
    rows are read back.  Rows outside the particle's leaf keep the base value; their results are ignored."""

    def __init__(self, full_logp, base_value):
        self.full_logp = full_logp
        self.base = np.array(base_value, np.float64).ravel()

    def set_base(self, value):
        self.base = np.array(value, np.float64).ravel()

    def __call__(self, y, mu, rows):
        out = np.empty(mu.size)
        cuts = np.flatnonzero(np.diff(rows) <= 0) + 1          # a row index that does not increase: next particle
        for seg in np.split(np.arange(mu.size), cuts):
            if seg.size == 0:
                continue
            full = self.base.copy()
            full[rows[seg]] = mu[seg]
            out[seg] = np.asarray(self.full_logp(full), np.float64).ravel()[rows[seg]]
        return out


class CallbackBinding:
    """The fallback binding: any single-output observed distribution, through the model's own logp."""

    has_offset = False
    kind = "callback"

    def __init__(self, likelihood, shape):
        self.likelihood = likelihood
        self.shape = shape

    def current(self):
        return [], None


def bind_model(vars, model=None, initial_point=None, compile_kwargs=None):  # noqa: A002  [P]
    """Everything ``PGBART.__init__`` takes from a PyMC model: the BART value variable and its op, the
    shared replacements ``ArrayStepShared`` keeps current, the observed response, and ``probe``.

    [P] Written against PyMC >= 5 / PyTensor from memory of the public API (``modelcontext``,
    ``make_shared_replacements``, ``join_nonshared_inputs``, ``model.replace_rvs_by_values``), the
    calls upstream's PGBART made to compile its log-likelihood.  PyMC is not installable on the build
    box; ``tests/test_pymc_bind_model_double.py`` executes this function end to end against a double of
    exactly these calls (closures for graphs) -- it proves the glue, not the real API's shapes."""
    from pymc.model import modelcontext
    from pymc.pytensorf import inputvars, join_nonshared_inputs, make_shared_replacements

    try:  # PyMC >= 5.22 renamed compile_pymc to compile (the old name is deprecated, then removed)
        from pymc.pytensorf import compile as compile_pymc
    except ImportError:
        from pymc.pytensorf import compile_pymc
    try:  # newer PyTensor moved the graph walkers
        from pytensor.graph.traversal import ancestors
    except ImportError:
        from pytensor.graph.basic import ancestors

    model = modelcontext(model)
    if initial_point is None:
        initial_point = model.initial_point()
    if vars is None:
        vars = model.value_vars  # noqa: A001
    else:
        vars = inputvars([model.rvs_to_values.get(v, v) for v in vars])  # noqa: A001
    value_bart = vars[0]
    bart_rv = model.values_to_rvs[value_bart]
    shared = make_shared_replacements(initial_point, [value_bart], model)
    users = [rv for rv in model.observed_RVs if bart_rv in set(ancestors([rv]))]
    if len(users) != 1:
        raise NotImplementedError(f"the BART variable must feed exactly one observed variable, found {len(users)}")
    rv = users[0]
    op = rv.owner.op
    params = list(op.dist_params(rv.owner)) if hasattr(op, "dist_params") else list(rv.owner.inputs[2:])
    outs = model.replace_rvs_by_values(params)
    out_list, inarray0 = join_nonshared_inputs(initial_point, outs, [value_bart], shared)
    fn = compile_pymc([inarray0], out_list, **(compile_kwargs or {}))
    fn.trust_input = True
    shape = tuple(np.shape(initial_point[value_bart.name]))
    dist_name = getattr(op, "name", type(op).__name__)

    def probe(x):
        return dist_name, [np.asarray(o) for o in fn(np.asarray(x, dtype=inarray0.dtype).ravel())]

    y_obs = np.asarray(model.rvs_to_values[rv].data if hasattr(model.rvs_to_values[rv], "data")
                       else model.rvs_to_values[rv].eval(), dtype=np.float64)
    try:
        binding = identify(probe, shape)
    except NotImplementedError:
        if len(shape) != 1:
            raise
        # outside the closed family: the model's own elementwise logp of the observed variable, as a
        # function of the whole BART vector (slow host-callback path, CallbackLikelihood)
        logp_el = model.logp(vars=[rv], sum=False)[0]
        lp_list, lp_in = join_nonshared_inputs(initial_point, [logp_el], [value_bart], shared)
        lp_fn = compile_pymc([lp_in], lp_list[0], **(compile_kwargs or {}))
        lp_fn.trust_input = True
        full = FullVectorLogp(lambda v: lp_fn(np.asarray(v, dtype=lp_in.dtype)), initial_point[value_bart.name])
        binding = CallbackBinding(CallbackLikelihood(full), shape)
    return {"value_var": value_bart, "op": bart_rv.owner.op, "shared": shared, "observed": y_obs,
            "binding": binding, "initial_point": initial_point}
