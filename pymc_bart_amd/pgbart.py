"""``PGBART`` -- the particle-Gibbs step method, drop-in for ``bartrs.PGBART``.

Reference boundary (the only direct use in the reference tree): ``tests/test_bart.py:231-235``
``PGBART([mu1], num_particles=5)`` passed to ``pm.sample(step=[...])``; auto-assignment relies
on ``competence`` (``pymc_bart/__init__.py:15-18``).  Stats wire format: ``tests/test_bart.py:59``
and ``utils.py:1387-1398``; tree hand-off: ``bart.py:134-135`` -> ``utils.py:124-127``.

The class works duck-typed without PyMC (PyMC is not installed on the build box):
``vars`` entries may be PyMC BART random variables (``var.owner.op`` carries the attributes set at
``bart.py:141-158``) or any object with those attributes, e.g. :class:`BARTOp`.  All sampling
work happens in ``libpgbart_hip.so``; this file only marshals.
"""

from __future__ import annotations

import numpy as np

from . import _abi
from .sampler import PyBartSettings, PySampler, jitter_duplicated
from .utils import _encode_vi

try:  # pragma: no cover - PyMC is optional
    from pymc.step_methods.arraystep import ArrayStepShared as _Base
    from pymc.step_methods.compound import Competence as _Competence

    _HAVE_PYMC = True
except Exception:  # noqa: BLE001
    _Base = object
    _Competence = None
    _HAVE_PYMC = False


class BARTOp:
    """Stand-in for the per-instance ``BART_<name>`` op of the reference (``bart.py:141-158``):
    a mailbox of settings plus the ``all_trees`` history list."""

    def __init__(self, X, Y, m=50, alpha=0.95, beta=2.0, response="constant", split_rules=None,
                 split_prior=None, name="mu", all_trees=None):
        if response not in ("constant", "linear", "mix"):
            raise ValueError("response must be 'constant', 'linear' or 'mix'")
        if response != "constant":  # same caveat as the reference (bart.py:128-132)
            import warnings

            warnings.warn(f"response={response!r} is experimental (upstream flags it the same way); "
                          "check the fit before relying on it", stacklevel=2)
        self.name = name
        self.X = np.asarray(X, dtype=float)
        self.Y = np.asarray(Y, dtype=float)
        self.m = int(m)
        self.alpha = float(alpha)
        self.beta = float(beta)
        self.response = response
        self.split_rules = split_rules
        self.split_prior = np.array([]) if split_prior is None else np.asarray(split_prior)
        self.all_trees = [] if all_trees is None else all_trees
        self.initval = float(self.Y.mean())


def _from_point(value, point):
    """A likelihood parameter: a number, a shared variable (``get_value()``: what ``ArrayStepShared.step``
    keeps current under PyMC), a callable, or the name of an entry of ``point``."""
    if isinstance(value, str):
        if point is None or value not in point:
            raise KeyError(f"point has no value for {value!r}; under PyMC bind the parameter to a shared "
                           "variable (PGBART(..., shared={name: var})) or let PGBART derive it from the model")
        value = point[value]
    elif hasattr(value, "get_value"):
        value = value.get_value()
    elif callable(value):
        value = value()
    return float(np.asarray(value))


class NormalLikelihood:
    """``y ~ Normal(mu = BART, sigma)``; ``sigma`` is read from the point by name or fixed."""

    family = "normal"

    def __init__(self, sigma=1.0):
        self.sigma = sigma

    def params(self, point=None):
        return [_from_point(self.sigma, point)]


class BernoulliLikelihood:
    """``y ~ Bernoulli(link^-1(BART))`` with ``link`` "probit" (cfg4 of BASELINE.json) or "logit"."""

    def __init__(self, link="probit"):
        if link not in ("probit", "logit"):
            raise ValueError("link must be 'probit' or 'logit'")
        self.family = "bernoulli_" + link

    def params(self, point=None):
        return []


class PoissonLikelihood:
    """``y ~ Poisson(exp(BART))`` -- the count model of the PyMC-BART documentation (BART is built on
    ``Y = log(counts)``; pass the counts as ``observed=`` to the step method)."""

    family = "poisson_log"

    def params(self, point=None):
        return []


class NegativeBinomialLikelihood:
    """``y ~ NegativeBinomial(mu = exp(BART), alpha)``; ``alpha`` is read from the point by name or
    fixed."""

    family = "negbin_log"

    def __init__(self, alpha=1.0):
        self.alpha = alpha

    def params(self, point=None):
        return [_from_point(self.alpha, point)]


class AsymmetricLaplaceLikelihood:
    """``y ~ AsymmetricLaplace(b, q, mu = BART)``: BART as the ``q``-quantile of y (the quantile
    regression example of the PyMC-BART documentation).  ``b`` by name from the point or fixed."""

    family = "asymmetric_laplace"

    def __init__(self, q=0.5, b=1.0):
        self.q, self.b = q, b

    def params(self, point=None):
        return [_from_point(self.b, point), _from_point(self.q, point)]


class GammaLikelihood:
    """``y ~ Gamma(alpha, mean = exp(BART))`` for positive responses (BART built on ``log(y)``)."""

    family = "gamma_log"

    def __init__(self, alpha=1.0):
        self.alpha = alpha

    def params(self, point=None):
        return [_from_point(self.alpha, point)]


class StudentTLikelihood:
    """``y ~ StudentT(nu, mu = BART, sigma)`` -- outlier-robust regression."""

    family = "student_t"

    def __init__(self, nu=4.0, sigma=1.0):
        self.nu, self.sigma = nu, sigma

    def params(self, point=None):
        return [_from_point(self.sigma, point), _from_point(self.nu, point)]


class CallbackLikelihood:
    """Any per-observation likelihood: ``logp(y, mu) -> log p(y_i | mu_i)`` evaluated on the HOST (NumPy
    arrays in and out, elementwise).  This is the slow fallback for models outside the closed family --
    upstream evaluates the model's ``datalogp`` through PyTensor for every particle (``SURVEY.md`` section 7);
    here the GPU still does the tree work, but every SMC round costs a device round trip and a Python call,
    so expect it to run two orders of magnitude below the built-in families.  ``params`` may return scalars
    the callable closes over; they are not sent to the device."""

    family = "callback"

    def __init__(self, logp):
        if not callable(logp):
            raise TypeError("logp must be callable: logp(y, mu) -> array of per-row log-likelihoods")
        self.logp = logp

    def params(self, point=None):
        return []


class CategoricalLikelihood:
    """``y ~ Categorical(softmax(BART[0..K-1]))`` -- K-vector leaves sharing one tree structure
    (reference ``tests/test_bart.py:140-164``: ``shape=(3, 9)``; cfg5 of BASELINE.json)."""

    family = "categorical"

    def __init__(self, n_outputs: int):
        if not 2 <= int(n_outputs) <= _abi.MAX_OUTPUTS:
            raise ValueError(f"n_outputs must be in [2, {_abi.MAX_OUTPUTS}]")
        self.n_outputs = int(n_outputs)

    def params(self, point=None):
        return []


class NormalMeanScaleLikelihood:
    """``y ~ Normal(BART[0], |BART[1]|)`` -- the ``shape=(2, n)`` model of reference
    ``tests/test_bart.py:107-123``."""

    family = "normal_meanscale"
    n_outputs = 2

    def params(self, point=None):
        return []


def _op_of(var):
    owner = getattr(var, "owner", None)
    return owner.op if owner is not None and hasattr(owner, "op") else var


# serialises the append + index pair of PGBART._publish (once per chain; chains run as threads of one
# process in chains.sample_chains and share the op)
_PUBLISH_LOCK = __import__("threading").Lock()


def _managed_list_beside(proxy):
    """A new, empty managed list on the manager server that serves ``proxy``.

    In the process that owns the manager this is ``manager.list()``.  A PyMC worker process only has the
    proxy it was given (``_manager`` is ``None`` after a fork or an unpickle), so the list is created the
    way ``BaseManager._create`` does it: a ``create`` request for the registered type ``"list"`` on the
    proxy's server, a proxy for the returned id, and the server-side creation reference dropped.  Nested
    managed lists are supported by ``multiprocessing.managers`` (the server keeps an object alive while
    another managed object refers to it, also after the creating process has gone).  Raises for anything
    that is not a ``multiprocessing`` list proxy."""
    from multiprocessing import managers

    mgr = getattr(proxy, "_manager", None)
    if mgr is not None:
        return mgr.list()
    token, authkey, serializer = proxy._token, proxy._authkey, proxy._serializer
    client = managers.listener_client[serializer][1]

    def request(method, args):
        conn = client(token.address, authkey=authkey)
        try:
            return managers.dispatch(conn, None, method, args)
        finally:
            conn.close()

    ident, exposed = request("create", ("list",))
    new = managers.ListProxy(managers.Token("list", token.address, ident), serializer, authkey=authkey,
                             exposed=exposed)
    request("decref", (ident,))  # the proxy holds its own reference now
    return new


def _pick_device():
    """A step method unpickled in a worker process (PyMC runs chains as processes and is not told which
    chain it carries) picks its GPU: LOCAL_RANK (torch.distributed launchers), PGBART_DEVICE, else the
    worker's ordinal in its pool, modulo the visible devices.  No-op without a GPU."""
    import os

    try:
        import torch

        ndev = torch.cuda.device_count()
        if ndev < 1:
            return None
        if "LOCAL_RANK" in os.environ:
            idx = int(os.environ["LOCAL_RANK"])
        elif "PGBART_DEVICE" in os.environ:
            idx = int(os.environ["PGBART_DEVICE"])
        else:
            import multiprocessing as mp

            ident = getattr(mp.current_process(), "_identity", ())
            idx = (ident[0] - 1) if ident else torch.cuda.current_device()
        idx %= ndev
        torch.cuda.set_device(idx)
        return idx
    except Exception:  # noqa: BLE001 - never fatal: the current device stays
        return None


def _set_on_op(op, name, value):
    """The op is a mailbox: reference ``utils.py:125`` reads ``op.n_outputs``, which nothing in ``bart.py`` sets --
    the step method does.  WHERE matters: the reference's reader is ``BARTRV.rng_fn``, a *classmethod*
    (``bart.py:47-49``) that hands ``cls`` -- the per-variable class ``BART_<name>`` built at ``bart.py:141-158``
    with ``all_trees``, ``X``, ``m``, ... as CLASS attributes -- to ``_get_posterior_sampler(cls)`` (``:66``), while
    the step method holds ``rv.owner.op``, an *instance* of that class.  An attribute set on the instance is
    invisible from ``cls``; so when the op's class is such a per-variable class (it carries its own
    ``all_trees``), the attribute goes on the class as well.  A shared class (this package's :class:`BARTOp`)
    is left alone: its instances are separate variables."""
    setattr(op, name, value)
    if isinstance(op, type):
        return
    cls = type(op)
    if "all_trees" in vars(cls):
        try:
            setattr(cls, name, value)
        except (AttributeError, TypeError):  # a class that refuses attributes: the instance has it
            pass


def _compat_bits(semantics) -> int:
    """``semantics``: ``None`` (or ``PGBART_SEMANTICS`` unset) = this sampler; ``"upstream"`` = both upstream-semantics
    switches of ``include/pgbart_spec.h`` (fresh particles at log-weight 0, empty right leaves of one-hot splits);
    an ``int`` = the ``PGB_COMPAT_*`` bits themselves.  The environment variable lets a model that builds its step
    method through ``pm.sample()`` -- which passes no such argument -- pick the mode
    (``tools/pymc_selfcheck.py`` runs both and reports which one ``bartrs`` agrees with)."""
    import os

    if semantics is None:
        semantics = os.environ.get("PGBART_SEMANTICS") or 0
    if isinstance(semantics, str):
        if semantics.lstrip("-").isdigit():
            return int(semantics)
        try:
            return {"default": 0, "native": 0, "upstream": 3}[semantics.lower()]
        except KeyError:
            raise ValueError(f"semantics must be 'default', 'upstream' or the PGB_COMPAT_* bits, got {semantics!r}") from None
    return int(semantics)


def _eval(x):
    return x.eval() if hasattr(x, "eval") and not isinstance(x, np.ndarray) else x


class PGBART(_Base):
    """Particle Gibbs BART sampling step.

    Parameters
    ----------
    vars : list
        One BART variable (PyMC RV or :class:`BARTOp`).
    num_particles : int
        Number of particles, including the reference particle (upstream default 10).
    batch : tuple
        Fraction (or count) of the ``m`` trees re-sampled per step during (tuning, draws).
    """

    name = "pgbart"
    default_blocked = False
    generates_stats = True
    stats_dtypes_shapes = {"variable_inclusion": (object, []), "tune": (bool, [])}
    # older PyMC versions read this attribute instead
    stats_dtypes = [{"variable_inclusion": object, "tune": bool}]

    def __init__(self, vars=None, num_particles=10, batch=(0.1, 0.1), model=None,  # noqa: A002
                 initial_point=None, compile_kwargs=None, *, likelihood=None, observed=None, shared=None,
                 random_seed=None, chain=0, backend=None, range_exp=None, semantics=None):
        self._binding = None
        duck = vars is not None and len(vars) == 1 and getattr(vars[0], "owner", None) is None
        if _HAVE_PYMC and not duck and likelihood is None:
            # reference call convention (tests/test_bart.py:231-235): PGBART([rv], num_particles=...) inside
            # a model context; family, observed response and shared variables come from the model
            from ._pymc_bridge import bind_model

            bound = bind_model(vars, model, initial_point, compile_kwargs)
            vars = [bound["value_var"]]  # noqa: A001
            self._binding = bound["binding"]
            likelihood, observed, shared = self._binding.likelihood, bound["observed"], bound["shared"]
            op = bound["op"]
        else:
            if vars is None or len(vars) != 1:
                raise ValueError("PGBART samples exactly one BART variable per step method")
            op = _op_of(vars[0])
        self._var = vars[0]
        self.bart = op
        X = np.asarray(_eval(op.X), dtype=np.float64)
        Y = np.asarray(_eval(op.Y), dtype=np.float64)
        if X.ndim != 2:
            raise ValueError("X must be 2-dimensional")
        self.num_observations, self.num_variates = X.shape
        self.m = int(op.m)
        self.response = getattr(op, "response", "constant")
        if self.response not in ("constant", "linear", "mix"):
            raise ValueError("response must be 'constant', 'linear' or 'mix'")
        split_prior = np.asarray(getattr(op, "split_prior", np.array([])), dtype=np.float64)
        if split_prior.size == 0:  # bart.py:139 -> all covariates equally likely
            split_prior = np.ones(self.num_variates)
        rules = getattr(op, "split_rules", None)
        if not rules:
            rule_ids = np.zeros(self.num_variates, np.int32)
        else:
            try:
                rule_ids = np.array(
                    [_abi.RULES[r if isinstance(r, str) else getattr(r, "__name__", str(r))]
                     for r in rules], np.int32)
            except KeyError as e:
                raise NotImplementedError(f"split rule {e} is not implemented") from e
        seed = int(np.random.SeedSequence(random_seed).generate_state(1, np.uint64)[0]) \
            if random_seed is None else int(random_seed)
        seed = (seed + 0x9E3779B97F4A7C15 * int(chain)) & 0xFFFFFFFFFFFFFFFF
        # [U] whole-number continuous columns are jittered (CHANGELOG.md:329-332)
        jrng = np.random.default_rng(seed)
        X = X.copy()
        for j in range(self.num_variates):
            if rule_ids[j] == _abi.RULE_CONTINUOUS:
                X[:, j] = jitter_duplicated(X[:, j], jrng)
            elif rule_ids[j] == _abi.RULE_SUBSET:
                col = X[:, j][~np.isnan(X[:, j])]
                if col.size and (np.any(col != np.floor(col)) or col.min() < 0
                                 or col.max() >= _abi.SUBSET_BITS):
                    raise ValueError(
                        f"SubsetSplit column {j}: categories must be integer codes in "
                        f"[0, {_abi.SUBSET_BITS}) (NaN = missing)")
        self.likelihood = likelihood if likelihood is not None else NormalLikelihood(1.0)
        y_obs = Y if observed is None else np.asarray(observed, np.float64)
        self._y_obs = np.array(y_obs, np.float64, copy=True)
        n_outputs = int(getattr(self.likelihood, "n_outputs", 1))
        self.settings = PyBartSettings.from_data(
            X, Y, m=self.m, num_particles=num_particles, n_outputs=n_outputs,
            family=self.likelihood.family, alpha=float(op.alpha), beta=float(op.beta),
            batch=batch, seed=seed, response=self.response, y_obs=y_obs, range_exp=range_exp,
            compat=_compat_bits(semantics),
        )
        self._X, self._rule_ids, self._split_prior = X, rule_ids, split_prior
        self._backend_arg = backend  # (not pickled: a worker process builds on its own default backend)
        self._base_seed = seed       # before any per-chain re-keying (set_rng / first astep in a PyMC worker)
        self._keyed = random_seed is not None and not (_HAVE_PYMC and not duck)  # explicit seed, no PyMC: final
        self._stepped = False
        # The native sampler (design matrix uploaded and transposed in HBM) is built when the Philox key is FINAL:
        # at once for a keyed step method; under PyMC -- where every chain's copy takes its own key through
        # set_rng or at its first astep -- on first use, so that re-keying never builds a second sampler and
        # uploads X again (several GB at the sizes this backend is for; round-3 ADVICE).
        self._sampler = None
        if self._keyed:
            self._sampler = self._build_sampler()
        _set_on_op(op, "n_outputs", n_outputs)
        self.tune = True
        self._baseline = None
        self._batches = []
        self._registered = False
        self._published = 0      # batches already sent to the op's history list
        self._offset = None      # last offset applied (re-applied after unpickling)
        self.shape = (self.num_observations,) if n_outputs == 1 else (n_outputs, self.num_observations)
        if _Base is not object:
            # ArrayStepShared keeps `shared` current (step(point) copies the other variables' values into
            # them before astep(q) runs): likelihood parameters bound to those variables are read there
            super().__init__([self._var], dict(shared or {}))

    def _build_sampler(self):
        smp = PySampler(self.settings, self._X, self._y_obs, self._rule_ids, self._split_prior,
                        backend=self._backend_arg)
        if self.likelihood.family == "callback":
            smp.set_loglik_callback(self.likelihood.logp)
        return smp

    @property
    def sampler(self):
        if self._sampler is None:
            self._sampler = self._build_sampler()
            if getattr(self, "_offset", None) is not None:
                self._apply_offset(self._offset)
        return self._sampler

    @sampler.setter
    def sampler(self, value):
        self._sampler = value

    # -- pickling: PyMC sends the step method to its worker processes ----------
    def __getstate__(self):
        """Everything but the native handle; the chain itself travels as a checkpoint image, so
        a step method pickled mid-run resumes bit-identically in the process that unpickles it
        (on that process's current GPU)."""
        d = dict(self.__dict__)
        d.pop("_sampler", None)
        d.pop("_backend_arg", None)
        # A chain that has not stepped has no state beyond its settings: it travels WITHOUT an image and without a
        # native sampler ever being built here -- this is how PyMC sends the step method to its spawn / forkserver
        # workers, and building one only to photograph it cost an upload + transpose of X in the parent and a second
        # sampler in the worker once its key arrived (round-4 ADVICE).
        d["_checkpoint"] = self._sampler.checkpoint() if self._stepped and self._sampler is not None else None
        return d

    def __setstate__(self, d):
        blob = d.pop("_checkpoint")
        self.__dict__.update(d)
        self._backend_arg = None
        self._sampler = None
        self._device_index = _pick_device()  # (None without a GPU) -- what the launcher tests assert per rank
        if blob is None:  # un-stepped: the sampler is built on first use, when this copy's Philox key is final
            return
        self.sampler = self._build_sampler()
        self.sampler.restore(blob)
        if self._offset is not None:  # the image carries the chain, not the caller-owned response
            self._apply_offset(self._offset)

    # -- PyMC step-method surface ---------------------------------------------------
    @staticmethod
    def competence(var, has_grad=False):
        """IDEAL for BART variables (upstream: ``isinstance(var.owner.op, BARTRV)``)."""
        op = _op_of(var)
        is_bart = all(hasattr(op, a) for a in ("X", "Y", "m", "alpha", "beta", "all_trees"))
        if _Competence is not None:  # pragma: no cover
            return _Competence.IDEAL if is_bart else _Competence.INCOMPATIBLE
        return 3 if is_bart else 0

    def stop_tuning(self):
        self.tune = False

    # -- per-chain random streams ----------------------------------------------------
    # PyMC builds ONE step method and hands a copy of it to every chain (fork or pickle): the copies share the
    # seed they were built with, and the sampler's random numbers are addressed by (seed, iteration, ...) -- so
    # without the two hooks below every chain of `pm.sample(chains=4)` would draw the SAME forests.
    def set_rng(self, rng):
        """[P] PyMC >= 5.17 gives every chain's copy of a step method its own generator (`step.set_rng(rng)`) before
        the first draw: the chain's Philox key is derived from it (reproducible under `pm.sample(random_seed=...)`).
        Once the chain has stepped the key stays (re-keying would restart the forest)."""
        gen = np.random.default_rng(rng)
        self.rng = gen
        if not self._stepped:
            self._rekey(int(gen.integers(0, 2**63 - 1)))
            self._keyed = True

    def _rekey(self, seed: int) -> None:
        """The same sampler on another Philox key (before its first step: there is no state to lose)."""
        self.settings.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self._sampler = None  # (somebody looked at the sampler before the key was final: the next use rebuilds it)

    def _key_for_this_process(self) -> None:
        """Older PyMC has no `set_rng`: it seeds NumPy's global generator per chain (`np.random.seed(chain_seed)`)
        in the process that runs the chain.  At the first astep of a chain that nobody keyed, the key is therefore
        mixed with the STATE of that generator (read, not advanced) and with the worker's ordinal in its pool --
        different in every chain, reproducible under `pm.sample(random_seed=...)`."""
        import multiprocessing as mp

        ident = getattr(mp.current_process(), "_identity", ()) or (0,)
        # The global generator's STATE is read, not advanced (round-3 ADVICE: drawing from it changed what the
        # user's own code draws next): two chains that one worker runs one after the other differ only in the
        # seed pm.sample gave that generator, so it has to enter the key -- next to the explicit random_seed of
        # the step method, which is always mixed in.
        st = np.random.get_state()
        glob = [int(x) for x in np.asarray(st[1][:4], np.uint64)] + [int(st[2])]
        mix = np.random.SeedSequence([self._base_seed & 0xFFFFFFFF, self._base_seed >> 32, *glob,
                                      *[int(i) for i in ident]])
        self._rekey(int(mix.generate_state(1, np.uint64)[0]))
        self._keyed = True

    def astep(self, _q=None, point=None, offset=None):
        """Re-sample the next batch of trees; returns ``(sum_trees, [stats])``.

        ``offset``: contribution of the *other* additive terms of the model at the current point
        (a second BART variable, reference ``tests/test_bart.py:211-241``; a log-exposure of a count
        model): a Normal model fits ``observed - offset``, the per-row families add it to the linear
        predictor.
        """
        if not self._stepped:
            if not self._keyed:
                self._key_for_this_process()
            self._stepped = True
        if self._binding is not None:  # PyMC model: parameters and offset at the shared values
            params, model_offset = self._binding.current()
            if model_offset is not None:
                offset = model_offset
        else:
            params = self.likelihood.params(point)
        if offset is not None and not self._same_offset(offset):
            self._apply_offset(offset)
        self.sampler.set_likelihood(params)
        if not self.tune and self._baseline is None:
            # first draw: freeze the forest the per-draw batches are deltas of (utils.py:124-127)
            self._baseline = self.sampler.export_trees(1)
        sum_trees, vi = self.sampler.step(self.tune)
        if not self.tune:
            self._batches.append(self.sampler.export_trees(0))
            self._publish()
        stats = {"variable_inclusion": _encode_vi(vi), "tune": self.tune}
        return sum_trees, [stats]

    def _same_offset(self, offset) -> bool:
        """True when ``offset`` is what the device already holds (no offset yet = zeros): the model's
        other terms are reported on every step, but uploaded only when they moved."""
        offset = np.asarray(offset, np.float64)
        if self._offset is None:
            return offset.shape == self.shape and not offset.any()
        return offset.shape == self._offset.shape and np.array_equal(offset, self._offset)

    def _apply_offset(self, offset):
        offset = np.asarray(offset, np.float64)
        if self.likelihood.family == "normal":      # additive Normal model: fit what is left
            self.sampler.set_response(self._y_obs - offset)
        else:                                        # per-row families: offset of the linear predictor(s)
            if offset.shape != self.shape:
                raise ValueError(f"offset must have the BART variable's shape {self.shape}, got {offset.shape}")
            self.sampler.set_offset(offset)
        self._offset = np.array(offset, copy=True)

    if _Base is object:  # without PyMC: the part of ArrayStepShared.step this class needs

        def step(self, point):
            """Duck-typed ``step``: writes the new ``sum_trees`` into ``point[<bart name>]``."""
            sum_trees, stats = self.astep(None, point)
            out = dict(point)
            out[getattr(self.bart, "name", "mu")] = sum_trees
            return out, stats

    def _publish(self):
        """Hand this chain's history to the op (reference ``bart.py:134-135`` -> ``utils.py:124-127``):
        ONE ``(baseline_forest, batches)`` entry per chain, current after every draw, O(1) per draw.

        * A plain ``list`` (this package's :class:`BARTOp`) holds ``self._batches`` by reference.
        * The reference's op carries a ``multiprocessing.Manager().list()`` proxy, which pickles what it is
          given: there ``batches`` is a second managed list on the SAME manager server
          (:func:`_managed_list_beside`) and every draw appends its batch to it -- one small message per
          draw, nothing to flush when the worker process ends (PyMC gives a step method no end-of-sampling
          hook), and the parent sees exactly as many batches as the trace has draws.  A batch that is on the
          managed list is dropped here: the worker's memory does not grow with the draws (:attr:`history` reads
          the managed list back).
        * Any other list-like gets its entry re-assigned on every draw (always current, O(draws^2) bytes).  WHICH
          entry is this chain's is read off the list itself: the baseline forest carries an owner token
          (pid, Philox key, nonce) that survives pickling, and the entry is looked up by it -- ``len(trees) - 1``
          after the append is another chain's entry when two worker PROCESSES register at the same instant (the
          lock below only orders the threads of one process; round-5 VERDICT, smaller #9)."""
        trees = self.bart.all_trees
        if not self._registered:
            import os
            import uuid

            self._shared = None
            self._slot = None
            self._is_proxy = not isinstance(trees, list)
            self._token = f"{os.getpid()}:{self.settings.seed:016x}:{uuid.uuid4().hex[:12]}"
            if self._is_proxy:
                try:
                    self._shared = _managed_list_beside(trees)
                    self._shared.extend(self._batches)
                except Exception:  # noqa: BLE001 - not a manager proxy: fall back to re-assignment
                    self._shared = None
                try:
                    self._baseline.owner = self._token
                except AttributeError:  # (a baseline that takes no attributes: the position fallback below)
                    pass
            entry = (self._baseline, self._batches if self._shared is None else self._shared)
            with _PUBLISH_LOCK:
                trees.append(entry)
                if not self._is_proxy:
                    self._slot = len(trees) - 1  # a plain list: this process's threads only, ordered by the lock
                elif self._shared is None:
                    self._slot = self._find_slot(trees)
            self._registered = True
            self._published = len(self._batches)
            if self._shared is not None:
                self._sent = len(self._batches)
                del self._batches[:]
                self._published = 0
            return
        if self._shared is not None:
            for b in self._batches[self._published:]:
                self._shared.append(b)
            self._sent = getattr(self, "_sent", 0) + len(self._batches) - self._published
            del self._batches[:]  # on the managed list now
            self._published = 0
        elif self._is_proxy:
            self.flush_history()

    def _find_slot(self, trees) -> int:
        """Index of this chain's entry in a list-like history: the entry whose baseline forest carries this chain's
        owner token, searched from the end (entries are only ever appended, so an index stays valid)."""
        n = len(trees)
        for i in range(n - 1, -1, -1):
            if getattr(trees[i][0], "owner", None) == self._token:
                return i
        return n - 1  # (the container dropped the token: the position right after the append, as before)

    def flush_history(self):
        """Make the op's history entry current.  Every draw already does (see :meth:`_publish`); kept for
        callers of earlier versions and as the re-assignment of the list-like fallback."""
        if not self._registered or not getattr(self, "_is_proxy", False):
            return
        if self._published < len(self._batches):
            if getattr(self, "_shared", None) is not None:
                self._shared.extend(self._batches[self._published:])
                self._sent = getattr(self, "_sent", 0) + len(self._batches) - self._published
                del self._batches[:]
                self._published = 0
            else:
                if getattr(self, "_slot", None) is None:
                    self._slot = self._find_slot(self.bart.all_trees)
                self.bart.all_trees[self._slot] = (self._baseline, self._batches)
                self._published = len(self._batches)

    @property
    def history(self):
        """This chain's ``(baseline_forest, batches)`` as a plain list of batches -- read back from the managed list
        when the batches live there."""
        if getattr(self, "_registered", False) and getattr(self, "_shared", None) is not None:
            return self._baseline, list(self._shared[:]) + list(self._batches)
        return self._baseline, self._batches

    def reset_history(self):
        """Forget the tree history kept so far: the next draw freezes a new baseline forest and registers a new
        ``(baseline_forest, batches)`` entry on the op (the old entry stays where it is).  The chain itself is
        untouched.  For long-running callers that consume and drop histories in pieces (bench.py's end-of-run
        gather after thousands of timed draws); PyMC never calls it."""
        self._baseline = None
        self._batches = []
        self._registered = False
        self._published = 0
        self._shared = None
        self._sent = 0

    @property
    def counters(self) -> dict:
        return self.sampler.counters.as_dict()
