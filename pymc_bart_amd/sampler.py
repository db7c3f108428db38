"""``PySampler`` / ``PyBartSettings``: Python owners of a native sampler handle.

Counterparts of ``bartrs.bartrs.PySampler`` and ``PyBartSettings`` (reference
``pymc_bart/pymc_bart.py:2``).  :class:`PySampler` owns one ``pgb_handle`` bound to
one GPU and one HIP stream; every call is a thin ctypes hop into
``libpgbart_hip.so``.
"""

from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np

from . import _abi
from .trees import PackedTrees, TreeArrays  # noqa: F401


@dataclass
class Backend:
    """A loaded ABI library plus the memory holder its "device" pointers live in."""

    lib: _abi.PGBLibrary
    mem: object


_DEFAULT_BACKEND: Backend | None = None


def default_backend(device: int | None = None) -> Backend:
    """The HIP backend on the current GPU.  Raises if the extension or the GPU is
    missing -- by design there is no CPU fallback in the product path."""
    global _DEFAULT_BACKEND
    if _DEFAULT_BACKEND is None or device is not None:
        from ._device import TorchHipMemory

        lib = _abi.load_hip_library()
        be = Backend(lib=lib, mem=TorchHipMemory(device))
        if device is not None:
            return be
        _DEFAULT_BACKEND = be
    return _DEFAULT_BACKEND


def prior_leaf_table(alpha: float, beta: float) -> np.ndarray:
    """P(node at depth d stays a leaf) = 1 - alpha (1 + d)^-beta  (reference
    ``bart.py:107-109``).  As upstream, the table is cut once it reaches 0.9999:
    deeper nodes never split."""
    tab = np.ones(_abi.MAX_DEPTH, np.float64)
    for d in range(_abi.MAX_DEPTH):
        v = 1.0 - alpha * (1.0 + d) ** (-beta)
        if v >= 0.9999:
            break
        tab[d] = v
    return tab


def range_exponent(*arrays) -> int:
    """Fixed-point range: |sum_trees| and residuals stay below 2^e, with 8x headroom over the
    largest magnitude in ``arrays`` (the BART response ``Y`` and, when it differs, the observed
    response the likelihood sees)."""
    a = 1e-30
    for arr in arrays:
        arr = np.asarray(arr, np.float64)
        if arr.size:
            a = max(a, float(np.nanmax(np.abs(arr))))
    return int(math.ceil(math.log2(a))) + 3


def jitter_duplicated(col: np.ndarray, rng: np.random.Generator) -> np.ndarray:
    """Upstream pre-processing of whole-number continuous columns: every repeated
    non-zero value gets N(0, nanstd/12) added so that split values are distinct
    (reference ``CHANGELOG.md:329-332``, NaN-safe ``:305``).  Vectorised; the first
    occurrence of each value is kept."""
    finite = col[~np.isnan(col)]
    if finite.size == 0 or not np.all(np.mod(finite, 1) == 0):
        return col
    std = float(np.nanstd(col))
    out = col.copy()
    _, first = np.unique(col, return_index=True)
    dup = np.ones(col.shape[0], bool)
    dup[first] = False
    dup &= ~np.isnan(col) & (np.abs(col) > 0)
    out[dup] = col[dup] + rng.normal(0.0, std / 12.0, size=int(dup.sum()))
    return out


@dataclass
class PyBartSettings:
    """Sampler settings (counterpart of ``bartrs.bartrs.PyBartSettings``)."""

    n: int
    p: int
    m: int = 50
    num_particles: int = 10
    n_outputs: int = 1
    family: str = "normal"
    alpha: float = 0.95
    beta: float = 2.0
    batch: tuple = (0.1, 0.1)
    seed: int = 0
    init_sum: float = 0.0
    init_leaf: float = 0.0
    init_leaf_sd: float = 1.0
    range_exp: int = 4
    response: str = "constant"
    # Upstream-semantics switches (``include/pgbart_spec.h``, ``PGB_COMPAT_*``): 0 = this sampler; bit 0: a particle
    # that has not grown keeps log-weight 0 (upstream's ``ParticleTree.log_weight``); bit 1: a one-hot split grows
    # an empty right leaf.  ``"upstream"`` in :class:`PGBART` sets both.
    compat: int = 0
    prior_leaf: np.ndarray = field(default_factory=lambda: np.ones(_abi.MAX_DEPTH))

    @classmethod
    def from_data(cls, X, Y, m=50, num_particles=10, n_outputs=1, family="normal", alpha=0.95,
                  beta=2.0, batch=(0.1, 0.1), seed=0, response="constant", y_obs=None,
                  range_exp=None, compat=0) -> "PyBartSettings":
        """``y_obs``: the observed response when it is not ``Y`` itself (it enters the fixed-point
        range); ``range_exp``: override the range, e.g. for additive models whose offsets move the
        partial residuals far outside the data's own range."""
        Y = np.asarray(Y, np.float64)
        n, p = X.shape
        mean = float(Y.mean())
        is_binary = bool(np.all((Y == 0) | (Y == 1)))
        # [U] leaf_sd = 3/sqrt(m) for 0/1 responses, std(Y)/sqrt(m) otherwise
        leaf_sd = 3.0 / math.sqrt(m) if is_binary else float(Y.std()) / math.sqrt(m)
        rexp = range_exponent(Y) if y_obs is None else range_exponent(Y, y_obs)
        if range_exp is not None:
            rexp = int(range_exp)
        elif family != "normal":
            rexp = max(rexp, 10)  # latent link scale: |sum_trees| < 1024 (separable data drifts far)
        return cls(
            n=n, p=p, m=m, num_particles=num_particles, n_outputs=n_outputs, family=family,
            alpha=alpha, beta=beta, batch=tuple(batch), seed=int(seed), init_sum=mean,
            init_leaf=mean / m, init_leaf_sd=leaf_sd, range_exp=rexp, response=response, compat=int(compat),
            prior_leaf=prior_leaf_table(alpha, beta),
        )

    def batch_sizes(self) -> tuple[int, int]:
        """Trees re-sampled per step while (tuning, drawing).  Upstream's rule for the fractions it
        takes: ``max(1, int(m * b))`` -- so ``batch=(1.0, 1.0)`` means every tree, every step.  A
        Python ``int`` (not a float) is taken as a tree count, an extension upstream does not have."""
        out = []
        for b in self.batch:
            if isinstance(b, (int, np.integer)) and not isinstance(b, (bool, np.bool_)):
                out.append(max(1, min(int(b), self.m)))
            else:
                out.append(max(1, min(self.m, int(self.m * float(b)))))
        return out[0], out[1]

    def as_c(self) -> _abi.Settings:
        s = _abi.Settings()
        s.n, s.p, s.m = self.n, self.p, self.m
        s.num_particles = self.num_particles
        s.n_outputs = self.n_outputs
        s.family = _abi.FAMILIES[self.family]
        s.batch_tune, s.batch_draw = self.batch_sizes()
        s.range_exp = self.range_exp
        s.response = _abi.RESPONSES[self.response]
        s.compat = int(self.compat)
        s.seed = self.seed & 0xFFFFFFFFFFFFFFFF
        s.init_sum = self.init_sum
        s.init_leaf = self.init_leaf
        s.init_leaf_sd = self.init_leaf_sd
        for d in range(_abi.MAX_DEPTH):
            s.prior_leaf[d] = float(self.prior_leaf[d])
        return s


class PySampler:
    """One chain's native sampler state (counterpart of ``bartrs.bartrs.PySampler``)."""

    def __init__(self, settings: PyBartSettings, X: np.ndarray, y_obs: np.ndarray,
                 rules: np.ndarray, split_prior: np.ndarray, backend: Backend | None = None):
        self.backend = backend if backend is not None else default_backend()
        self.settings = settings
        if settings.num_particles > self.backend.lib.max_particles and self.backend.lib.backend_name == "hip-gfx950":
            # more than one particle per lane: the build of the same library that puts two on a lane (include/pgbart.h)
            self.backend = Backend(lib=_abi.load_hip_library(settings.num_particles), mem=self.backend.mem)
        lib, mem = self.backend.lib, self.backend.mem
        self._h = C.c_void_p()
        cs = settings.as_c()
        # (device backends: a stream object to keep alive; the CPU oracle has none)
        self._stream = mem.sampler_stream() if hasattr(mem, "sampler_stream") else None
        stream_ptr = int(self._stream.cuda_stream) if self._stream is not None else mem.stream_ptr
        lib.check(lib.lib.pgb_create(C.byref(cs), stream_ptr, C.byref(self._h)), "pgb_create")
        if hasattr(mem, "output_stream"):
            self._out_stream = mem.output_stream()
            lib.check(lib.lib.pgb_set_output_stream(self._h, int(self._out_stream.cuda_stream)), "pgb_set_output_stream")
        X = np.ascontiguousarray(X, dtype=np.float64)
        self._rules = np.ascontiguousarray(rules, dtype=np.int32)
        prior = np.ascontiguousarray(split_prior, dtype=np.float64)
        xd = mem.from_host(X)
        lib.check(
            lib.lib.pgb_set_data(self._h, mem.ptr(xd), X.shape[1], self._rules.ctypes.data,
                                 prior.ctypes.data),
            "pgb_set_data",
        )
        del xd  # the library keeps its own column-major copy
        self._y = mem.from_host(np.ascontiguousarray(y_obs, dtype=np.float64))
        lib.check(lib.lib.pgb_set_response(self._h, mem.ptr(self._y)), "pgb_set_response")
        self._out = mem.empty((settings.n_outputs * settings.n,), np.float64)
        self._vi = np.zeros(settings.p, np.int32)
        self._pack_buf = np.empty(64 << 10, np.uint8)  # packed tree records land here (grown on demand)
        self.counters = _abi.Counters()
        self._sat_seen = 0   # saturation events already reported
        self._steps = 0      # asteps issued (to name the one that overflowed)

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self.backend.lib.lib.pgb_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    # -- observed response (e.g. y minus the other additive terms of the model) --------
    def set_response(self, y_obs) -> None:
        lib, mem = self.backend.lib, self.backend.mem
        self._y = mem.from_host(np.ascontiguousarray(y_obs, dtype=np.float64))
        lib.check(lib.lib.pgb_set_response(self._h, mem.ptr(self._y)), "pgb_set_response")

    def set_offset(self, offset) -> None:
        """Per-row offset of the linear predictor (single-output per-row families): what the other
        additive terms of the model contribute at the current point; ``None`` resets it."""
        lib, mem = self.backend.lib, self.backend.mem
        if offset is None:
            lib.check(lib.lib.pgb_set_offset(self._h, None), "pgb_set_offset")
            return
        offset = np.ascontiguousarray(offset, dtype=np.float64)
        want = self.settings.n_outputs * self.settings.n  # the ABI takes no size: it reads exactly K*n doubles
        if offset.size != want:
            raise ValueError(f"offset must hold n_outputs * n = {want} values ([K][n]), got {offset.size}")
        self._off = mem.from_host(offset)
        lib.check(lib.lib.pgb_set_offset(self._h, mem.ptr(self._off)), "pgb_set_offset")

    # -- likelihood parameters at the current point -------------------------------
    def set_likelihood(self, params) -> None:
        key = tuple(params) if isinstance(params, (list, tuple)) else None
        if key is not None and key == getattr(self, "_lik_key", None):
            return  # unchanged since the last call (sigma fixed, parameter-free families): nothing to send
        self._lik_key = key
        a = np.ascontiguousarray(np.atleast_1d(np.asarray(params, np.float64)))
        keep = np.zeros(1) if a.size == 0 else a  # a valid pointer even for parameter-free families
        lib = self.backend.lib
        lib.check(lib.lib.pgb_set_likelihood(self._h, keep.ctypes.data, a.size), "pgb_set_likelihood")

    def set_loglik_callback(self, fn) -> None:
        """Family "callback": ``fn(y, mu)`` or ``fn(y, mu, rows)`` -> per-row log-likelihood (NumPy arrays in,
        array out): a function of each row's observed value, linear predictor and -- if it takes a third
        argument -- row index (per-row covariates of the likelihood).  The native library calls it once per
        SMC round with all the rows that round re-labelled, one particle after the other, ascending rows
        within a particle (``pgb_set_loglik_callback``)."""
        import inspect

        lib = self.backend.lib
        try:
            wants_rows = len(inspect.signature(fn).parameters) >= 3
        except (TypeError, ValueError):
            wants_rows = False

        def trampoline(_ctx, row_ptr, y_ptr, mu_ptr, n, out_ptr):
            try:
                n = int(n)
                y = np.ctypeslib.as_array(y_ptr, shape=(n,))
                mu = np.ctypeslib.as_array(mu_ptr, shape=(n,))
                val = fn(y, mu, np.ctypeslib.as_array(row_ptr, shape=(n,))) if wants_rows else fn(y, mu)
                np.ctypeslib.as_array(out_ptr, shape=(n,))[:] = np.asarray(val, dtype=np.float64).reshape(n)
                return 0
            except Exception as e:  # noqa: BLE001 - reported through the ABI's error code
                self._callback_error = e
                return 1

        self._callback_error = None
        self._callback = _abi.LOGLIK_FN(trampoline)  # keep it alive as long as the sampler
        lib.check(lib.lib.pgb_set_loglik_callback(self._h, self._callback, None), "pgb_set_loglik_callback")

    # -- one astep -----------------------------------------------------------------
    def step(self, tune: bool, fetch: bool = True):
        """One astep.  ``fetch=True`` (what ``PGBART.astep`` does): ``sum_trees`` comes back as a host
        array through ``pgb_step_host`` -- one device->host transaction that also carries the step's
        trees, so the following ``export_trees(0)`` does not touch the device.  ``fetch=False``:
        ``pgb_step`` leaves ``sum_trees`` in the device buffer :meth:`sum_trees_device` returns."""
        lib, mem = self.backend.lib, self.backend.mem
        K, n = self.settings.n_outputs, self.settings.n
        self._steps += 1
        if fetch:
            st = mem.host_result(K * n)
            rc = lib.lib.pgb_step_host(self._h, int(bool(tune)), st.ctypes.data, self._vi.ctypes.data,
                                       C.byref(self.counters))
            self._check(rc, "pgb_step_host")
            self._check_saturation()
            return (st.reshape(K, n) if K > 1 else st), self._vi.copy()
        rc = lib.lib.pgb_step(self._h, int(bool(tune)), mem.ptr(self._out), self._vi.ctypes.data,
                              C.byref(self.counters))
        self._check(rc, "pgb_step")
        self._check_saturation()
        return None, self._vi.copy()

    def step_async(self, tune: bool, n_steps: int) -> None:
        """Start ``n_steps`` asteps and return while they run; :meth:`sync` waits for them."""
        lib = self.backend.lib
        self._steps += int(n_steps)
        lib.check(lib.lib.pgb_step_async(self._h, int(bool(tune)), int(n_steps)), "pgb_step_async")

    def sync(self) -> dict:
        lib = self.backend.lib
        self._check(lib.lib.pgb_sync(self._h, C.byref(self.counters)), "pgb_sync")
        self._check_saturation()
        return self.counters.as_dict()

    def _check(self, rc: int, what: str) -> None:
        """``lib.check`` that re-raises an exception a Python log-likelihood callback raised."""
        err = getattr(self, "_callback_error", None)
        if rc != _abi.PGB_OK and err is not None:
            self._callback_error = None
            raise _abi.PGBError(f"{what}: the log-likelihood callback raised {type(err).__name__}: {err}") from err
        self.backend.lib.check(rc, what)

    def _check_saturation(self) -> None:
        """Fixed-point sums are exact only inside the declared range (PyBartSettings.range_exp);
        a saturated term makes the draws invalid, so it is an error, not a statistic."""
        new = int(self.counters.saturations) - self._sat_seen
        if new > 0:
            self._sat_seen = int(self.counters.saturations)  # later steps are judged on their own
            raise _abi.PGBError(
                f"{new} fixed-point saturation events by astep {self._steps}: |sum_trees| or the "
                f"residuals left the range 2^{self.settings.range_exp} (an offset / observed response far "
                "outside the range the sampler was sized for?); this draw is invalid -- re-create the "
                "sampler with a larger PyBartSettings.range_exp (PGBART(range_exp=...))"
            )

    def sum_trees_device(self):
        """The device buffer the last ``step`` wrote ``sum_trees`` into."""
        return self._out

    # -- tree export -----------------------------------------------------------------
    def export_trees(self, which: int):
        """The trees of the last step (``which=0``: a per-draw batch) or all m current trees (``which=1``: a
        baseline forest) as ONE packed record (``pgb_export_trees_packed``), decoded lazily: a
        :class:`~pymc_bart_amd.trees.PackedTrees`, which behaves like :class:`TreeArrays`."""
        lib = self.backend.lib
        buf = self._pack_buf
        nb = C.c_int64()
        rc = lib.lib.pgb_export_trees_packed(self._h, which, buf.ctypes.data, buf.size, C.byref(nb))
        if rc == _abi.PGB_E_NOMEM:  # first baseline forest / an unusually bushy batch: grow once, retry
            self._pack_buf = buf = np.empty(int(nb.value) * 2, np.uint8)
            rc = lib.lib.pgb_export_trees_packed(self._h, which, buf.ctypes.data, buf.size, C.byref(nb))
        lib.check(rc, "pgb_export_trees_packed")
        return PackedTrees(buf[: int(nb.value)].tobytes())

    def state(self) -> dict:
        lib = self.backend.lib
        sd = np.zeros(self.settings.n_outputs, np.float64)
        it = C.c_int64()
        lo = C.c_int32()
        lib.check(lib.lib.pgb_get_state(self._h, sd.ctypes.data, C.byref(it), C.byref(lo)),
                  "pgb_get_state")
        return {"leaf_sd": sd, "iter": int(it.value), "lower": int(lo.value)}

    def split_weights(self) -> np.ndarray:
        lib = self.backend.lib
        a = np.zeros(self.settings.p, np.float64)
        lib.check(lib.lib.pgb_get_split_weights(self._h, a.ctypes.data), "pgb_get_split_weights")
        return a

    # -- checkpoint / resume ---------------------------------------------------------
    def checkpoint(self) -> bytes:
        """Opaque image of the chain at this idle point (see ``pgb_checkpoint_save``)."""
        lib = self.backend.lib
        nb = C.c_int64()
        lib.check(lib.lib.pgb_checkpoint_size(self._h, C.byref(nb)), "pgb_checkpoint_size")
        buf = np.empty(int(nb.value), np.uint8)
        lib.check(lib.lib.pgb_checkpoint_save(self._h, buf.ctypes.data, buf.size), "pgb_checkpoint_save")
        return buf.tobytes()

    def restore(self, blob: bytes) -> None:
        """Continue a chain from :meth:`checkpoint`.  This sampler must have been built with the
        same settings and data; the continuation is bit-identical to the uninterrupted chain."""
        lib = self.backend.lib
        buf = np.frombuffer(blob, np.uint8)
        lib.check(lib.lib.pgb_checkpoint_load(self._h, buf.ctypes.data, buf.size), "pgb_checkpoint_load")
        self._lik_key = None  # the image carries its own likelihood parameters

    def profile(self, enable: bool) -> tuple[float, int]:
        lib = self.backend.lib
        ms = C.c_double()
        nl = C.c_int64()
        lib.check(lib.lib.pgb_profile(self._h, int(enable), C.byref(ms), C.byref(nl)), "pgb_profile")
        return float(ms.value), int(nl.value)

    def profile_kernels(self) -> dict:
        """Per-kernel event time of the last profiled region (after ``profile(False)``):
        ``{name: {"ms", "launches", "workgroups"}}`` for the kernels that ran."""
        lib = self.backend.lib
        out = {}
        for which, name in enumerate(("k_ctrl", "k_rows", "k_loglik")):
            ms, nl, wg = C.c_double(), C.c_int64(), C.c_int32()
            lib.check(lib.lib.pgb_profile_kernel(self._h, which, C.byref(ms), C.byref(nl), C.byref(wg)),
                      "pgb_profile_kernel")
            if nl.value:
                out[name] = {"ms": float(ms.value), "launches": int(nl.value), "workgroups": int(wg.value)}
        return out

    def profile_clock(self) -> tuple[float, int]:
        """Device-clock duration of the row-pass launches of the last profiled region (after
        ``profile(False)``): (total ms, launches)."""
        lib = self.backend.lib
        ms = C.c_double()
        nl = C.c_int64()
        lib.check(lib.lib.pgb_profile_clock(self._h, C.byref(ms), C.byref(nl)), "pgb_profile_clock")
        return float(ms.value), int(nl.value)
