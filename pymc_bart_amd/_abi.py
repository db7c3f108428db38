"""ctypes binding of the C ABI declared in ``include/pgbart.h``.

This is the stub a pymc-bart maintainer would add in place of the PyO3 import
``from bartrs.bartrs import PosteriorSampler, PyBartSettings, PySampler, TreeArrays``
(reference ``pymc_bart/pymc_bart.py:2``).  It binds any shared library exporting the
``pgb_*`` symbols; the product only ever loads the gfx950 HIP build
(:func:`load_hip_library`) and raises if it is missing -- there is no CPU fallback.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

PGB_OK = 0
PGB_E_NOMEM = -3
MAX_DEPTH = 64
MAX_PARTICLES = 128  # of the p128 build of the library; the default build takes 64 (PGBLibrary.max_particles)
MAX_OUTPUTS = 16  # PGB_MAX_OUTPUTS (include/pgbart_spec.h)
MAX_NODES = 255
ABI_VERSION = 6  # PGB_ABI_VERSION of include/pgbart.h this binding was written against

RULE_CONTINUOUS = 0
RULE_ONEHOT = 1
RULE_SUBSET = 2
SUBSET_BITS = 52
RESPONSES = {"constant": 0, "linear": 1, "mix": 2}
RULES = {
    "ContinuousSplit": RULE_CONTINUOUS,
    "ContinuousSplitRule": RULE_CONTINUOUS,
    "OneHotSplit": RULE_ONEHOT,
    "OneHotSplitRule": RULE_ONEHOT,
    "SubsetSplit": RULE_SUBSET,
    "SubsetSplitRule": RULE_SUBSET,
}

FAMILY_NORMAL = 0
FAMILY_BERNOULLI_PROBIT = 1
FAMILY_BERNOULLI_LOGIT = 2
FAMILY_CATEGORICAL = 3
FAMILY_NORMAL_MEANSCALE = 4
FAMILIES = {
    "normal": FAMILY_NORMAL,
    "bernoulli_probit": FAMILY_BERNOULLI_PROBIT,
    "bernoulli_logit": FAMILY_BERNOULLI_LOGIT,
    "categorical": FAMILY_CATEGORICAL,
    "normal_meanscale": FAMILY_NORMAL_MEANSCALE,
    "poisson_log": 5,
    "negbin_log": 6,
    "asymmetric_laplace": 7,
    "student_t": 8,
    "gamma_log": 9,
    "callback": 10,
}

#: every symbol ``include/pgbart.h`` declares (checked by tests/test_abi.py)
SYMBOLS = (
    "pgb_last_error",
    "pgb_backend_name",
    "pgb_max_particles",
    "pgb_create",
    "pgb_destroy",
    "pgb_set_data",
    "pgb_set_response",
    "pgb_set_offset",
    "pgb_set_likelihood",
    "pgb_set_loglik_callback",
    "pgb_set_output_stream",
    "pgb_step",
    "pgb_step_host",
    "pgb_step_async",
    "pgb_sync",
    "pgb_export_trees",
    "pgb_export_trees_packed",
    "pgb_get_state",
    "pgb_get_split_weights",
    "pgb_predict",
    "pgb_profile",
    "pgb_profile_clock",
    "pgb_profile_kernel",
    "pgb_checkpoint_size",
    "pgb_checkpoint_save",
    "pgb_checkpoint_load",
    "pgb_abi_version",
)


class Settings(C.Structure):
    """``pgb_settings`` -- counterpart of bartrs' ``PyBartSettings``."""

    _fields_ = [
        ("n", C.c_int64),
        ("p", C.c_int32),
        ("m", C.c_int32),
        ("num_particles", C.c_int32),
        ("n_outputs", C.c_int32),
        ("family", C.c_int32),
        ("batch_tune", C.c_int32),
        ("batch_draw", C.c_int32),
        ("range_exp", C.c_int32),
        ("response", C.c_int32),
        ("compat", C.c_int32),
        ("seed", C.c_uint64),
        ("init_sum", C.c_double),
        ("init_leaf", C.c_double),
        ("init_leaf_sd", C.c_double),
        ("prior_leaf", C.c_double * MAX_DEPTH),
    ]


class Counters(C.Structure):
    _fields_ = [
        ("particle_steps", C.c_int64),
        ("tree_updates", C.c_int64),
        ("rows_touched", C.c_int64),
        ("rounds", C.c_int64),
        ("saturations", C.c_int64),
        ("slots", C.c_int64),
        ("partitions", C.c_int64),
    ]

    def as_dict(self) -> dict:
        return {name: int(getattr(self, name)) for name, _ in self._fields_}


class TreeArraysC(C.Structure):
    """``pgb_tree_arrays`` -- counterpart of bartrs' ``TreeArrays``."""

    _fields_ = [
        ("n_trees", C.c_int32),
        ("n_outputs", C.c_int32),
        ("total_nodes", C.c_int32),
        ("tree_id", C.POINTER(C.c_int32)),
        ("node_off", C.POINTER(C.c_int32)),
        ("var", C.POINTER(C.c_int32)),
        ("split", C.POINTER(C.c_double)),
        ("left", C.POINTER(C.c_int32)),
        ("right", C.POINTER(C.c_int32)),
        ("count", C.POINTER(C.c_int64)),
        ("value", C.POINTER(C.c_double)),
        ("slope", C.POINTER(C.c_double)),
        ("xbar", C.POINTER(C.c_double)),
        ("svar", C.POINTER(C.c_int32)),
        ("rule", C.POINTER(C.c_int32)),
    ]


#: ``pgb_loglik_fn``: int fn(void* ctx, const int64_t* row, const double* y, const double* mu, int64_t n, double* out)
LOGLIK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                        C.c_int64, C.POINTER(C.c_double))


class PGBError(RuntimeError):
    pass


def _ptr(arr: np.ndarray, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


class PGBLibrary:
    """A loaded shared library implementing ``include/pgbart.h``."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.path = path
        self.lib = C.CDLL(path)
        lib = self.lib
        # a library built against another revision of include/pgbart.h would be called with shifted arguments
        # (pgb_predict lost one between rounds 4 and 5, pgb_tree_arrays gained `rule`): refuse it by name
        try:
            have = int(lib.pgb_abi_version())
        except AttributeError:
            have = -1
        if have != ABI_VERSION:
            raise PGBError(f"{path} implements revision {have if have >= 0 else '< 6 (no pgb_abi_version)'} of "
                           f"include/pgbart.h, this binding revision {ABI_VERSION}: rebuild it "
                           "(`python -c 'import __graft_entry__ as g; g.build()'`)")
        vp = C.c_void_p
        lib.pgb_last_error.restype = C.c_char_p
        lib.pgb_last_error.argtypes = []
        lib.pgb_backend_name.restype = C.c_char_p
        lib.pgb_backend_name.argtypes = []
        lib.pgb_create.argtypes = [C.POINTER(Settings), vp, C.POINTER(vp)]
        lib.pgb_destroy.argtypes = [vp]
        lib.pgb_set_data.argtypes = [vp, vp, C.c_int64, vp, vp]
        lib.pgb_set_response.argtypes = [vp, vp]
        lib.pgb_set_offset.argtypes = [vp, vp]
        lib.pgb_set_likelihood.argtypes = [vp, vp, C.c_int32]
        lib.pgb_set_loglik_callback.argtypes = [vp, LOGLIK_FN, vp]
        lib.pgb_set_output_stream.argtypes = [vp, vp]
        lib.pgb_step.argtypes = [vp, C.c_int32, vp, vp, C.POINTER(Counters)]
        lib.pgb_step_host.argtypes = [vp, C.c_int32, vp, vp, C.POINTER(Counters)]
        lib.pgb_step_async.argtypes = [vp, C.c_int32, C.c_int32]
        lib.pgb_sync.argtypes = [vp, C.POINTER(Counters)]
        lib.pgb_export_trees.argtypes = [vp, C.c_int32, C.POINTER(TreeArraysC)]
        lib.pgb_export_trees_packed.argtypes = [vp, C.c_int32, vp, C.c_int64, C.POINTER(C.c_int64)]
        lib.pgb_get_state.argtypes = [vp, vp, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        lib.pgb_get_split_weights.argtypes = [vp, vp]
        lib.pgb_predict.argtypes = [
            C.POINTER(TreeArraysC), vp, C.c_int32, C.c_int32, vp, C.c_int64, C.c_int32,
            C.c_int64, vp, C.c_int32, vp, vp,
        ]
        lib.pgb_profile.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        lib.pgb_profile_clock.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        lib.pgb_profile_kernel.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                           C.POINTER(C.c_int32)]
        lib.pgb_checkpoint_size.argtypes = [vp, C.POINTER(C.c_int64)]
        lib.pgb_checkpoint_save.argtypes = [vp, vp, C.c_int64]
        lib.pgb_checkpoint_load.argtypes = [vp, vp, C.c_int64]
        for name in SYMBOLS:
            fn = getattr(lib, name)
            if name not in ("pgb_last_error", "pgb_backend_name"):
                fn.restype = C.c_int
        lib.pgb_max_particles.argtypes = []
        lib.pgb_abi_version.argtypes = []

    @property
    def backend_name(self) -> str:
        return self.lib.pgb_backend_name().decode()

    @property
    def max_particles(self) -> int:
        return int(self.lib.pgb_max_particles())

    def check(self, rc: int, what: str) -> None:
        if rc != PGB_OK:
            msg = self.lib.pgb_last_error()
            raise PGBError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


_HIP_LIB: dict = {}


def hip_library_path(max_particles: int = 64) -> str:
    """The in-tree gfx950 build -- ``libpgbart_hip.so`` (up to 64 particles, one per lane) or, for more,
    ``libpgbart_hip_p128.so`` (the same source with ``-DPGB_MAX_PARTICLES=128``).  ``PGBART_HIP_LIB`` names another
    build of the 64-particle library (profiling / experiment builds -- it must still be the HIP backend, see
    :func:`load_hip_library`)."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    if max_particles > 64:
        return os.path.join(here, "libpgbart_hip_p128.so")
    override = os.environ.get("PGBART_HIP_LIB")
    if override:
        return override
    return os.path.join(here, "libpgbart_hip.so")


def load_hip_library(max_particles: int = 64) -> PGBLibrary:
    """Load the gfx950 HIP build that takes ``max_particles`` particles.  Fails loudly when it has not been built."""
    key = 128 if max_particles > 64 else 64
    if key not in _HIP_LIB:
        path = hip_library_path(key)
        if not os.path.exists(path):
            raise PGBError(
                f"{path} is missing: the HIP extension has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` at the repository root. "
                "pymc_bart_amd has no CPU fallback."
            )
        # The library must share ONE HIP runtime with the torch tensors that hold its device
        # buffers: import torch first so that its libamdhip64.so.7 is the one already loaded.
        import torch  # noqa: F401

        lib = PGBLibrary(path)
        if lib.backend_name != "hip-gfx950":
            raise PGBError(f"{path} is not the HIP backend (it reports {lib.backend_name!r})")
        if lib.max_particles < key:
            raise PGBError(f"{path} takes {lib.max_particles} particles, {key} wanted")
        _HIP_LIB[key] = lib
    return _HIP_LIB[key]
