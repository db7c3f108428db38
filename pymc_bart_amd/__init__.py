"""MI355X-native particle-Gibbs BART sampler behind PyMC-BART's PGBART step API.

Public names mirror what the reference imports from the external ``bartrs`` wheel
(``pymc_bart/pymc_bart.py:2``, ``tests/test_bart.py:4``).  All compute runs in
``csrc/libpgbart_hip.so`` (gfx950); there is no CPU fallback.
"""

from . import _abi
from .pgbart import (PGBART, AsymmetricLaplaceLikelihood, BARTOp, BernoulliLikelihood, CallbackLikelihood, GammaLikelihood, CategoricalLikelihood, NegativeBinomialLikelihood,
                     NormalLikelihood, NormalMeanScaleLikelihood, PoissonLikelihood, StudentTLikelihood)
from .sampler import PyBartSettings, PySampler
from .trees import PosteriorSampler, TreeArrays
from .importance import compute_variable_importance, get_variable_inclusion, vi_to_kulprit
from .partial import individual_conditional_expectation, partial_dependence


def _register_step_method():
    """Import side effect the reference relies on (``pymc_bart/__init__.py:15-18``: ``import bartrs``
    makes PGBART known to ``pm.sample``'s step assignment): BART variables get PGBART through
    ``competence`` without an explicit ``step=`` (reference ``tests/test_bart.py:167-208``)."""
    try:
        import pymc as pm
    except Exception:  # noqa: BLE001 - PyMC is optional
        return False
    methods = list(getattr(pm, "STEP_METHODS", ()))
    if PGBART not in methods:
        pm.STEP_METHODS = methods + [PGBART]
    return True


_register_step_method()

__version__ = "0.1.0"
__all__ = [
    "PGBART", "BARTOp", "CallbackLikelihood", "NormalLikelihood", "BernoulliLikelihood", "CategoricalLikelihood", "NormalMeanScaleLikelihood", "PoissonLikelihood", "NegativeBinomialLikelihood", "AsymmetricLaplaceLikelihood", "StudentTLikelihood", "GammaLikelihood",
    "PyBartSettings", "PySampler", "TreeArrays", "PosteriorSampler", "compute_variable_importance", "get_variable_inclusion", "vi_to_kulprit",
    "partial_dependence", "individual_conditional_expectation", "_abi",
]
