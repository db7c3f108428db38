"""MI355X-native particle-Gibbs BART sampler behind PyMC-BART's PGBART step API."""

from . import _abi
from .sampler import PyBartSettings, PySampler
from .trees import PosteriorSampler, TreeArrays

__version__ = "0.1.0"
__all__ = ["PyBartSettings", "PySampler", "TreeArrays", "PosteriorSampler", "_abi"]
