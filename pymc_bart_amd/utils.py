"""Host-side glue either side of the sampler, mirroring reference ``pymc_bart/utils.py``.

Only the pieces on or next to the hot path are here:
the variable-inclusion wire format the step method emits and the posterior-sampling
dispatch.  Variable importance lives in ``importance.py``; plotting / PDP are out of scope.
"""

from __future__ import annotations

import base64

import numpy as np

from .trees import PosteriorSampler


def _encode_vi(vec) -> str:
    """LEB128 varints of the per-draw split-variable counts, base64 encoded.

    Wire format of ``sample_stats["variable_inclusion"]`` (reference
    ``utils.py:1387-1398``); golden vectors in ``tests/golden/vi_codec.json``.
    """
    out = bytearray()
    for num in vec:
        n = int(num)
        if n < 0:
            raise ValueError("variable inclusion counts must be non-negative")
        while n > 127:
            out.append((n & 0x7F) | 0x80)
            n >>= 7
        out.append(n & 0x7F)
    return base64.b64encode(bytes(out)).decode("ascii")


def _decode_vi(s: str, length: int) -> list[int]:
    """Inverse of :func:`_encode_vi` (reference ``utils.py:1368-1384``)."""
    data = base64.b64decode(s)
    result: list[int] = []
    i = 0
    while len(result) < length and i < len(data):
        num = 0
        shift = 0
        while i < len(data):
            byte = data[i]
            i += 1
            num |= (byte & 0x7F) << shift
            if not byte & 0x80:
                break
            shift += 7
        result.append(num)
    return result


def _sample_posterior(sampler, X, rng: np.random.Generator, size=None, excluded=None) -> np.ndarray:
    """Draw posterior predictions; same contract as reference ``utils.py:26-71``:
    result shape ``(*size, n_rows, n_outputs)``; draw indices depend only on ``rng``."""
    if size is None:
        size_iter: tuple[int, ...] | list[int] = ()
    elif isinstance(size, int):
        size_iter = [size]
    else:
        size_iter = size
    flatten_size = 1
    for s in size_iter:
        flatten_size *= s

    X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
    excl = list(excluded) if excluded is not None else None
    first = sampler[0] if isinstance(sampler, list) else sampler
    draw_indices = rng.integers(0, first.n_draws, size=flatten_size).tolist()
    if isinstance(sampler, list):
        pred = np.concatenate([s.sample_posterior(X, draw_indices, excl) for s in sampler], axis=1)
    else:
        pred = sampler.sample_posterior(X, draw_indices, excl)
    return pred.transpose((0, 2, 1)).reshape((*size_iter, -1, pred.shape[1]))


class _MultiChainSampler:
    """Dispatch each draw to the chain it came from (reference ``utils.py:74-107``)."""

    def __init__(self, chain_samplers: list):
        if not chain_samplers:
            raise ValueError("No posterior draws available yet: run the sampler first.")
        self._chain_samplers = chain_samplers
        self._offsets = np.cumsum([0] + [s.n_draws for s in chain_samplers])

    @property
    def n_draws(self) -> int:
        return int(self._offsets[-1])

    @property
    def n_outputs(self) -> int:
        return self._chain_samplers[0].n_outputs

    def sample_posterior(self, X, draw_indices, excluded):
        draw_indices = np.asarray(draw_indices)
        chain_of_draw = np.searchsorted(self._offsets, draw_indices, side="right") - 1
        out = None
        for chain_idx, sampler in enumerate(self._chain_samplers):
            mask = chain_of_draw == chain_idx
            if not np.any(mask):
                continue
            local = (draw_indices[mask] - self._offsets[chain_idx]).tolist()
            preds = sampler.sample_posterior(X, local, excluded)
            if out is None:
                out = np.empty((len(draw_indices), *preds.shape[1:]), dtype=preds.dtype)
            out[mask] = preds
        return out


_posterior_sampler_cache: dict[int, tuple] = {}


def _get_posterior_sampler(op, backend=None) -> _MultiChainSampler:
    """Rebuild (and cache) the per-chain samplers from ``op.all_trees``
    (reference ``utils.py:110-130``).

    The reference keys its cache on ``id(op)`` and the number of chains alone; an ``id`` can be
    reused by a later object and a chain can grow, so the entry here also pins the op itself
    (weakly), the number of stored draws and the backend."""
    import weakref

    n_chains = len(op.all_trees)
    n_batches = sum(len(batches) for _, batches in op.all_trees)
    cached = _posterior_sampler_cache.get(id(op))
    if cached is not None:
        ref, c_chains, c_batches, c_backend, sampler = cached
        if ref() is op and c_chains == n_chains and c_batches == n_batches and c_backend is backend:
            return sampler
    rules = getattr(op, "_rule_ids", None)
    chain_samplers = [
        PosteriorSampler.from_history(batches, baseline, op.m, op.n_outputs, rules=rules,
                                      backend=backend)
        for baseline, batches in op.all_trees
    ]
    sampler = _MultiChainSampler(chain_samplers)
    try:
        ref = weakref.ref(op)
    except TypeError:  # not weak-referenceable: never trust the cache for it
        return sampler
    for key in [k for k, v in _posterior_sampler_cache.items() if v[0]() is None]:
        del _posterior_sampler_cache[key]  # entries of ops that are gone
    _posterior_sampler_cache[id(op)] = (ref, n_chains, n_batches, backend, sampler)
    return sampler
