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
    """Per-draw split-variable counts -> the string stored in ``sample_stats["variable_inclusion"]``.

    Wire format (reference ``utils.py:1387-1398``): every count as an unsigned LEB128 varint --
    7 value bits per byte, least significant group first, bit 7 set on all but the last byte of a
    count -- and the byte string base64-encoded.  Golden vectors: ``tests/golden/vi_codec.json``."""
    counts = np.asarray(vec, dtype=np.int64).ravel()
    if counts.size == 0:
        return ""
    lo, hi = int(counts.min()), int(counts.max())
    if lo < 0:
        raise ValueError("variable inclusion counts must be non-negative")
    if hi < 0x80:  # what a step produces: every count is its own single byte
        return base64.b64encode(counts.astype(np.uint8).tobytes()).decode("ascii")
    # general case, vectorised: septet k of every count, kept while the count still has bits at or above it
    n_groups = max(1, -(-hi.bit_length() // 7))
    shifts = 7 * np.arange(n_groups, dtype=np.int64)
    groups = (counts[:, None] >> shifts) & 0x7F
    used = (counts[:, None] >> shifts) > 0
    used[:, 0] = True
    more = np.zeros_like(used)
    more[:, :-1] = used[:, 1:]                      # a continuation bit on all but a count's last septet
    payload = (groups | (more.astype(np.int64) << 7))[used].astype(np.uint8)  # row-major: counts in order
    return base64.b64encode(payload.tobytes()).decode("ascii")


def _decode_vi(s: str, length: int) -> list[int]:
    """Inverse of :func:`_encode_vi` (reference ``utils.py:1368-1384``): at most ``length`` counts;
    a truncated trailing varint yields the bits that are there, as upstream does."""
    raw = np.frombuffer(base64.b64decode(s), dtype=np.uint8)
    last_bytes = np.flatnonzero(raw < 0x80)  # bytes that close a varint
    bounds = [0] + (last_bytes + 1).tolist()
    if bounds[-1] < raw.size:
        bounds.append(int(raw.size))  # unterminated tail
    out: list[int] = []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        if len(out) == length:
            break
        septets = (raw[lo:hi] & 0x7F).tolist()
        out.append(sum(v << (7 * k) for k, v in enumerate(septets)))
    return out


def _sample_posterior(sampler, X, rng: np.random.Generator, size=None, excluded=None) -> np.ndarray:
    """Posterior predictions at the rows of ``X`` for randomly chosen stored draws.

    Contract of reference ``utils.py:26-71``: the draw indices come from ONE
    ``rng.integers(0, n_draws, size=prod(size))`` call (so equal generators give equal draws for
    any ``X``), a list of samplers contributes its outputs side by side, and the result has shape
    ``(*size, n_rows, n_outputs)``."""
    if size is None:
        lead: tuple[int, ...] = ()
    elif np.isscalar(size):
        lead = (int(size),)
    else:
        lead = tuple(int(v) for v in size)
    n_pred = int(np.prod(lead, dtype=np.int64)) if lead else 1
    group = sampler if isinstance(sampler, list) else [sampler]
    rows = X if not isinstance(X, (np.ndarray, list, tuple)) and hasattr(X, "data_ptr") \
        else np.ascontiguousarray(X, dtype=np.float64)  # (a handle from _resident_rows passes through)
    picks = rng.integers(0, group[0].n_draws, size=n_pred).tolist()
    drop = None if excluded is None else [int(v) for v in excluded]
    blocks = [np.asarray(g.sample_posterior(rows, picks, drop)) for g in group]  # (n_pred, K_g, n_rows)
    stacked = blocks[0] if len(blocks) == 1 else np.concatenate(blocks, axis=1)
    return np.moveaxis(stacked, 1, 2).reshape(lead + (stacked.shape[2], stacked.shape[1]))


def _resident_rows(sampler, X):
    """``X`` uploaded once for a sweep of predictions on the same rows (see
    ``PosteriorSampler.resident_rows``); a list of samplers shares the handle of its first chain."""
    first = sampler[0] if isinstance(sampler, list) else sampler
    upload = getattr(first._chain_samplers[0], "resident_rows", None)
    return upload(X) if upload is not None else X


class _MultiChainSampler:
    """All chains of one BART variable behind one draw index (reference ``utils.py:74-107``):
    draw ``d`` belongs to the chain whose range of the concatenated history contains it."""

    def __init__(self, chain_samplers: list):
        self._parts = list(chain_samplers)
        if not self._parts:
            raise ValueError("No posterior draws available yet: run the sampler first.")
        self._starts = np.concatenate([[0], np.cumsum([part.n_draws for part in self._parts])]).astype(np.int64)

    @property
    def _chain_samplers(self):  # name used by callers that reach into the chains
        return self._parts

    @property
    def n_draws(self) -> int:
        return int(self._starts[-1])

    @property
    def n_outputs(self) -> int:
        return self._parts[0].n_outputs

    def sample_posterior(self, X, draw_indices, excluded):
        want = np.asarray(draw_indices, dtype=np.int64).ravel()
        owner = np.digitize(want, self._starts[1:])  # chain of every requested draw
        result = None
        for c in np.unique(owner).tolist():
            where = np.flatnonzero(owner == c)
            part = self._parts[c].sample_posterior(X, (want[where] - self._starts[c]).tolist(), excluded)
            if result is None:
                result = np.empty((want.size,) + part.shape[1:], dtype=part.dtype)
            result[where] = part
        return result


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
    chain_samplers = [  # the reference's call (utils.py:124-127); the trees carry their own split rules
        PosteriorSampler.from_history(batches, baseline, op.m, op.n_outputs, backend=backend)
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
