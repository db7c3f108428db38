"""Reader of the chain image ``pgb_checkpoint_save`` writes (``include/pgbart_image.h``).

The image is the state of one chain between two asteps in a layout that belongs to no backend; this module gives
its sections as NumPy views -- to inspect a checkpoint, and for the tests that compare the image two backends
write at the same point of a chain field by field.  (The reference pickles its step method instead and has no such
file; its tree history ``(baseline_forest, batches)``, ``utils.py:124-127``, is :mod:`pymc_bart_amd.trees`.)
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _abi

IMAGE_VERSION = 1


class ImageHeader(C.Structure):
    """``pgb_image_header``."""

    _fields_ = [
        ("magic", C.c_char * 8),
        ("version", C.c_int32),
        ("header_bytes", C.c_int32),
        ("total_bytes", C.c_int64),
        ("s", _abi.Settings),
        ("iter", C.c_int64),
        ("rs_count", C.c_int64),
        ("lower", C.c_int32),
        ("last_lower", C.c_int32),
        ("last_n", C.c_int32),
        ("total_nodes", C.c_int32),
        ("leaf_sd", C.c_double * _abi.MAX_OUTPUTS),
        ("lik_param", C.c_double * 2),
        ("ctr", _abi.Counters),
        ("writer", C.c_char * 16),
    ]


def _pad8(b: int) -> int:
    return (b + 7) & ~7


@dataclass
class ChainImage:
    """The sections of one image.  Arrays are read-only views into the blob."""

    header: ImageHeader
    sum_trees: np.ndarray   # (K, n)
    rs_mean: np.ndarray     # (K, n)
    rs_m2: np.ndarray       # (K, n)
    alpha: np.ndarray       # (p,) int64
    cdf: np.ndarray         # (p,) int64
    node_off: np.ndarray    # (m + 1,)
    var: np.ndarray
    left: np.ndarray
    right: np.ndarray
    depth: np.ndarray
    label: np.ndarray
    svar: np.ndarray
    count: np.ndarray
    split: np.ndarray
    xbar: np.ndarray
    value: np.ndarray       # (N, K)
    slope: np.ndarray       # (N, K)
    lid: np.ndarray         # (m, n) uint8

    @classmethod
    def parse(cls, blob: bytes) -> "ChainImage":
        if len(blob) < C.sizeof(ImageHeader):
            raise ValueError("chain image truncated")
        hd = ImageHeader.from_buffer_copy(blob[: C.sizeof(ImageHeader)])
        if hd.magic != b"PGBIMAGE" or hd.version != IMAGE_VERSION or hd.header_bytes != C.sizeof(ImageHeader):
            raise ValueError("not a chain image of this release (include/pgbart_image.h)")
        if hd.total_bytes > len(blob):
            raise ValueError("chain image truncated")
        n, p, m, K, N = int(hd.s.n), int(hd.s.p), int(hd.s.m), int(hd.s.n_outputs), int(hd.total_nodes)
        buf = np.frombuffer(blob, np.uint8)
        off = _pad8(C.sizeof(ImageHeader))

        def take(dtype, count, pad=False):
            nonlocal off
            nb = np.dtype(dtype).itemsize * count
            a = buf[off: off + nb].view(dtype)
            off += _pad8(nb) if pad else nb
            return a

        st = take(np.float64, K * n).reshape(K, n)
        mean = take(np.float64, K * n).reshape(K, n)
        m2 = take(np.float64, K * n).reshape(K, n)
        alpha = take(np.int64, p)
        cdf = take(np.int64, p)
        node_off = take(np.int32, m + 1, pad=True)
        ints = take(np.int32, 6 * N, pad=True).reshape(6, N)
        count = take(np.int64, N)
        split = take(np.float64, N)
        xbar = take(np.float64, N)
        value = take(np.float64, N * K).reshape(N, K)
        slope = take(np.float64, N * K).reshape(N, K)
        lid = take(np.uint8, m * n, pad=True).reshape(m, n)
        if off != hd.total_bytes:
            raise ValueError("chain image is inconsistent (section sizes)")
        return cls(hd, st, mean, m2, alpha, cdf, node_off, ints[0], ints[1], ints[2], ints[3], ints[4], ints[5],
                   count, split, xbar, value, slope, lid)

    @property
    def writer(self) -> str:
        return self.header.writer.decode()

    @property
    def leaf_sd(self) -> np.ndarray:
        return np.array(self.header.leaf_sd[: int(self.header.s.n_outputs)])

    def forest(self, rules=None):
        """The m accepted trees of the image as a :class:`~pymc_bart_amd.trees.TreeArrays` -- what
        ``pgb_export_trees(h, 1, ...)`` returns for the chain the image was taken from, so a checkpoint can be
        predicted from (``PosteriorSampler(forest, [[0 .. m-1]], m, K)``) without a sampler.  ``rules``: the
        ``PGB_RULE_*`` of the p columns (the image holds the trees, not the model's split rules); ``None``: every
        split is continuous."""
        from .trees import TreeArrays

        m = int(self.header.s.m)
        var = np.array(self.var, np.int32)
        if rules is None:
            rule = np.zeros(var.shape[0], np.int32)
        else:
            rule = np.where(var >= 0, np.asarray(rules, np.int32)[np.maximum(var, 0)], 0).astype(np.int32)
        return TreeArrays(n_outputs=int(self.header.s.n_outputs), tree_id=np.arange(m, dtype=np.int32),
                          node_off=np.array(self.node_off, np.int32), var=var, split=np.array(self.split),
                          left=np.array(self.left, np.int32), right=np.array(self.right, np.int32),
                          count=np.array(self.count, np.int64), value=np.array(self.value),
                          slope=np.array(self.slope), xbar=np.array(self.xbar), svar=np.array(self.svar, np.int32),
                          rule=rule)

    def chain_fields(self) -> dict:
        """Everything two backends must agree on at the same point of a chain: all of the image except who wrote it
        and the backend-specific ``slots`` counter."""
        hd = self.header
        ctr = hd.ctr.as_dict()
        ctr.pop("slots")
        out = {f: getattr(self, f) for f in ("sum_trees", "rs_mean", "rs_m2", "alpha", "cdf", "node_off", "var", "left",
                                             "right", "depth", "label", "svar", "count", "split", "xbar", "value",
                                             "slope", "lid")}
        out.update(iter=int(hd.iter), rs_count=int(hd.rs_count), lower=int(hd.lower), last_lower=int(hd.last_lower),
                   last_n=int(hd.last_n), leaf_sd=self.leaf_sd, lik_param=np.array(hd.lik_param[:]), counters=ctr,
                   settings=bytes(hd.s))
        return out


def differing_fields(a: "ChainImage", b: "ChainImage") -> list:
    """Names of the chain fields in which two images differ (empty: the same chain state)."""
    fa, fb = a.chain_fields(), b.chain_fields()
    bad = []
    for k, va in fa.items():
        vb = fb[k]
        if isinstance(va, np.ndarray):
            same = va.shape == vb.shape and va.dtype == vb.dtype and va.tobytes() == vb.tobytes()  # (bits: NaN-safe, -0.0-strict)
        else:
            same = va == vb
        if not same:
            bad.append(k)
    return bad
