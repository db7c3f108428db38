"""Tree storage, tree history and prediction from stored trees.

Counterparts of the native classes the reference imports from ``bartrs``
(``pymc_bart/pymc_bart.py:2``): :class:`TreeArrays` and :class:`PosteriorSampler`.
The per-chain history layout ``(baseline_forest, batches)`` is the one consumed at
reference ``utils.py:124-127``; the prediction contract is ``utils.py:60-71,93-107``:
``sample_posterior(X, draw_indices, excluded) -> (len(draw_indices), n_outputs, n_rows)``.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _abi


@dataclass
class TreeArrays:
    """SoA storage of a list of trees (``pgb_tree_arrays``).  Plain numpy => picklable,
    so it can travel through the ``multiprocessing.Manager().list()`` mailbox the
    reference attaches to the op (``bart.py:133-135``)."""

    n_outputs: int
    tree_id: np.ndarray   # int32 [n_trees]
    node_off: np.ndarray  # int32 [n_trees + 1]
    var: np.ndarray       # int32 [nodes]
    split: np.ndarray     # float64 [nodes]
    left: np.ndarray      # int32 [nodes]
    right: np.ndarray     # int32 [nodes]
    count: np.ndarray     # int64 [nodes]
    value: np.ndarray     # float64 [nodes, n_outputs]
    # linear response: a leaf predicts value + slope * (x[svar] - xbar); svar = -1: constant leaf
    slope: np.ndarray = None  # float64 [nodes, n_outputs]
    xbar: np.ndarray = None   # float64 [nodes]
    svar: np.ndarray = None   # int32 [nodes]
    # The trees describe themselves: the rule (``_abi.RULE_*``) every split node was grown under, 0 for a leaf.
    # The reference rebuilds its predictors with ``from_history(batches, baseline_forest, op.m, op.n_outputs)``
    # (``utils.py:124-127``) -- no split rules in sight -- so a one-hot / subset split has to say so itself.
    rule: np.ndarray = None   # int32 [nodes]

    def __post_init__(self):
        nn = int(np.asarray(self.var).shape[0])
        if self.rule is None:  # a record written before the trees carried their rules: every split is `x <= v`
            self.rule = np.zeros(nn, np.int32)
        if self.slope is None:
            self.slope = np.zeros((nn, self.n_outputs), np.float64)
        self.slope = np.ascontiguousarray(self.slope, np.float64).reshape(nn, self.n_outputs)
        if self.xbar is None:
            self.xbar = np.zeros(nn, np.float64)
        if self.svar is None:
            self.svar = np.full(nn, -1, np.int32)

    def __setstate__(self, state):
        """Objects pickled before a field existed (``rule`` came in round 5; ``slope`` / ``xbar`` / ``svar`` before
        it) are restored without ``__init__``: fill what is missing the way ``__post_init__`` does."""
        self.__dict__.update(state)
        for f in ("slope", "xbar", "svar", "rule"):
            self.__dict__.setdefault(f, None)
        self.__post_init__()

    @property
    def n_trees(self) -> int:
        return int(self.tree_id.shape[0])

    @property
    def total_nodes(self) -> int:
        return int(self.var.shape[0])

    @classmethod
    def empty(cls, n_trees: int, total_nodes: int, n_outputs: int) -> "TreeArrays":
        return cls(
            n_outputs=n_outputs,
            tree_id=np.zeros(n_trees, np.int32),
            node_off=np.zeros(n_trees + 1, np.int32),
            var=np.zeros(total_nodes, np.int32),
            split=np.zeros(total_nodes, np.float64),
            left=np.zeros(total_nodes, np.int32),
            right=np.zeros(total_nodes, np.int32),
            count=np.zeros(total_nodes, np.int64),
            value=np.zeros((total_nodes, n_outputs), np.float64),
            slope=np.zeros((total_nodes, n_outputs), np.float64),
            xbar=np.zeros(total_nodes, np.float64),
            svar=np.full(total_nodes, -1, np.int32),
            rule=np.zeros(total_nodes, np.int32),
        )

    def as_c(self) -> _abi.TreeArraysC:
        c = _abi.TreeArraysC()
        c.n_trees = self.n_trees
        c.n_outputs = self.n_outputs
        c.total_nodes = self.total_nodes
        c.tree_id = _abi._ptr(self.tree_id, C.c_int32)
        c.node_off = _abi._ptr(self.node_off, C.c_int32)
        c.var = _abi._ptr(self.var, C.c_int32)
        c.split = _abi._ptr(self.split, C.c_double)
        c.left = _abi._ptr(self.left, C.c_int32)
        c.right = _abi._ptr(self.right, C.c_int32)
        c.count = _abi._ptr(self.count, C.c_int64)
        c.value = _abi._ptr(self.value, C.c_double)
        c.slope = _abi._ptr(self.slope, C.c_double)
        c.xbar = _abi._ptr(self.xbar, C.c_double)
        c.svar = _abi._ptr(self.svar, C.c_int32)
        c.rule = _abi._ptr(self.rule, C.c_int32)
        return c

    def split_variables(self, t: int) -> np.ndarray:
        a, b = self.node_off[t], self.node_off[t + 1]
        v = self.var[a:b]
        return v[v >= 0]

    @classmethod
    def from_packed(cls, raw) -> "TreeArrays":
        """Decode the record of ``pgb_export_trees_packed`` (``include/pgbart_pack.h``): zero-copy views of
        ``raw`` (read-only when ``raw`` is ``bytes``)."""
        hdr = np.frombuffer(raw, np.int32, 4)
        nt, K, N, flags = (int(v) for v in hdr)
        lin, has_rules = bool(flags & 1), bool(flags & 2)
        o = 16

        def take(dtype, count):
            nonlocal o
            a = np.frombuffer(raw, dtype, count, o)
            o += a.nbytes
            return a

        tree_id, node_off = take(np.int32, nt), take(np.int32, nt + 1)
        var, left, right = take(np.int32, N), take(np.int32, N), take(np.int32, N)
        rule = take(np.int32, N) if has_rules else None
        svar = take(np.int32, N) if lin else None
        o = (o + 7) & ~7
        split, count = take(np.float64, N), take(np.int64, N)
        value = take(np.float64, N * K).reshape(N, K)
        slope = take(np.float64, N * K).reshape(N, K) if lin else None
        xbar = take(np.float64, N) if lin else None
        return cls(n_outputs=K, tree_id=tree_id, node_off=node_off, var=var, split=split, left=left, right=right,
                   count=count, value=value, slope=slope, xbar=xbar, svar=svar, rule=rule)

    @staticmethod
    def concat(parts: list["TreeArrays"]) -> "TreeArrays":
        K = parts[0].n_outputs
        offs = [0]
        for p in parts:
            offs.append(offs[-1] + p.total_nodes)
        node_off = np.concatenate(
            [p.node_off[:-1] + o for p, o in zip(parts, offs[:-1])] + [np.array([offs[-1]], np.int32)]
        ).astype(np.int32)
        return TreeArrays(
            n_outputs=K,
            tree_id=np.concatenate([p.tree_id for p in parts]).astype(np.int32),
            node_off=node_off,
            var=np.concatenate([p.var for p in parts]).astype(np.int32),
            split=np.concatenate([p.split for p in parts]),
            left=np.concatenate([p.left for p in parts]).astype(np.int32),
            right=np.concatenate([p.right for p in parts]).astype(np.int32),
            count=np.concatenate([p.count for p in parts]).astype(np.int64),
            value=np.concatenate([p.value for p in parts], axis=0),
            slope=np.concatenate([p.slope for p in parts], axis=0).astype(np.float64),
            xbar=np.concatenate([p.xbar for p in parts]).astype(np.float64),
            svar=np.concatenate([p.svar for p in parts]).astype(np.int32),
            rule=np.concatenate([p.rule for p in parts]).astype(np.int32),
        )


class PackedTrees:
    """One tree export kept as the packed record the native library wrote (``pgb_export_trees_packed``) and
    decoded into a :class:`TreeArrays` on first use.  ``PGBART.astep`` stores one per draw: the step pays for
    a single small copy, the decoding happens when (and if) predictions are made, and the history travels
    through the reference's ``Manager().list()`` mailbox (``bart.py:134-135``) as a few KB of bytes."""

    __slots__ = ("raw", "_ta", "owner")

    def __init__(self, raw: bytes, owner: str | None = None):
        self.raw = raw
        self._ta = None
        self.owner = owner  # which chain published this forest as its baseline (PGBART._publish); survives pickling

    def decoded(self) -> TreeArrays:
        if self._ta is None:
            self._ta = TreeArrays.from_packed(self.raw)
        return self._ta

    def __getattr__(self, name):  # every TreeArrays attribute / method, through the decoded arrays
        if name.startswith("__"):
            raise AttributeError(name)
        return getattr(self.decoded(), name)

    def __reduce__(self):
        return (PackedTrees, (self.raw, self.owner))


def _as_list(batches) -> list:
    """The batches of one chain as a list.  Under PyMC they arrive as a ``multiprocessing`` list proxy
    (``PGBART._publish``): one slice request fetches them all instead of one round trip per draw."""
    return batches if isinstance(batches, list) else list(batches[:])


_HISTORY_FIELDS = ("tree_id", "node_off", "var", "split", "left", "right", "count", "value", "slope", "xbar", "svar",
                   "rule")
_HISTORY_FORMATS = ("pgbart-history-1", "pgbart-history-2")  # -1: before the nodes carried their split rules


def save_history(path, all_trees, m: int) -> None:
    """Write the per-chain tree histories ``[(baseline_forest, batches), ...]`` -- the list the
    step method keeps on ``op.all_trees`` (reference ``bart.py:134-135``, ``utils.py:124-127``) --
    to ONE ``.npz`` file (the reference has no on-disk format).

    Layout: every chain's forests are concatenated (baseline first, then the per-draw batches);
    ``c<i>_<field>`` holds the SoA arrays of chain ``i`` and ``c<i>_sizes`` the number of trees of
    each forest, which is all that is needed to cut the concatenation apart again."""
    out = {"format": np.array(_HISTORY_FORMATS[-1]), "n_chains": np.array(len(all_trees)), "m": np.array(int(m))}
    for i, (baseline, batches) in enumerate(all_trees):
        parts = [baseline] + _as_list(batches)
        cat = TreeArrays.concat(parts)
        out[f"c{i}_n_outputs"] = np.array(cat.n_outputs)
        out[f"c{i}_sizes"] = np.array([p.n_trees for p in parts], np.int64)
        for f in _HISTORY_FIELDS:
            out[f"c{i}_{f}"] = getattr(cat, f)
    np.savez_compressed(path, **out)


def load_history(path):
    """Inverse of :func:`save_history`: returns ``(all_trees, m)`` with ``all_trees`` in the layout
    ``PosteriorSampler.from_history`` / ``_get_posterior_sampler`` consume.  The split rules are part of the
    trees; a file of format 1 (per-column rules beside the trees) is read into the same form."""
    z = np.load(path, allow_pickle=False)
    if str(z["format"]) not in _HISTORY_FORMATS:
        raise ValueError("not a pgbart tree-history file")
    col_rules = z["rules"] if "rules" in z.files and z["rules"].size else None  # format 1 only
    all_trees = []
    for i in range(int(z["n_chains"])):
        K = int(z[f"c{i}_n_outputs"])
        arrs = {f: z[f"c{i}_{f}"] for f in _HISTORY_FIELDS if f"c{i}_{f}" in z.files}
        if "rule" not in arrs:
            var = arrs["var"].astype(np.int64)
            if col_rules is None:
                arrs["rule"] = np.zeros(var.shape[0], np.int32)
            else:
                cr = np.asarray(col_rules, np.int32)
                if var.size and int(var.max()) >= cr.size:  # a truncated rules array must not be an IndexError
                    raise ValueError(f"tree-history file: a tree splits on column {int(var.max())} but the file lists "
                                     f"the split rules of {cr.size} columns")
                arrs["rule"] = np.where(var >= 0, cr[np.maximum(var, 0)], 0).astype(np.int32)
        sizes = z[f"c{i}_sizes"].tolist()
        forests, t0 = [], 0
        for nt in sizes:
            a, b = int(arrs["node_off"][t0]), int(arrs["node_off"][t0 + nt])
            forests.append(TreeArrays(
                n_outputs=K,
                tree_id=arrs["tree_id"][t0:t0 + nt].astype(np.int32),
                node_off=(arrs["node_off"][t0:t0 + nt + 1] - a).astype(np.int32),
                var=arrs["var"][a:b].astype(np.int32), split=arrs["split"][a:b].astype(np.float64),
                left=arrs["left"][a:b].astype(np.int32), right=arrs["right"][a:b].astype(np.int32),
                count=arrs["count"][a:b].astype(np.int64),
                value=arrs["value"][a:b].reshape(b - a, K).astype(np.float64),
                slope=arrs["slope"].reshape(-1, K)[a:b].astype(np.float64), xbar=arrs["xbar"][a:b].astype(np.float64),
                svar=arrs["svar"][a:b].astype(np.int32), rule=arrs["rule"][a:b].astype(np.int32),
            ))
            t0 += nt
        all_trees.append((forests[0], forests[1:]))
    return all_trees, int(z["m"])


def predict_numpy(trees: TreeArrays, forest_idx: np.ndarray, X: np.ndarray, excluded=None) -> np.ndarray:
    """Slow host restatement of ``pgb_predict`` used by tests only (tiny inputs)."""
    X = np.asarray(X, np.float64)
    n_rows, p = X.shape
    K = trees.n_outputs
    excl = np.zeros(p, bool)
    if excluded is not None:
        excl[list(excluded)] = True
    out = np.zeros((forest_idx.shape[0], K, n_rows))

    def rec(base, k, x, w, acc):
        g = base + k
        while trees.var[g] >= 0:
            j = trees.var[g]
            if excl[j] or np.isnan(x[j]):
                l, r = trees.left[g], trees.right[g]
                cl, cr = float(trees.count[base + l]), float(trees.count[base + r])
                tot = cl + cr
                if not tot > 0:
                    return
                rec(base, l, x, w * (cl / tot), acc)
                rec(base, r, x, w * (cr / tot), acc)
                return
            if trees.rule[g] == _abi.RULE_CONTINUOUS:
                go_left = x[j] <= trees.split[g]
            elif trees.rule[g] == _abi.RULE_ONEHOT:
                go_left = x[j] == trees.split[g]
            else:  # subset: the split value is the bit mask of the categories that go left
                code = min(max(int(x[j]), 0), _abi.SUBSET_BITS - 1)
                go_left = bool((int(trees.split[g]) >> code) & 1)
            k = trees.left[g] if go_left else trees.right[g]
            g = base + k
        leaf = np.array(trees.value[g], dtype=np.float64)
        js = int(trees.svar[g])
        if js >= 0 and not excl[js] and not np.isnan(x[js]):  # linear leaf (a missing regressor: the mean)
            leaf = leaf + trees.slope[g] * (x[js] - trees.xbar[g])
        acc += w * leaf

    for d in range(forest_idx.shape[0]):
        for i in range(n_rows):
            acc = np.zeros(K)
            for ti in forest_idx[d]:
                rec(trees.node_off[ti], 0, X[i], 1.0, acc)
            out[d, :, i] = acc
    return out


class PosteriorSampler:
    """Prediction-only sampler rebuilt from one chain's tree history.

    ``from_history(batches, baseline_forest, m, n_outputs)`` IS the call at reference
    ``utils.py:124-127`` -- nothing else is needed: the split rules travel inside the trees
    (``TreeArrays.rule``).  Draw ``d`` is the baseline forest with batches ``0..d`` applied (each
    batch replaces the trees at its ``tree_id`` slots).  Prediction runs in the HIP library
    (``pgb_predict``); ``backend`` is a test hook (the oracle library as the checker).
    """

    def __init__(self, pool: TreeArrays, forest_idx: np.ndarray, m: int, n_outputs: int, backend=None):
        self.pool = pool
        self.forest_idx = np.ascontiguousarray(forest_idx, dtype=np.int32)
        self.m = int(m)
        self._n_outputs = int(n_outputs)
        self._backend = backend

    @classmethod
    def from_history(cls, batches, baseline_forest: TreeArrays, m: int, n_outputs: int,
                     backend=None) -> "PosteriorSampler":
        batches = _as_list(batches)
        parts = [baseline_forest] + batches
        pool = TreeArrays.concat(parts)
        cur = np.empty(m, np.int64)
        cur[baseline_forest.tree_id] = np.arange(baseline_forest.n_trees)
        table = np.empty((len(batches), m), np.int32)
        off = baseline_forest.n_trees
        for d, b in enumerate(batches):
            cur[b.tree_id] = off + np.arange(b.n_trees)
            off += b.n_trees
            table[d] = cur
        return cls(pool, table, m, n_outputs, backend=backend)

    @property
    def n_draws(self) -> int:
        return int(self.forest_idx.shape[0])

    @property
    def n_outputs(self) -> int:
        return self._n_outputs

    def _get_backend(self):
        if self._backend is None:
            from .sampler import default_backend

            self._backend = default_backend()
        return self._backend

    def resident_rows(self, X):
        """Upload ``X`` once; the returned handle is accepted by :meth:`sample_posterior` in place of the
        matrix (backends without device memory hand the matrix back)."""
        be = self._get_backend()
        X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
        if X.ndim == 1:
            X = X[:, None]
        return be.mem.from_host(X) if hasattr(be.mem, "is_resident") else X

    def sample_posterior(self, X, draw_indices, excluded=None) -> np.ndarray:
        be = self._get_backend()
        # `X` may already be resident in HBM (``resident_rows``): sweeps that predict on the same rows
        # again and again -- variable importance, partial dependence -- then upload them once
        resident = getattr(be.mem, "is_resident", lambda a: False)(X)
        if not resident:
            X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
            if X.ndim == 1:
                X = X[:, None]
        n_rows, p = (int(v) for v in X.shape)
        idx = np.asarray(draw_indices, dtype=np.int64)
        fidx = np.ascontiguousarray(self.forest_idx[idx], dtype=np.int32)
        excl = np.ascontiguousarray(np.asarray([] if excluded is None else excluded, dtype=np.int32))
        K = self._n_outputs
        xd = X if resident else be.mem.from_host(X)
        outd = be.mem.empty((fidx.shape[0] * K * n_rows,), np.float64)
        carr = self.pool.as_c()
        rc = be.lib.lib.pgb_predict(
            C.byref(carr), fidx.ctypes.data, fidx.shape[0], self.m, be.mem.ptr(xd), n_rows, p, p,
            excl.ctypes.data if excl.size else None, int(excl.size),
            be.mem.ptr(outd), be.mem.stream_ptr,
        )
        be.lib.check(rc, "pgb_predict")
        return be.mem.to_host(outd).reshape(fidx.shape[0], K, n_rows)
