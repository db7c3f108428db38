"""Upstream-SEMANTICS mirror of PGBART in plain NumPy (test infrastructure, CPU only, small problems).

SURVEY.md section 7 step 2 / Appendix A: the algorithm of pymc-bart <= 0.12 `pgbart.py` / `tree.py` as recalled there,
WITHOUT the deviations DESIGN.md section 0 lists for the oracle and the HIP backend:

  * a SEQUENTIAL NumPy ``Generator`` stream (deviation 1);
  * a fresh particle keeps ``log_weight = 0`` until it grows (deviation 2);
  * the final tree is ``particles[systematic(W)[int(U P)]]`` (deviation 3);
  * ``update_weight`` is the FULL-n log-likelihood of ``sum_trees_noi + tree.predict()`` (deviation 4);
  * trees are unbounded (deviation 5);
  * rows with a missing split value are filtered BEFORE the ">= 2 candidates" test and the draw (deviation 6);
  * cumulative weights / the split-variable CDF are serial float sums (deviation 7);
  * a failed grow leaves the node untouched (deviation 8);
  * ``leaf_sd`` adopts whatever the running sd returns from the third update on (deviation 12).

It shares NO code with `oracle/` or the product: different data structure (row-index arrays per leaf, as upstream),
different RNG, float sums.  `tests/test_upstream_mirror.py` runs the oracle and this mirror over many seeds and
compares what a user sees -- fit, variable inclusion, tree sizes, tuned leaf_sd -- within Monte-Carlo error.
Reference call sites of the behaviour mirrored: tests/test_bart.py:44-64 (VI), :67-81 (missing values), :140-164
(3-class softmax), bart.py:107-113 (tree prior).
"""
from __future__ import annotations

import numpy as np


class _Tree:
    """Heap-indexed binary tree; a leaf owns its row indices (upstream `idx_data_points`)."""

    def __init__(self, K, n, leaf_value):
        self.K = K
        self.n = n
        self.nodes = {0: {"leaf": True, "value": np.full(K, leaf_value), "idx": np.arange(n), "var": -1, "split": np.nan}}

    def copy(self):
        t = _Tree.__new__(_Tree)
        t.K, t.n = self.K, self.n
        t.nodes = {i: dict(nd) for i, nd in self.nodes.items()}  # (arrays are never modified in place)
        return t

    def predict(self):
        out = np.zeros((self.K, self.n))
        for nd in self.nodes.values():
            if nd["leaf"]:
                out[:, nd["idx"]] = nd["value"][:, None]
        return out

    def split_vars(self):
        return [nd["var"] for nd in self.nodes.values() if not nd["leaf"]]

    def n_leaves(self, nonempty=False):
        return sum(1 for nd in self.nodes.values() if nd["leaf"] and (not nonempty or nd["idx"].size > 0))


class _Particle:
    def __init__(self, tree, expansion=(0,)):
        self.tree = tree
        self.expansion = list(expansion)
        self.log_weight = 0.0  # upstream: a fresh particle carries 0 until update_weight runs for it

    def copy(self):
        p = _Particle(self.tree.copy(), self.expansion)
        p.log_weight = self.log_weight
        return p


class UpstreamMirror:
    """One chain.  ``family``: "normal" (sigma fixed) or "categorical" (K-class softmax, Y = class index)."""

    def __init__(self, X, Y, m=50, num_particles=10, alpha=0.95, beta=2.0, family="normal", sigma=1.0, K=1,
                 split_rules=None, seed=0, batch=(0.1, 0.1), fresh_weight="zero"):
        # fresh_weight: "zero" = upstream as recalled (a fresh particle keeps log_weight 0 until it grows);
        # "stump" = DESIGN.md deviation 2 switched ON in this mirror (it carries the likelihood of its stump), every
        # other upstream semantic kept -- what isolates that one deviation from the other eleven
        self.fresh_weight = fresh_weight
        self.X = np.asarray(X, float)
        self.Y = np.asarray(Y, float)
        self.n, self.p = self.X.shape
        self.m, self.P, self.K = int(m), int(num_particles), int(K)
        self.family, self.sigma = family, float(sigma)
        self.rules = list(split_rules) if split_rules else ["ContinuousSplit"] * self.p
        self.rng = np.random.default_rng(seed)
        self.alpha_vec = np.ones(self.p)
        self.ssv = self.alpha_vec.cumsum() / self.alpha_vec.sum()
        binary = np.all(np.isin(self.Y, (0.0, 1.0)))
        self.leaf_sd = np.full(self.K, 3.0 / np.sqrt(self.m) if binary else self.Y.std() / np.sqrt(self.m))
        prior, d = [], 0
        while True:  # bart.py:107-113
            q = 1.0 - alpha * (1.0 + d) ** (-beta)
            prior.append(q)
            d += 1
            if q >= 0.9999:
                break
        prior.append(1.0)
        self.prior_leaf = np.array(prior)
        init = self.Y.mean()
        self.sum_trees = np.full((self.K, self.n), init)
        self.trees = [_Particle(_Tree(self.K, self.n, init / self.m), ()) for _ in range(self.m)]
        self.batch = tuple(max(1, int(self.m * b)) if b < 1 else int(b) for b in batch)
        self.lower = 0
        self.iter = 0
        self.tune = True
        self.rs_count = 0
        self.rs_mean = np.zeros((self.K, self.n))
        self.rs_m2 = np.zeros((self.K, self.n))

    # ---- likelihood of a full prediction (K x n): upstream evaluates the model's logp for every particle
    def loglik(self, pred):
        if self.family == "normal":
            r = (self.Y - pred[0]) / self.sigma
            return float(-0.5 * (r @ r))
        z = pred - pred.max(axis=0)
        lse = np.log(np.exp(z).sum(axis=0))
        return float((z[self.Y.astype(int), np.arange(self.n)] - lse).sum())

    # ---- [U] systematic resampling: u = (U + arange(L)) / L, inverse-CDF walk
    def systematic(self, w):
        L = len(w)
        u = (self.rng.random() + np.arange(L)) / L
        c = np.cumsum(w)
        out, j = np.empty(L, int), 0
        for i in range(L):
            while j < L - 1 and u[i] > c[j]:
                j += 1
            out[i] = j
        return out

    @staticmethod
    def normalize(lw):
        lw = np.asarray(lw, float)
        w = np.exp(lw - lw.max()) + 1e-12
        return w / w.sum()

    def draw_leaf_value(self, rows):
        if rows.size == 0:
            return np.zeros(self.K)
        eps = self.rng.normal(size=self.K) * self.leaf_sd
        return self.sum_trees[:, rows].mean(axis=1) / self.m + eps

    def grow(self, pt, i):
        nd = pt.tree.nodes[i]
        j = int(np.searchsorted(self.ssv, self.rng.random()))
        j = min(j, self.p - 1)
        rows = nd["idx"]
        x = self.X[rows, j]
        keep = ~np.isnan(x)  # missing values are filtered BEFORE the candidate test
        rows, x = rows[keep], x[keep]
        if x.size <= 1:
            return False
        v = x[int(self.rng.random() * x.size)]
        left = (x <= v) if self.rules[j] == "ContinuousSplit" else (x == v)
        kids = (rows[left], rows[~left])
        for c, r in zip((2 * i + 1, 2 * i + 2), kids):
            pt.tree.nodes[c] = {"leaf": True, "value": self.draw_leaf_value(r), "idx": r, "var": -1, "split": np.nan}
        pt.tree.nodes[i] = {"leaf": False, "value": nd["value"], "idx": nd["idx"], "var": j, "split": v}
        pt.expansion += [2 * i + 1, 2 * i + 2]
        return True

    def sample_tree(self, pt):
        if not pt.expansion:
            return False
        i = pt.expansion.pop(0)
        depth = int(np.floor(np.log2(i + 1)))
        q = self.prior_leaf[min(depth, len(self.prior_leaf) - 1)]
        if q < self.rng.random():
            return self.grow(pt, i)
        return False

    def astep(self):
        vi = np.zeros(self.p, int)
        upper = min(self.lower + self.batch[0 if self.tune else 1], self.m)
        ids = range(self.lower, upper)
        self.lower = upper if upper < self.m else 0
        for t in ids:
            self.iter += 1
            p0 = self.trees[t]
            noi = self.sum_trees - p0.tree.predict()
            p0.log_weight = self.loglik(noi + p0.tree.predict())
            p0.expansion = []
            parts = [p0] + [_Particle(_Tree(self.K, self.n, self.Y.mean() / self.m)) for _ in range(self.P - 1)]
            if self.fresh_weight == "stump":
                lw = self.loglik(noi + parts[1].tree.predict())
                for pt in parts[1:]:
                    pt.log_weight = lw
            while True:
                stop = True
                for pt in parts[1:]:
                    if self.sample_tree(pt):
                        pt.log_weight = self.loglik(noi + pt.tree.predict())
                    if pt.expansion:
                        stop = False
                if stop:
                    break
                idx = self.systematic(self.normalize([q.log_weight for q in parts[1:]])) + 1
                seen, new = set(), []
                for k in idx:
                    new.append(parts[k].copy() if k in seen else parts[k])
                    seen.add(k)
                parts[1:] = new
            w = self.normalize([q.log_weight for q in parts])
            new = parts[self.systematic(w)[int(self.rng.random() * self.P)]]
            self.trees[t] = new
            pred = new.tree.predict()
            self.sum_trees = noi + pred
            if self.tune:
                if self.iter > self.m:
                    self.ssv = self.alpha_vec.cumsum() / self.alpha_vec.sum()
                for v in new.tree.split_vars():
                    self.alpha_vec[v] += 1
                self.rs_count += 1  # [U] RunningSd.update (Welford per row), mean over the rows
                delta = pred - self.rs_mean
                self.rs_mean += delta / self.rs_count
                self.rs_m2 += delta * (pred - self.rs_mean)
                sd = np.sqrt(self.rs_m2 / self.rs_count).mean(axis=1)
                if self.iter > 2:
                    self.leaf_sd = sd
            else:
                for v in new.tree.split_vars():
                    vi[v] += 1
        return self.sum_trees.copy(), vi

    def leaves_per_tree(self):
        return np.array([q.tree.n_leaves() for q in self.trees])
