#!/usr/bin/env python3
"""Generate tests/golden/utils_glue.json by RUNNING the reference's own posterior-sampling glue and
variable-importance helpers.

`pymc_bart/utils.py` cannot be imported here (it imports pymc / pytensor / numba / arviz at module top), but
`_sample_posterior` (:26-71), `_MultiChainSampler` (:74-107), `generate_sequences` (:1330-1336) and `pearsonr2`
(:1339-1346, minus its numba decorator) are plain NumPy: their definitions are extracted with `ast` and executed in
isolation against a deterministic stand-in for the native `PosteriorSampler` (the contract at :60-71, 93-107:
`.n_draws`, `.n_outputs`, `.sample_posterior(X, draw_indices, excluded) -> (len(idx), n_outputs, n_rows)`).
Only inputs and outputs are committed (data, not source).  Run in the build container only -- /root/reference
does not exist on the GPU box."""
import ast
import json
import os

import numpy as np

REF = "/root/reference/pymc_bart/utils.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "utils_glue.json")


class FakeChain:
    """Stand-in for one chain's native PosteriorSampler: out[d, k, r] is a known function of (chain, draw, output,
    row, excluded), so that any mistake in draw selection, chain dispatch or reshaping changes the numbers."""

    def __init__(self, chain, n_draws, n_outputs):
        self.chain, self.n_draws, self.n_outputs = chain, n_draws, n_outputs
        self.calls = []

    def sample_posterior(self, X, draw_indices, excluded):
        X = np.asarray(X, dtype=np.float64)
        idx = [int(i) for i in draw_indices]
        self.calls.append(idx)
        ex = 0.0 if not excluded else 0.5 * sum(int(e) + 1 for e in excluded)
        out = np.empty((len(idx), self.n_outputs, X.shape[0]))
        for a, d in enumerate(idx):
            for k in range(self.n_outputs):
                out[a, k] = 1000.0 * self.chain + 10.0 * d + k + ex + 0.001 * X.sum(axis=1)
        return out


def load_reference():
    tree = ast.parse(open(REF).read())
    names = {"_sample_posterior", "_MultiChainSampler", "generate_sequences", "pearsonr2"}
    wanted = []
    for n in tree.body:
        if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names:
            if isinstance(n, ast.FunctionDef):
                n.decorator_list = []          # pearsonr2: the numba decorator
                n.returns = None
                for a in n.args.args + n.args.kwonlyargs:
                    a.annotation = None        # annotations name types of modules that are not importable here
            wanted.append(n)
    assert {n.name for n in wanted} == names
    ns = {"np": np}
    exec(compile(ast.Module(body=wanted, type_ignores=[]), REF, "exec"), ns)  # noqa: S102
    return ns


def main():
    ns = load_reference()
    sp, MCS = ns["_sample_posterior"], ns["_MultiChainSampler"]
    cases = []
    rng0 = np.random.default_rng(20261002)
    specs = [
        dict(chains=[5], K=1, rows=4, p=3, size=None, excluded=None, seed=1),
        dict(chains=[5], K=1, rows=4, p=3, size=7, excluded=None, seed=2),
        dict(chains=[6, 4], K=1, rows=5, p=2, size=(3, 2), excluded=[1], seed=3),          # tests/test_utils.py:24-32
        dict(chains=[3, 3, 5], K=3, rows=2, p=4, size=(2, 50), excluded=[0, 2], seed=4),    # tests/test_bart.py:163-164
        dict(chains=[100, 100], K=2, rows=3, p=2, size=(2, 100), excluded=None, seed=5),    # tests/test_bart.py:84-104
        dict(chains=[1, 1, 1, 1], K=1, rows=1, p=1, size=9, excluded=[], seed=6),
    ]
    for s in specs:
        X = np.round(rng0.normal(size=(s["rows"], s["p"])), 3)
        chains = [FakeChain(c, nd, s["K"]) for c, nd in enumerate(s["chains"])]
        sampler = MCS(chains)
        out = sp(sampler, X, np.random.default_rng(s["seed"]), size=s["size"], excluded=s["excluded"])
        cases.append({**{k: v for k, v in s.items()}, "X": X.tolist(), "n_draws": sampler.n_draws,
                      "calls": [c.calls for c in chains], "shape": list(out.shape), "out": np.asarray(out).ravel().tolist()})
    # a list of samplers side by side (utils.py:66-67: outputs concatenated along the outputs axis)
    X = np.round(rng0.normal(size=(3, 2)), 3)
    group = [MCS([FakeChain(0, 4, 1)]), MCS([FakeChain(7, 4, 2)])]
    out = sp(group, X, np.random.default_rng(9), size=5, excluded=None)
    side = {"X": X.tolist(), "shape": list(out.shape), "out": np.asarray(out).ravel().tolist(), "seed": 9, "size": 5}
    seqs = [{"n_vars": a, "i_var": b, "include": c, "out": [list(t) for t in ns["generate_sequences"](a, b, c)]}
            for a, b, c in [(4, 0, []), (4, 2, [1]), (5, 1, []), (6, 3, [0, 4]), (3, 2, [0, 1, 2])]]
    pr = []
    for sd in range(4):
        r = np.random.default_rng(sd)
        A, B = r.normal(size=(6, 5)), r.normal(size=(6, 5))
        B = B + (0.5 * sd) * A
        pr.append({"A": A.tolist(), "B": B.tolist(), "out": float(ns["pearsonr2"](A, B))})
    json.dump({"source": "pymc_bart/utils.py:26-107, 1330-1346 executed via ast against a deterministic stand-in "
                         "for the native PosteriorSampler (tests/golden/make_utils_golden.py)",
               "sample_posterior": cases, "sampler_list": side, "generate_sequences": seqs, "pearsonr2": pr},
              open(OUT, "w"), indent=0)
    print("wrote", OUT, len(cases), "cases", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
