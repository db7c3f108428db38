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
OUT_VI = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variable_importance.json")


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


class WeightedChain:
    """Stand-in for a chain's native PosteriorSampler whose predictions depend on the EXCLUDED set the way
    marginalisation does (an excluded column contributes its mean): draw d, output k predicts X_e @ W[d, k] with
    X_e = X but excluded columns replaced by their column mean.  The weights decay over the columns in a fixed,
    shuffled order, so the rankings of compute_variable_importance are not trivial and differ by method."""

    def __init__(self, chain, n_draws, n_outputs, p, seed):
        r = np.random.default_rng(seed)
        self.chain, self.n_draws, self.n_outputs = chain, n_draws, n_outputs
        scale = 2.0 ** -r.permutation(p)
        self.W = scale * (1.0 + 0.3 * r.normal(size=(n_draws, n_outputs, p)))

    def sample_posterior(self, X, draw_indices, excluded):
        Xe = np.array(X, dtype=np.float64, copy=True)
        for j in (excluded or []):
            Xe[:, int(j)] = Xe[:, int(j)].mean()
        return np.stack([self.W[int(d)] @ Xe.T for d in draw_indices])  # (n_idx, K, n_rows)


class FakeVI:
    """What `idata["sample_stats"]["variable_inclusion"]` offers the reference: `.values`, `.sel({dim: i})` and
    `.variable_inclusion_dim_0.size` (the number of BART variables of the model)."""

    def __init__(self, values):
        self.values = np.asarray(values, dtype=object)  # (chain, draw) or (chain, draw, n_bart)

    @property
    def variable_inclusion_dim_0(self):
        return np.empty(self.values.shape[2] if self.values.ndim == 3 else 1)

    def sel(self, d):
        return FakeVI(self.values[:, :, d["variable_inclusion_dim_0"]])


class _Named:
    def __init__(self, name):
        self.name = name


def hdi_standin(x, prob):
    """arviz_stats.array_stats.hdi is an external dependency that is absent here: the narrowest interval holding
    `prob` of the sample (pins that the same r2 sample reaches it; the test's side uses the package's own hdi)."""
    x = np.sort(np.asarray(x, np.float64).ravel())
    n = x.size
    k = max(int(np.floor(prob * n)), 1)
    if k >= n:
        return np.array([x[0], x[-1]])
    i = int(np.argmin(x[k:] - x[: n - k]))
    return np.array([x[i], x[i + k]])


def load_reference_vi(ns):
    """compute_variable_importance (:868-1090), get_variable_inclusion (:747-806), vi_to_kulprit (:1093-1108) and
    the codec (:1368-1398) on top of the glue `ns` already holds."""
    tree = ast.parse(open(REF).read())
    names = {"compute_variable_importance", "get_variable_inclusion", "vi_to_kulprit", "_decode_vi", "_encode_vi"}
    wanted = []
    for n in tree.body:
        if isinstance(n, ast.FunctionDef) and n.name in names:
            n.returns = None
            for a in n.args.args + n.args.kwonlyargs:
                a.annotation = None
            wanted.append(n)
    assert {n.name for n in wanted} == names
    import base64

    ns.update({"base64": base64, "rcParams": {"stats.ci_prob": 0.94},
               "array_stats": type("array_stats", (), {"hdi": staticmethod(hdi_standin)}),
               "_get_posterior_sampler": lambda op: op.sampler})
    exec(compile(ast.Module(body=wanted, type_ignores=[]), REF, "exec"), ns)  # noqa: S102
    return ns


def vi_cases(ns):
    """The reference's ranking functions run on deterministic stand-ins; inputs are re-creatable from the recipe
    (seeds), outputs are stored."""
    enc = ns["_encode_vi"]
    out = []
    specs = [
        dict(name="vi_p5", p=5, rows=30, chains=[6, 5], K=1, method="VI", samples=7, seed=11, ndim=1),
        dict(name="backward_p5", p=5, rows=30, chains=[6, 5], K=1, method="backward", samples=6, seed=12, ndim=1),
        dict(name="vi_k3", p=4, rows=12, chains=[4], K=3, method="VI", samples=5, seed=13, ndim=2),
        dict(name="backward_k2", p=4, rows=12, chains=[3, 3], K=2, method="backward", samples=4, seed=14, ndim=2),
        dict(name="vi_ties", p=6, rows=20, chains=[5], K=1, method="VI", samples=5, seed=15, ndim=1, ties=True),
        dict(name="backward_vi_p5", p=5, rows=30, chains=[6, 5], K=1, method="backward_VI", fixed=2, samples=6, seed=16, ndim=1),
        dict(name="vi_two_bart", p=4, rows=15, chains=[5, 5], K=1, method="VI", samples=5, seed=17, ndim=1, n_bart=2, which=1),
    ]
    for s in specs:
        r = np.random.default_rng(s["seed"])
        X = np.round(r.normal(size=(s["rows"], s["p"])), 3)
        chains = [WeightedChain(c, nd, s["K"], s["p"], 100 * s["seed"] + c) for c, nd in enumerate(s["chains"])]
        sampler = ns["_MultiChainSampler"](chains)
        n_bart = s.get("n_bart", 1)
        n_draws = max(s["chains"])
        counts = r.integers(0, 4, size=(len(s["chains"]), n_draws, n_bart, s["p"]))
        if s.get("ties"):
            counts[..., 1] = counts[..., 3]
            counts[..., 4] = counts[..., 0]
        strings = np.empty((len(s["chains"]), n_draws, n_bart), dtype=object)
        for idx in np.ndindex(strings.shape):
            strings[idx] = enc(counts[idx])
        vals = strings if n_bart > 1 else strings[:, :, 0]
        idata = {"sample_stats": {"variable_inclusion": FakeVI(vals)}}
        names = [f"mu{i}" for i in range(n_bart)]
        rv = type("RV", (), {})()
        rv.owner = type("O", (), {})()
        rv.owner.op = type("Op", (), {"sampler": sampler})()
        rv.name, rv.ndim = names[s.get("which", 0)], s["ndim"]
        model = type("M", (), {"free_RVs": [_Named(nm) for nm in names]})() if n_bart > 1 else None
        rec = {k: v for k, v in s.items()}
        rec.update(X=X.tolist(), strings=vals.tolist(), counts=counts.tolist())
        try:
            res = ns["compute_variable_importance"](idata, rv, X, model=model, method=s["method"],
                                                    fixed=s.get("fixed", 0), samples=s["samples"], random_seed=s["seed"])
            rec["result"] = {"indices": np.asarray(res["indices"]).tolist(), "labels": [str(v) for v in res["labels"]],
                             "r2_mean": res["r2_mean"].tolist(), "r2_hdi": res["r2_hdi"].tolist(),
                             "preds_shape": list(res["preds"].shape), "preds": np.asarray(res["preds"]).ravel().tolist(),
                             "preds_all_shape": list(res["preds_all"].shape),
                             "preds_all": np.asarray(res["preds_all"]).ravel().tolist(),
                             "kulprit": ns["vi_to_kulprit"](res)}
        except Exception as e:  # noqa: BLE001 - "backward_VI": the reference's own code stops here
            rec["raises"] = type(e).__name__
        share, labels = ns["get_variable_inclusion"](idata, X, model=model, bart_var_name=rv.name)
        rec["inclusion"] = {"share": np.asarray(share).tolist(), "labels": list(labels),
                            "kulprit": ns["get_variable_inclusion"](idata, X, model=model, bart_var_name=rv.name,
                                                                    to_kulprit=True)}
        out.append(rec)
    return out


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
    vi = vi_cases(load_reference_vi(ns))
    json.dump({"source": "pymc_bart/utils.py:747-806, 868-1108 executed via ast against WeightedChain / FakeVI "
                         "stand-ins (tests/golden/make_utils_golden.py); arviz-stats' hdi replaced by the narrowest "
                         "interval", "cases": vi}, open(OUT_VI, "w"), indent=0)
    print("wrote", OUT_VI, len(vi), "cases", os.path.getsize(OUT_VI), "bytes",
          {c["name"]: c.get("raises", "ok") for c in vi})


if __name__ == "__main__":
    main()
