#!/usr/bin/env python3
"""Self-check for a maintainer on a box that HAS PyMC (the build image never will): the assertions of the
reference's own sampler tests -- /root/reference/tests/test_bart.py:44-64 (variable inclusion), :67-81 (missing
data), :84-104 (shared X, posterior predictive shapes), :107-123 (shape=(2, n)), :140-164 (categorical, three
split rules), :167-208 (two BART variables, automatic step assignment), :211-241 (manual ``PGBART([mu],
num_particles=5)``), :244-256 (mutable named dim) -- run through ``pymc_bart_amd`` on cuda:0.  The tests are held
here as a TABLE (shapes, keyword arguments, sampling arguments, named assertions: ``CASES``) and one generic runner
builds each model from its row.

    python tools/pymc_selfcheck.py [--quick] [--only NAME ...] [--semantics]

``--semantics`` (needs the real ``bartrs`` wheel as well): which of this sampler's two modes does ``bartrs`` follow --
the default, or the upstream-semantics switches (``PGBART_SEMANTICS=upstream``: fresh particles at log-weight 0, empty
right leaves of one-hot splits)?  Runs the reference's sampler and both modes on BASELINE's cfg1
problem over 8 seeds and compares leaves per accepted tree (from the ``variable_inclusion`` stat: splits + 1), the
metric the two modes differ most in (3.24 +- 0.03 against 2.41 +- 0.03).

Needs: pymc, pymc_bart (the reference package, unmodified: this script answers its ``import bartrs`` with
``pymc_bart_amd``, which is the one-line change INTEGRATION.md section 3 describes), a MI355X.  Prints one
PASS / FAIL / SKIP line per item plus the versions; exit code = number of failures.  Not part of the test
suite (nothing here can run without PyMC)."""
import argparse
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


REAL_BARTRS = None


def _imports():
    global REAL_BARTRS
    try:  # the reference's own native sampler, if this box has it: kept aside for --semantics
        import importlib

        REAL_BARTRS = importlib.import_module("bartrs")
        for k in [k for k in sys.modules if k == "bartrs" or k.startswith("bartrs.")]:
            del sys.modules[k]
    except Exception:  # noqa: BLE001
        REAL_BARTRS = None
    import pymc_bart_amd

    # the reference imports its native sampler as `bartrs` (pymc_bart/pymc_bart.py:2, __init__.py:15,
    # tests/test_bart.py:4): answer those imports with the MI355X backend
    sys.modules.setdefault("bartrs", pymc_bart_amd)
    sys.modules.setdefault("bartrs.bartrs", pymc_bart_amd)
    import pymc as pm
    import pymc_bart as pmb

    return pm, pmb, pymc_bart_amd


def check_registration(pm, pmb, amd, q):
    assert amd.PGBART in list(pm.STEP_METHODS), "PGBART not in pm.STEP_METHODS after import"
    X, Y = np.random.normal(size=(30, 2)), np.random.normal(size=30)
    with pm.Model():
        mu = pmb.BART("mu", X, Y, m=3)
        pm.Normal("y", mu, 1.0, observed=Y)
        comp = amd.PGBART.competence(mu, has_grad=False)
    assert "IDEAL" in str(comp).upper() or int(comp) >= 3, comp


# The reference's sampler tests as DATA: one row per test of /root/reference/tests/test_bart.py -- which lines, the
# shapes of the synthetic inputs, the BART variables (name, which X / Y, keyword arguments), the observed distribution,
# the pm.sample arguments (`quick` halves the ones marked True) and the assertions to make, by name.  One generic
# runner builds the model from the row; nothing of the reference's test source is repeated here.
#   data:      n rows; xs = column counts of the design matrices; "tie" = X[:, 0] is Y plus noise; "nan" = a block of
#              missing values in X[:, 0]; "classes" = the 9-row three-class problem of :140-164; "sum" = Y_k built
#              from the leading columns of X_k
#   observed:  "normal_hn" Normal(sum of the BART variables, HalfNormal(1)); "normal_1" Normal(., 1);
#              "meanscale" Normal(w[0], |w[1]|); "softmax" Categorical(softmax)
CASES = [
    dict(name="variable_inclusion", ref=":44-64", data=dict(n=250, xs=[3], tie=True), bart=[dict(m=10)],
         observed="normal_hn", sample=dict(tune=200, draws=200), quick=True, expect=["vi_first_dominates"]),
    dict(name="variable_inclusion_linear", ref=":44-64 (response=linear)", data=dict(n=250, xs=[3], tie=True),
         bart=[dict(m=10, response="linear")], observed="normal_hn", sample=dict(tune=200, draws=200), quick=True,
         expect=["vi_first_dominates"]),
    dict(name="missing_data", ref=":67-81", data=dict(n=50, xs=[2], nan=(10, 20)), bart=[dict(m=10)],
         observed="normal_hn", sample=dict(tune=100, draws=100, chains=1), quick=True, expect=[]),
    dict(name="shared_variable", ref=":84-104", data=dict(n=50, xs=[2]), bart=[dict(m=2)], shared_x=True,
         observed="normal_hn", sample=dict(tune=100, draws=100, chains=2), quick=True, expect=["ppc_shapes"]),
    dict(name="shape_2xn", ref=":107-123", data=dict(n=250, xs=[3]), bart=[dict(m=2, shape=(2, 250), name="w")],
         observed="meanscale", sample=dict(tune=50, draws=10), expect=["coords_2xn"]),
    dict(name="categorical", ref=":140-164", data=dict(classes=True), per_rule=("ContinuousSplit", "OneHotSplit"),
         bart=[dict(m=2, shape=(3, 9), name="logodds")], observed="softmax", sample=dict(tune=600, draws=600),
         expect=["classes_recovered", "vi_preds_shape"]),
    dict(name="two_bart_auto", ref=":167-208", data=dict(n=50, xs=[2, 3], sum=True), bart=[dict(m=5, name="mu1"),
         dict(m=5, name="mu2")], observed="normal_hn", sample=dict(tune=50, draws=50, chains=1),
         expect=["separate_histories", "posterior_shapes", "vi_utils"]),
    dict(name="two_bart_manual_step", ref=":211-241", data=dict(n=30, xs=[2, 2], sum=True), bart=[dict(m=3, name="mu1"),
         dict(m=3, name="mu2")], observed="normal_hn", manual_steps=5, sample=dict(tune=20, draws=20, chains=1),
         expect=["posterior_shapes"]),
    dict(name="mutable_named_dim", ref=":244-256", data=dict(n=50, xs=[2]), bart=[dict(m=10, dims="obs")],
         named_dims=True, observed="normal_1", sample=dict(tune=20, draws=20, chains=1, seedless=True), expect=[]),
]


def _case_data(d):
    if d.get("classes"):
        Y = np.repeat(np.arange(3), 3)
        X = np.concatenate([Y[:, None], np.random.default_rng(12345).integers(0, 6, size=(9, 4))], axis=1)
        return [X], [Y], Y
    n = d["n"]
    Xs = [np.random.normal(0, 1, size=(n, k)) for k in d["xs"]]
    Y = np.random.normal(0, 1, size=n)
    if d.get("tie"):
        Xs[0][:, 0] = np.random.normal(Y, 0.1)
    if d.get("nan"):
        Xs[0][d["nan"][0]: d["nan"][1], 0] = np.nan
    Ys = [Y] * len(Xs)
    if d.get("sum"):  # each BART variable gets a response of its own, built from its leading columns
        Ys = [X[:, : 1 + i].sum(axis=1) + np.random.normal(0, 0.1, size=n) for i, X in enumerate(Xs)]
    return Xs, Ys, Y


def run_case(case, pm, pmb, amd, q, rule=None):
    from pymc_bart.utils import _decode_vi

    Xs, Ys, Y = _case_data(case["data"])
    n = Y.shape[0]
    div = q if case.get("quick") else 1
    sk = dict(case["sample"])
    seedless = sk.pop("seedless", False)
    sk.update(tune=sk["tune"] // div, draws=sk["draws"] // div, progressbar=False)
    if not seedless:
        sk["random_seed"] = 3415
    coords = {"obs": np.arange(n), "feature": ["a", "b"]} if case.get("named_dims") else None
    with pm.Model(coords=coords) as model:
        rvs = []
        for i, kw in enumerate(case["bart"]):
            kw = dict(kw)
            name = kw.pop("name", "mu")
            X_in = Xs[i]
            if case.get("shared_x"):
                X_in = pm.Data("data_X", X_in)
            if case.get("named_dims"):
                X_in = pm.Data("x", X_in, dims=("obs", "feature"))
            if rule is not None:
                kw["split_rules"] = [rule] * Xs[i].shape[1]
            rvs.append(pmb.BART(name, X_in, Ys[i], **kw))
        total = rvs[0] if len(rvs) == 1 else sum(rvs[1:], rvs[0])
        obs = case["observed"]
        if obs == "normal_hn":
            extra = {"shape": total.shape} if case.get("shared_x") else {}
            pm.Normal("y", total, pm.HalfNormal("sigma", 1), observed=Y, **extra)
        elif obs == "normal_1":
            pm.Normal("y", mu=total, sigma=1.0, observed=Y, dims="obs")
        elif obs == "meanscale":
            pm.Normal("y", total[0], pm.math.abs(total[1]), observed=Y)
        else:
            pm.Categorical("y", p=pm.math.softmax(total.T, axis=-1), observed=Y)
        if case.get("manual_steps"):
            sk["step"] = [amd.PGBART([rv], num_particles=case["manual_steps"]) for rv in rvs]
        idata = pm.sample(**sk)
        d, chains = sk["draws"], sk.get("chains")
        for what in case["expect"]:
            if what == "vi_first_dominates":
                vals = idata["sample_stats"]["variable_inclusion"].values.ravel()
                imp = np.array([_decode_vi(v, Xs[0].shape[1]) for v in vals]).sum(axis=0)
                imp = imp / imp.sum()
                assert imp[0] > imp[1:].sum(), imp
            elif what == "ppc_shapes":
                ppc = pm.sample_posterior_predictive(idata, progressbar=False)
                pm.set_data({"data_X": Xs[0][:3]})
                ppc2 = pm.sample_posterior_predictive(idata, sample_vars=["mu", "y"], progressbar=False)
                assert ppc.posterior_predictive["y"].shape == (chains, d, n)
                assert ppc2.posterior_predictive["y"].shape == (chains, d, 3)
            elif what == "coords_2xn":
                assert model.initial_point()["w"].shape == (2, n)
                assert idata.posterior.coords["w_dim_0"].data.size == 2 and idata.posterior.coords["w_dim_1"].data.size == n
            elif what == "classes_recovered":
                idata = pm.sample_posterior_predictive(idata, predictions=True, extend_inferencedata=True,
                                                       random_seed=3415, progressbar=False)
                assert (idata.predictions.y.median(["chain", "draw"]) == Y).all(), rule
            elif what == "vi_preds_shape":
                assert pmb.compute_variable_importance(idata, bartrv=rvs[0], X=Xs[0])["preds"].shape == (5, 50, 9, 3)
            elif what == "separate_histories":
                assert rvs[0].owner.op.all_trees is not rvs[1].owner.op.all_trees
            elif what == "posterior_shapes":
                for kw in case["bart"]:
                    assert idata.posterior[kw["name"]].shape == (chains, d, n)
            elif what == "vi_utils":
                vi = pmb.compute_variable_importance(idata, rvs[0], Xs[0], model=model)
                k = Xs[0].shape[1]
                assert vi["labels"].shape == (k,) and vi["preds"].shape == (k, 50, n) and vi["preds_all"].shape == (50, n)
                vt = pmb.get_variable_inclusion(idata, Xs[0], model=model, bart_var_name=case["bart"][0]["name"])
                assert vt[0].shape == (k,) and len(vt[1]) == k and isinstance(vt[1][0], str)


def _table_check(case):
    def fn(pm, pmb, amd, q):
        for rule in case.get("per_rule", (None,)):
            run_case(case, pm, pmb, amd, q, rule)
    return fn


def leaves_per_tree(idata, p, m, batch=0.1):
    """Mean leaves per accepted tree of the draws: every draw re-samples max(1, int(m * batch)) trees and reports
    their split counts per column; a tree with s splits has s + 1 leaves."""
    from pymc_bart_amd.utils import _decode_vi

    vals = idata["sample_stats"]["variable_inclusion"].values.ravel()
    splits = np.array([sum(_decode_vi(v, p)) for v in vals], float)
    return float(splits.mean() / max(1, int(m * batch)) + 1.0)


def semantics_report(pm, pmb, amd, q):
    """Which mode does bartrs follow?  (Not a PASS / FAIL item: prints the three numbers and the verdict.)"""
    if REAL_BARTRS is None:
        print("SKIP semantics: the real `bartrs` wheel is not importable here")
        return
    rows = {"bartrs": [], "pymc_bart_amd default": [], "pymc_bart_amd upstream": []}
    for seed in range(8 // q):
        rng = np.random.default_rng(1000 + seed)
        X = rng.uniform(0, 1, (500, 10))
        Y = (10 * np.sin(np.pi * X[:, 0] * X[:, 1]) + 20 * (X[:, 2] - 0.5) ** 2 + 10 * X[:, 3] + 5 * X[:, 4]
             + rng.normal(0, 1, 500))
        for key in rows:
            os.environ.pop("PGBART_SEMANTICS", None)
            if key.endswith("upstream"):
                os.environ["PGBART_SEMANTICS"] = "upstream"
            with pm.Model():
                mu = pmb.BART("mu", X, Y, m=50)
                pm.Normal("y", mu, 1.0, observed=Y)
                step = (REAL_BARTRS if key == "bartrs" else amd).PGBART([mu], num_particles=10)
                idata = pm.sample(tune=60, draws=40, chains=1, step=[step], random_seed=seed, progressbar=False)
            rows[key].append(leaves_per_tree(idata, 10, 50))
    os.environ.pop("PGBART_SEMANTICS", None)
    stat = {k: (float(np.mean(v)), float(np.std(v, ddof=1) / np.sqrt(len(v)))) for k, v in rows.items()}
    for k, (mean, se) in stat.items():
        print(f"  leaves per tree  {k:24s} {mean:.3f} +- {se:.3f}")
    ref = stat["bartrs"][0]
    near = min(("default", "upstream"), key=lambda mode: abs(stat[f"pymc_bart_amd {mode}"][0] - ref))
    print(f"semantics: bartrs is closest to the `{near}` mode"
          + ("" if near == "default" else "  (run models with PGBART_SEMANTICS=upstream to match it)"))


CHECKS = [("registration", check_registration)] + [(c["name"], _table_check(c)) for c in CASES]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="halve tune / draws where the reference's counts allow")
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--semantics", action="store_true", help="compare both sampler modes with the real bartrs wheel")
    a = ap.parse_args()
    try:
        pm, pmb, amd = _imports()
    except Exception as e:  # noqa: BLE001
        print(f"SKIP all: cannot import pymc / pymc_bart / pymc_bart_amd here ({type(e).__name__}: {e})")
        return 0
    import pytensor

    print(f"pymc {pm.__version__}  pytensor {pytensor.__version__}  pymc_bart {getattr(pmb, '__version__', '?')}  "
          f"pymc_bart_amd {amd.__version__}  backend {amd._abi.load_hip_library().backend_name}")
    np.random.seed(3415)
    fails = 0
    for name, fn in CHECKS:
        if a.only and name not in a.only:
            continue
        try:
            fn(pm, pmb, amd, 2 if a.quick else 1)
            print(f"PASS {name}")
        except Exception:  # noqa: BLE001
            fails += 1
            print(f"FAIL {name}\n" + "".join("    " + ln for ln in traceback.format_exc(limit=4).splitlines(True)))
    if a.semantics:
        try:
            semantics_report(pm, pmb, amd, 2 if a.quick else 1)
        except Exception:  # noqa: BLE001
            print("semantics: could not be determined\n" + "".join("    " + ln for ln in traceback.format_exc(limit=4).splitlines(True)))
    print(f"{fails} failure(s)")
    return fails


if __name__ == "__main__":
    sys.exit(main())
