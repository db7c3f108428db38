#!/usr/bin/env python3
"""Self-check for a maintainer on a box that HAS PyMC (the build image never will): the assertions of the
reference's own sampler tests -- /root/reference/tests/test_bart.py:44-64 (variable inclusion), :67-81 (missing
data), :84-104 (shared X, posterior predictive shapes), :107-123 (shape=(2, n)), :140-164 (categorical, three
split rules), :167-208 (two BART variables, automatic step assignment), :211-241 (manual ``PGBART([mu],
num_particles=5)``), :244-256 (mutable named dim) -- run through ``pymc_bart_amd`` on cuda:0.

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


def check_vi(pm, pmb, amd, q, response="constant"):  # tests/test_bart.py:44-64
    from pymc_bart.utils import _decode_vi

    X = np.random.normal(0, 1, size=(250, 3))
    Y = np.random.normal(0, 1, size=250)
    X[:, 0] = np.random.normal(Y, 0.1)
    with pm.Model():
        mu = pmb.BART("mu", X, Y, m=10, response=response)
        sigma = pm.HalfNormal("sigma", 1)
        pm.Normal("y", mu, sigma, observed=Y)
        idata = pm.sample(tune=200 // q, draws=200 // q, random_seed=3415, progressbar=False)
    vi_vals = idata["sample_stats"]["variable_inclusion"].values.ravel()
    var_imp = np.array([_decode_vi(val, 3) for val in vi_vals]).sum(axis=0)
    var_imp = var_imp / var_imp.sum()
    assert var_imp[0] > var_imp[1:].sum(), var_imp
    np.testing.assert_almost_equal(var_imp.sum(), 1)


def check_vi_linear(pm, pmb, amd, q):
    check_vi(pm, pmb, amd, q, response="linear")


def check_missing(pm, pmb, amd, q):  # :67-81
    X = np.random.normal(0, 1, size=(50, 2))
    Y = np.random.normal(0, 1, size=50)
    X[10:20, 0] = np.nan
    with pm.Model():
        mu = pmb.BART("mu", X, Y, m=10)
        sigma = pm.HalfNormal("sigma", 1)
        pm.Normal("y", mu, sigma, observed=Y)
        pm.sample(tune=100 // q, draws=100 // q, chains=1, random_seed=3415, progressbar=False)


def check_shared(pm, pmb, amd, q):  # :84-104
    X = np.random.normal(0, 1, size=(50, 2))
    Y = np.random.normal(0, 1, size=50)
    d = 100 // q
    with pm.Model():
        data_X = pm.Data("data_X", X)
        mu = pmb.BART("mu", data_X, Y, m=2)
        sigma = pm.HalfNormal("sigma", 1)
        pm.Normal("y", mu, sigma, observed=Y, shape=mu.shape)
        idata = pm.sample(tune=d, draws=d, chains=2, random_seed=3415, progressbar=False)
        ppc = pm.sample_posterior_predictive(idata, progressbar=False)
        pm.set_data({"data_X": X[:3]})
        ppc2 = pm.sample_posterior_predictive(idata, sample_vars=["mu", "y"], progressbar=False)
    assert ppc.posterior_predictive["y"].shape == (2, d, 50)
    assert ppc2.posterior_predictive["y"].shape == (2, d, 3)


def check_shape(pm, pmb, amd, q):  # :107-123
    X = np.random.normal(0, 1, size=(250, 3))
    Y = np.random.normal(0, 1, size=250)
    with pm.Model() as model:
        w = pmb.BART("w", X, Y, m=2, shape=(2, 250))
        pm.Normal("y", w[0], pm.math.abs(w[1]), observed=Y)
        idata = pm.sample(tune=50, draws=10, random_seed=3415, progressbar=False)
    assert model.initial_point()["w"].shape == (2, 250)
    assert idata.posterior.coords["w_dim_0"].data.size == 2
    assert idata.posterior.coords["w_dim_1"].data.size == 250


def check_categorical(pm, pmb, amd, q):  # :140-164
    Y = np.array([0, 0, 0, 1, 1, 1, 2, 2, 2])
    rng = np.random.default_rng(12345)
    X = np.concatenate([Y[:, None], rng.integers(0, 6, size=(9, 4))], axis=1)
    for rule in ("ContinuousSplit", "OneHotSplit"):
        with pm.Model():
            lo = pmb.BART("logodds", X, Y, m=2, shape=(3, 9), split_rules=[rule] * 5)
            pm.Categorical("y", p=pm.math.softmax(lo.T, axis=-1), observed=Y)
            idata = pm.sample(tune=600, draws=600, random_seed=3415, progressbar=False)
            idata = pm.sample_posterior_predictive(idata, predictions=True, extend_inferencedata=True,
                                                   random_seed=3415, progressbar=False)
        assert (idata.predictions.y.median(["chain", "draw"]) == Y).all(), rule
        assert pmb.compute_variable_importance(idata, bartrv=lo, X=X)["preds"].shape == (5, 50, 9, 3)


def check_two_bart_auto(pm, pmb, amd, q):  # :167-208
    X1 = np.random.normal(0, 1, size=(50, 2))
    X2 = np.random.normal(0, 1, size=(50, 3))
    Y = np.random.normal(0, 1, size=50)
    Y1 = X1[:, 0] + np.random.normal(0, 0.1, size=50)
    Y2 = X2[:, 0] + X2[:, 1] + np.random.normal(0, 0.1, size=50)
    with pm.Model() as model:
        mu1 = pmb.BART("mu1", X1, Y1, m=5)
        mu2 = pmb.BART("mu2", X2, Y2, m=5)
        sigma = pm.HalfNormal("sigma", 1)
        pm.Normal("y", mu1 + mu2, sigma, observed=Y)
        idata = pm.sample(tune=50, draws=50, chains=1, random_seed=3415, progressbar=False)
        assert mu1.owner.op.all_trees is not mu2.owner.op.all_trees
        assert idata.posterior["mu1"].shape == (1, 50, 50) and idata.posterior["mu2"].shape == (1, 50, 50)
        vi = pmb.compute_variable_importance(idata, mu1, X1, model=model)
        assert vi["labels"].shape == (2,) and vi["preds"].shape == (2, 50, 50) and vi["preds_all"].shape == (50, 50)
        vt = pmb.get_variable_inclusion(idata, X1, model=model, bart_var_name="mu1")
        assert vt[0].shape == (2,) and len(vt[1]) == 2 and isinstance(vt[1][0], str)


def check_two_bart_manual(pm, pmb, amd, q):  # :211-241
    X1 = np.random.normal(0, 1, size=(30, 2))
    X2 = np.random.normal(0, 1, size=(30, 2))
    Y = np.random.normal(0, 1, size=30)
    Y1 = X1[:, 0] + np.random.normal(0, 0.1, size=30)
    Y2 = X2[:, 1] + np.random.normal(0, 0.1, size=30)
    with pm.Model():
        mu1 = pmb.BART("mu1", X1, Y1, m=3)
        mu2 = pmb.BART("mu2", X2, Y2, m=3)
        sigma = pm.HalfNormal("sigma", 1)
        pm.Normal("y", mu1 + mu2, sigma, observed=Y)
        step1 = amd.PGBART([mu1], num_particles=5)
        step2 = amd.PGBART([mu2], num_particles=5)
        idata = pm.sample(tune=20, draws=20, chains=1, step=[step1, step2], random_seed=3415, progressbar=False)
    assert idata.posterior["mu1"].shape == (1, 20, 30) and idata.posterior["mu2"].shape == (1, 20, 30)


def check_named_dim(pm, pmb, amd, q):  # :244-256
    rng = np.random.default_rng(0)
    N = 50
    X, Y = rng.normal(size=(N, 2)), rng.normal(size=N)
    with pm.Model(coords={"obs": np.arange(N), "feature": ["a", "b"]}):
        x = pm.Data("x", X, dims=("obs", "feature"))
        mu = pmb.BART("mu", X=x, Y=Y, m=10, dims="obs")
        pm.Normal("y", mu=mu, sigma=1.0, observed=Y, dims="obs")
        pm.sample(tune=20, draws=20, chains=1, progressbar=False)


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


CHECKS = [("registration", check_registration), ("variable_inclusion", check_vi),
          ("variable_inclusion_linear", check_vi_linear), ("missing_data", check_missing),
          ("shared_variable", check_shared), ("shape_2xn", check_shape), ("categorical", check_categorical),
          ("two_bart_auto", check_two_bart_auto), ("two_bart_manual_step", check_two_bart_manual),
          ("mutable_named_dim", check_named_dim)]


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
