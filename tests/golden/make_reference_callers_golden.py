#!/usr/bin/env python3
"""Generate tests/golden/reference_callers.json by RUNNING the reference's own posterior-predictive call path --
`BARTRV.rng_fn` (`pymc_bart/bart.py:47-68`, a classmethod handed `cls`), `_get_posterior_sampler`
(`utils.py:113-130`: `PosteriorSampler.from_history(batches, baseline_forest, op.m, op.n_outputs)`),
`_MultiChainSampler` and `_sample_posterior` (`utils.py:26-107`) -- extracted with `ast` and executed against a
class-attribute op double (`bart.py:141-158`) whose chains were sampled by the CPU oracle, with
`pymc_bart.pymc_bart.PosteriorSampler` bound to this package's class (predicting on the oracle).

Only inputs' recipe (tests/test_reference_callers.py::_data / run_chains, seeded) and outputs are committed: the
posterior-predictive array `rng_fn(cls, rng=default_rng(5), X=X_chain0[:7], size=(11,))` per kind of history.
The GPU box (no reference tree) samples the same chains with libpgbart_hip.so, predicts with k_predict through
the same four-argument call and must reproduce these numbers.  Run in the build container only."""
import json
import os
import sys
from multiprocessing import Manager

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE)]

import test_reference_callers as T  # noqa: E402
from _oracle import oracle_backend  # noqa: E402


def main():
    oracle = oracle_backend()
    glue = T.reference_glue(T.bound_to(oracle))
    sys.modules.update(glue["_modules"])
    cases = {}
    with Manager() as manager:
        for kind in T.KINDS:
            cls, d, mus = T.run_chains(kind, oracle, manager)
            Xnew = np.ascontiguousarray(mus[0][1][:7])
            out = glue["rng_fn"](cls, rng=np.random.default_rng(5), X=Xnew, size=(11,))
            cases[kind] = {"shape": list(out.shape), "rng_fn": np.asarray(out, np.float64).ravel().tolist()}
    json.dump({"source": "pymc_bart/bart.py:47-68 + utils.py:26-130 executed via ast against a class-attribute op "
                         "double; chains and predictions by the CPU oracle (tests/golden/make_reference_callers_golden.py)",
               "cases": cases}, open(T.GOLD_PATH, "w"), indent=0)
    print("wrote", T.GOLD_PATH, os.path.getsize(T.GOLD_PATH), "bytes")


if __name__ == "__main__":
    main()
