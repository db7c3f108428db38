#!/usr/bin/env python3
"""Soak: one long chain driven through EVERY way the ABI can step it, in random alternation -- host-output asteps,
device-output steps, asynchronous batches of 1..7 steps, tune flips, sigma changes, checkpoint -> fresh sampler ->
restore, exports -- on the HIP backend and on the oracle, compared bit for bit after every call.  The credit
feeding, the step-complete word and the second output stream are host-side state machines: this is the test that
they never lose a slot, a flag or a result over thousands of steps.
usage (GPU box): python tools/soak_parity.py [SECONDS] [SEED]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _oracle import oracle_backend  # noqa: E402
from pymc_bart_amd.sampler import PyBartSettings, PySampler, default_backend  # noqa: E402

def run(budget=60.0, seed=1, backs=None):
    rng = np.random.default_rng(seed)
    n, p, m, P = 6000, 6, 40, 20
    X = rng.normal(size=(n, p))
    X[rng.random(n) < 0.05, 2] = np.nan
    Y = np.sin(X[:, 0]) + 0.5 * (np.nan_to_num(X[:, 1]) > 0) + rng.normal(0, 0.3, n)
    st = PyBartSettings.from_data(X, Y, m=m, num_particles=P, seed=seed, batch=(0.2, 0.1))
    rules, prior = np.zeros(p, np.int32), np.ones(p)
    if backs is None:
        backs = {"hip": default_backend(0), "oracle": oracle_backend()}
    S = {k: PySampler(st, X, Y, rules, prior, backend=b) for k, b in backs.items()}
    for s in S.values():
        s.set_likelihood([1.0])
    t0 = time.time()
    calls = steps = 0
    tune = True
    kinds = {}
    while time.time() - t0 < budget:
        op = rng.choice(["host", "dev", "async", "tune", "sigma", "ckpt", "export"], p=[0.4, 0.15, 0.2, 0.05, 0.08, 0.05, 0.07])
        kinds[op] = kinds.get(op, 0) + 1
        calls += 1
        if op == "host":
            a, va = S["hip"].step(tune)
            b, vb = S["oracle"].step(tune)
            assert np.array_equal(a, b) and np.array_equal(va, vb), f"host step {steps} differs"
            steps += 1
        elif op == "dev":
            S["hip"].step(tune, fetch=False)
            S["oracle"].step(tune, fetch=False)
            a = backs["hip"].mem.to_host(S["hip"].sum_trees_device())
            assert np.array_equal(a, np.asarray(S["oracle"].sum_trees_device())), f"device step {steps} differs"
            steps += 1
        elif op == "async":
            k = int(rng.integers(1, 8))
            for s in S.values():
                s.step_async(tune, k)
            ca, cb = S["hip"].sync(), S["oracle"].sync()
            for key in ("particle_steps", "tree_updates", "rows_touched", "rounds", "partitions", "saturations"):
                assert ca[key] == cb[key], (key, ca[key], cb[key])
            steps += k
        elif op == "tune":
            tune = not tune
        elif op == "sigma":
            sg = float(rng.uniform(0.2, 2.0))
            for s in S.values():
                s.set_likelihood([sg])
        elif op == "ckpt":
            # the image belongs to no backend (include/pgbart_image.h): every other time the two chains swap images
            blobs = {k: S[k].checkpoint() for k in backs}
            cross = bool(rng.integers(0, 2)) and len(backs) == 2
            names = list(backs)
            for i, (k, b) in enumerate(backs.items()):
                S[k] = PySampler(st, X, Y, rules, prior, backend=b)
                S[k].restore(blobs[names[1 - i]] if cross else blobs[k])
        else:
            for which in (0, 1):
                assert S["hip"].export_trees(which).raw == S["oracle"].export_trees(which).raw, f"export {which} differs"
        sa, sb = S["hip"].state(), S["oracle"].state()
        assert np.array_equal(sa["leaf_sd"], sb["leaf_sd"]) and sa["iter"] == sb["iter"] and sa["lower"] == sb["lower"]
    return calls, steps, kinds


if __name__ == "__main__":
    _b = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    _s = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    _t = time.time()
    _calls, _steps, _kinds = run(_b, _s)
    print(f"soak: {_calls} calls, {_steps} asteps in {time.time() - _t:.0f} s, 0 mismatches; calls by kind: "
          f"{ {str(k): v for k, v in _kinds.items()} }")

