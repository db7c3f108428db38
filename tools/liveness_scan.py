#!/usr/bin/env python3
"""Liveness scan of the ALGORITHM (CPU oracle; both backends are bit-identical, so what dies here dies on the
GPU too): random configurations of tests/_cases.random_case are stepped and checked for states a sampler must
never be in -- a constant or non-finite sum_trees, a leaf_sd that is 0 or not finite, fixed-point saturations,
a chain that never moves although a column can be split.  usage: python tools/liveness_scan.py FIRST_SEED COUNT"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _cases import random_case  # noqa: E402
from _oracle import oracle_backend  # noqa: E402
from pymc_bart_amd.sampler import PyBartSettings, PySampler  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
be = oracle_backend()
bad = {}
for seed in range(first, first + count):
    c = random_case(seed)
    X, Y = c["X"], c["Y"]
    if X.shape[0] < 17:
        continue
    st = PyBartSettings.from_data(X, c.get("bart_Y", Y), m=c["m"], num_particles=c["P"], seed=c["seed"], batch=c["batch"],
                                  alpha=c["alpha"], beta=c["beta"], family=c["family"], n_outputs=c["K"],
                                  response=c["response"])
    s = PySampler(st, X, Y, c["rules"], c["prior"], backend=be)
    if c.get("offset") is not None:
        s.set_offset(c["offset"])
    s.set_likelihood(c.get("lik_params", [0.5] if c["family"] == "normal" else []))
    moved = False
    first_mu = None
    for it in range(40):
        mu, _ = s.step(True)
        if first_mu is None:
            first_mu = mu.copy()
        moved = moved or not np.array_equal(mu, first_mu)
    sd = s.state()["leaf_sd"]
    flags = []
    if not np.all(np.isfinite(mu)):
        flags.append("non-finite sum_trees")
    if not np.all(np.isfinite(sd)) or np.any(sd <= 0):
        flags.append(f"leaf_sd {sd}")
    splittable = any(len(np.unique(X[~np.isnan(X[:, j]), j])) >= 2 for j in range(X.shape[1]))
    if not moved and splittable:
        flags.append("sum_trees never moved in 40 steps although a column can be split")
    if s.counters.saturations:
        flags.append(f"{s.counters.saturations} saturations")
    if flags:
        bad[seed] = (c["family"], X.shape, c["m"], c["P"], c["K"], c["response"], flags)
        print("ANOMALY", seed, *bad[seed], flush=True)
print(f"liveness: seeds {first}..{first + count - 1}: {len(bad)} anomalies")
