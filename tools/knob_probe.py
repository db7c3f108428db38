#!/usr/bin/env python3
"""Resident single-chain rate of a workload under the current environment knobs (median of REPS blocks of 20
asteps after the standard burn-in).  usage (GPU box): [PGB_...=v] python tools/knob_probe.py [cfg2|cfg4|cfg5] [REPS] [K=6] [T=1]
(K: outputs of cfg5; T=1: measure tuning asteps instead of draws)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pymc_bart_amd import workloads  # noqa: E402
from pymc_bart_amd.sampler import PyBartSettings, PySampler, default_backend  # noqa: E402

wn = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
kw = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[3:] if "=" in a}
tune = bool(kw.pop("T", 0))
w = getattr(workloads, wn)(**kw)
X, Y = w["X"], w["Y"]
st = PyBartSettings.from_data(X, Y, m=w["m"], num_particles=w["num_particles"], seed=3415, family=w["family"],
                              n_outputs=w.get("K", 1))
s = PySampler(st, X, Y, np.zeros(X.shape[1], np.int32), np.ones(X.shape[1]), backend=default_backend(0))
s.set_likelihood([1.0] if w["family"] == "normal" else [])
s.step_async(True, 100)
s.sync()
s.step_async(False, 10)
s.sync()
steps = 20 if wn == "cfg2" else 8
rates = []
for _ in range(reps):
    c0 = s.sync()["particle_steps"]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.step_async(tune, steps)
    c1 = s.sync()["particle_steps"]
    rates.append((c1 - c0) / (time.perf_counter() - t0))
knobs = {k: v for k, v in os.environ.items() if k.startswith("PGB_")}
print(f"{wn} {kw} tune={int(tune)} {knobs}: {np.median(rates) / 1e6:.4f} M (min {min(rates) / 1e6:.4f}, max {max(rates) / 1e6:.4f})")
