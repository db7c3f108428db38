#!/usr/bin/env python3
"""Aggregate rate of C chains running concurrently on one GPU (resident path), for C in 1, 2, 4.
usage (GPU box): [PGBART_HIP_LIB=...] python tools/multichain_probe.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pymc_bart_amd import workloads  # noqa: E402
from pymc_bart_amd.sampler import PyBartSettings, PySampler, default_backend  # noqa: E402

be = default_backend(0)
w = workloads.cfg2()
X, Y = w["X"], w["Y"]
ss = []
import contextlib  # noqa: E402

# PROBE_DUMMY=N: N further streams that do a little work once and then stay idle (do they take hardware
# queues away from the chains?)
dummies = []
for i in range(int(os.environ.get("PROBE_DUMMY", "0"))):
    if os.environ.get("PROBE_DUMMY_NULL") and i == 0:
        torch.zeros(16, device="cuda").add_(1)   # the legacy default stream
        continue
    d = torch.cuda.Stream()
    with torch.cuda.stream(d):
        torch.zeros(16, device="cuda").add_(1)
    dummies.append(d)
torch.cuda.synchronize()

for c in range(4):
    # PROBE_DEFAULT0=1: chain 0 on the default (null) stream, like bench.py's first chain
    ctx = contextlib.nullcontext() if ((c == 0 and os.environ.get("PROBE_DEFAULT0")) or os.environ.get("PROBE_OWN")) else torch.cuda.stream(torch.cuda.Stream())
    with ctx:
        st = PyBartSettings.from_data(X, Y, m=200, num_particles=40, seed=3415 + 1000 * c)
        s = PySampler(st, X, Y, np.zeros(50, np.int32), np.ones(50), backend=be)
        s.set_likelihood([1.0])
        ss.append(s)
torch.cuda.synchronize()
for s in ss:
    s.step_async(True, 40)
for s in ss:
    s.sync()
for C in (4,) if os.environ.get("PROBE_ONLY4") else (1, 2, 4):
    rates = []
    for rep in range(5):
        sel = ss[:C]
        c0 = sum(s.sync()["particle_steps"] for s in sel)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in sel:
            s.step_async(False, 20)
        c1 = sum(s.sync()["particle_steps"] for s in sel)
        torch.cuda.synchronize()
        rates.append((c1 - c0) / (time.perf_counter() - t0))
    print(f"{C} chains: {np.median(rates) / 1e6:.3f} M particle-steps/s (min {min(rates) / 1e6:.3f}, max {max(rates) / 1e6:.3f})")
