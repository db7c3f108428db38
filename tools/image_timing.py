#!/usr/bin/env python3
"""Cost of the chain image (pgb_checkpoint_save / _load) at BASELINE's sizes, on the GPU box:
python tools/image_timing.py  ->  one JSON line {cfg: {bytes, save_ms, load_ms}}."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pymc_bart_amd import workloads  # noqa: E402
from pymc_bart_amd.sampler import PyBartSettings, PySampler, default_backend  # noqa: E402

be = default_backend(0)
out = {}
for name, kw in (("cfg2", {}), ("cfg4", dict(family="bernoulli_probit")), ("cfg5", dict(family="categorical", n_outputs=4))):
    w = getattr(workloads, name)(seed=3415)
    X, Y = w["X"], w["Y"]
    p = X.shape[1]
    st = PyBartSettings.from_data(X, Y, m=w["m"], num_particles=40, seed=1, **kw)
    s = PySampler(st, X, Y, np.zeros(p, np.int32), np.ones(p), backend=be)
    s.set_likelihood([1.0] if name == "cfg2" else [])
    s.step_async(True, 30)
    s.sync()
    ts, tl = [], []
    for _ in range(3):
        t0 = time.perf_counter()
        blob = s.checkpoint()
        t1 = time.perf_counter()
        s.restore(blob)
        t2 = time.perf_counter()
        ts.append(t1 - t0)
        tl.append(t2 - t1)
    out[name] = {"bytes": len(blob), "save_ms": round(min(ts) * 1e3, 2), "load_ms": round(min(tl) * 1e3, 2)}
    del s
print(json.dumps(out))
