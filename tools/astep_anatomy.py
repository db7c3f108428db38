#!/usr/bin/env python3
"""Where the time of one PGBART.astep goes beyond the device-resident rate (cfg2, steady state).
Run on the GPU box: python tools/astep_anatomy.py [--steps 200]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    import torch

    from pymc_bart_amd import workloads
    from pymc_bart_amd.pgbart import PGBART, BARTOp, NormalLikelihood
    from pymc_bart_amd.sampler import default_backend
    from pymc_bart_amd.utils import _encode_vi

    be = default_backend(0)
    w = workloads.cfg2()
    st = PGBART([BARTOp(w["X"], w["Y"], m=w["m"])], num_particles=40, likelihood=NormalLikelihood(1.0),
                observed=w["Y"], random_seed=3415, backend=be)
    s = st.sampler
    s.step_async(True, 100)
    s.sync()
    st.tune = False
    for _ in range(5):
        st.astep(None)
    N = args.steps
    lib, mem = be.lib, be.mem
    K, n = 1, w["X"].shape[0]

    def timed(fn, reps=N):
        torch.cuda.synchronize()
        c0 = s.sync()["particle_steps"]
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        c1 = s.sync()["particle_steps"]
        return dt / reps * 1e6, (c1 - c0) / max(reps, 1)

    # ---- interleaved blocks (the chain's state drifts: compare modes on alternating blocks of 20 steps)
    vi0 = np.zeros(50, np.int32)
    buf0 = mem.host_result(K * n)
    modes = {
        "resident (async 20)": lambda: (s.step_async(False, 20), s.sync()),
        "pgb_step x20": lambda: [s.step(False, fetch=False) for _ in range(20)],
        "pgb_step_host(no st) x20": lambda: [lib.lib.pgb_step_host(s._h, 0, None, vi0.ctypes.data, C.byref(s.counters)) for _ in range(20)],
        "pgb_step_host(st) x20": lambda: [lib.lib.pgb_step_host(s._h, 0, buf0.ctypes.data, vi0.ctypes.data, C.byref(s.counters)) for _ in range(20)],
        "astep x20": lambda: [st.astep(None) for _ in range(20)],
        "astep x20 (results dropped)": lambda: [st.astep(None) and None for _ in range(20)],
        "sampler.step(fetch) x20 dropped": lambda: [s.step(False) and None for _ in range(20)],
    }
    acc = {k: [0.0, 0, 0] for k in modes}
    for rep in range(max(4, N // 20)):
        for k, fn in modes.items():
            torch.cuda.synchronize()
            c0 = s.sync()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            c1 = s.sync()
            acc[k][0] += dt
            acc[k][1] += c1["particle_steps"] - c0["particle_steps"]
            acc[k][2] += c1["slots"] - c0["slots"]
    base = None
    for k, (dt, ps, sl) in acc.items():
        nsp = dt / ps * 1e9
        base = base or nsp
        steps = max(4, N // 20) * 20
        print(f"[interleaved] {k:28s} {dt / steps * 1e6:8.1f} us/step {nsp:7.2f} ns/p-step  ratio {base / nsp:5.3f}  "
              f"{sl / steps:6.1f} working slots/step  -> +{(nsp - base) * ps / steps / 1e3:6.1f} us/step")
    st._batches.clear()

    out = {}
    # resident: N asteps in one async call
    torch.cuda.synchronize()
    c0 = s.sync()["particle_steps"]
    t0 = time.perf_counter()
    s.step_async(False, N)
    c1 = s.sync()["particle_steps"]
    out["resident us/step"] = ((time.perf_counter() - t0) / N * 1e6, (c1 - c0) / N)
    out["pgb_step (sync, device out)"] = timed(lambda: s.step(False, fetch=False))
    vi = np.zeros(50, np.int32)
    ctr = s.counters

    def host_no_dma():
        lib.lib.pgb_step_host(s._h, 0, None, vi.ctypes.data, C.byref(ctr))

    out["pgb_step_host, no sum_trees"] = timed(host_no_dma)
    buf = mem.host_result(K * n)

    def host_dma_same_buf():
        lib.lib.pgb_step_host(s._h, 0, buf.ctypes.data, vi.ctypes.data, C.byref(ctr))

    out["pgb_step_host, reused pinned buffer"] = timed(host_dma_same_buf)
    out["sampler.step(fetch=True)"] = timed(lambda: s.step(False))
    out["PGBART.astep"] = timed(lambda: st.astep(None))
    t0 = time.perf_counter()
    for _ in range(2000):
        mem.host_result(K * n)
    out["host_result alloc"] = ((time.perf_counter() - t0) / 2000 * 1e6, 0)
    t0 = time.perf_counter()
    for _ in range(2000):
        s.export_trees(0)
    out["export_trees(0)"] = ((time.perf_counter() - t0) / 2000 * 1e6, 0)
    t0 = time.perf_counter()
    for _ in range(2000):
        _encode_vi(vi)
    out["_encode_vi"] = ((time.perf_counter() - t0) / 2000 * 1e6, 0)
    t0 = time.perf_counter()
    for _ in range(2000):
        s.set_likelihood([1.0])
    out["set_likelihood"] = ((time.perf_counter() - t0) / 2000 * 1e6, 0)
    base = out["resident us/step"][0] / max(out["resident us/step"][1], 1)
    for k, (us, ps) in out.items():
        if ps:
            print(f"{k:40s} {us:9.1f} us/step  {ps:8.1f} p-steps/step  {us / ps * 1000:7.2f} ns/p-step "
                  f"(+{(us / ps - base) * ps:6.1f} us/step over resident)")
        else:
            print(f"{k:40s} {us:9.1f} us/call")


if __name__ == "__main__":
    main()
