#!/usr/bin/env python3
"""bench.py -- particle-steps/sec of the PGBART hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one ``PGBART.astep`` (a batch of 10% of the m trees re-sampled by particle Gibbs) on the
configuration BASELINE.json's metric is quoted on: n=100k, p=50, m=200 trees, 40 particles, Gaussian
likelihood (cfg2), synthetic data resident in HBM.  ``value`` times the step method itself -- the call
``pm.sample`` makes: sum_trees back on the host, the step's trees exported, stats encoded, history
published (round-2 VERDICT: the honest answer to "PGBART.astep on one MI355X").  The device-resident rate
(``pgb_step_async``: no host outputs) is reported beside it as ``resident_path``.

N > 1: ``python bench.py --gpus N`` launches its own N ranks (``torch.distributed.run``, one process
per GPU, RCCL) BEFORE anything touches a GPU; launched under ``torch.distributed.run`` by somebody
else (RANK set) it is one of those ranks.  N independent chains run one per GPU (weak scaling, no
data-path collective); the draws are gathered over RCCL after the timed region.  Rank 0 prints ONE
JSON line.

Protocol (steady state): burn in ``--burnin`` asteps with tune=1 (default 100 = 10 sweeps
over the m trees), switch to tune=0, W untimed warm-up asteps, then blocks of EXACTLY K asteps, each
bracketed by barrier + synchronize, until at least ``--min-seconds`` of GPU time have been timed (or
``--repeats`` blocks when given); ``value`` is the median block (min / max alongside).  Further legs at
N=1: ``resident_path``, ``tune1``, the per-kernel profile behind ``roofline`` / ``roofline_kernels``, 4
chains on the one GPU -- and the other single-GPU configurations of BASELINE.json, cfg4 (Bernoulli-probit,
n=1M) and cfg5 (K=4 softmax, n=250k), each with its own value, dominant-kernel roofline and row-pass HBM
roofline from measured bytes, under ``workloads``.  ALL GPU legs run first, back to back (the GPU is held
for ~30 s contiguously); the CPU baselines (cfg2: 1 core and 8 chains on 8 cores; cfg4 / cfg5: a bounded
sample) follow, each with a GPU rate at the SAME chain age beside it (``matched_age``).  The line ENDS with
a compact ``summary`` object that repeats every headline figure (a log tail keeps it); the prose that
explains the fields is profiles/BENCH_NOTES.md.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# fp64 VALU issue: 256 CUs x 4 SIMDs x 2.4 GHz, a wave64 fp64 instruction occupies its SIMD for 4
# cycles (78.6 TFLOP/s fp64 vector = 16 lanes x 2 flop per SIMD and clock)
VALU_F64_PEAK_GINST = 256 * 4 * 2.4 / 4.0  # G wave-instructions / s
ROUND = "r06"
NOTES = "profiles/BENCH_NOTES.md"

METRIC = {
    "cfg2": "particle-steps/sec (n=100k, p=50, m=200, 40 particles)",
    "cfg4": "particle-steps/sec (Bernoulli-probit, n=1M, p=100, m=200, 40 particles)",
    "cfg5": "particle-steps/sec (K=4 softmax, n=250k, p=200, m=100, 40 particles)",
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed blocks of --steps asteps; value = median.  0 (default): as many as --min-seconds needs")
    ap.add_argument("--min-seconds", type=float, default=5.0,
                    help="GPU time the headline's timed blocks cover at least (also of each workload leg)")
    ap.add_argument("--burnin", type=int, default=100,
                    help="tune=1 asteps before anything is timed (100 = 10 sweeps at the default batch)")
    ap.add_argument("--n", type=int, default=100_000)
    ap.add_argument("--p", type=int, default=50)
    ap.add_argument("--m", type=int, default=200)
    ap.add_argument("--particles", type=int, default=40)
    ap.add_argument("--tune", type=int, default=0, help="headline block with tune=1 instead of tune=0")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg4", "cfg5"],
                    help="cfg2 (default) is the configuration the metric is quoted on")
    ap.add_argument("--chains-per-gpu", type=int, default=1,
                    help="independent chains run concurrently on each GPU (own stream each); the headline "
                         "is quoted at 1, as north_star shards one chain per GPU")
    ap.add_argument("--outputs", type=int, default=4, help="cfg5 only: number of outputs K (4 = BASELINE's cfg5)")
    ap.add_argument("--response", default="constant", choices=["constant", "linear", "mix"])
    ap.add_argument("--no-multichain", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only (skip resident / tune1 / workload legs)")
    ap.add_argument("--no-workloads", action="store_true", help="skip the cfg4 / cfg5 legs of the default run")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of timed CPU work per baseline leg")
    ap.add_argument("--gather-draws", type=int, default=100,
                    help="N-rank runs: posterior draws per rank that the end-of-run gather moves to rank 0 "
                         "(100 draws of n doubles = 80 MB per rank at cfg2; capped at 256 MiB per rank)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing check without a GPU: the launcher, the process group, the aggregation and "
                         "the JSON line run over a counter stub; the line says so and carries no throughput")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# N > 1 without a launcher: become the launcher.  Nothing here initialises a GPU (device_count does
# not, on this image), and the ranks are CHILD processes -- never an exec of this one.
def launch_ranks(args) -> int:
    if not args.dry_run:
        import torch

        ndev = torch.cuda.device_count()
        if ndev < args.gpus:
            sys.stderr.write(f"bench.py --gpus {args.gpus}: only {ndev} GPU(s) visible on this node; "
                             f"refusing to run (a {args.gpus}-GPU line is never printed from fewer GPUs)\n")
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------
# CPU baseline: the oracle (oracle/, a C restatement), 1 chain on 1 core and C chains on C cores
def _cpu_run(w, seed, budget_s, burn, response, so_path, trees_per_step=None, max_steps=256):
    """Time the oracle on workload ``w``.  ``trees_per_step``: re-sample that many trees per step instead of
    10 % of m (the bounded sample of the large configurations: one tree update there costs seconds)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle import NumpyMemory
    from pymc_bart_amd import _abi
    from pymc_bart_amd.sampler import Backend, PyBartSettings, PySampler

    X, Y = w["X"], w["Y"]
    batch = (0.1, 0.1) if trees_per_step is None else (int(trees_per_step), int(trees_per_step))
    st = PyBartSettings.from_data(X, Y, m=w["m"], num_particles=w["num_particles"], seed=seed,
                                  family=w["family"], n_outputs=w.get("K", 1), response=response, batch=batch)
    be = Backend(lib=_abi.PGBLibrary(so_path), mem=NumpyMemory())
    t_c = time.perf_counter()
    s = PySampler(st, X, Y, np.zeros(X.shape[1], np.int32), np.ones(X.shape[1]), backend=be)
    s.set_likelihood([1.0] if w["family"] == "normal" else [])
    create_s = time.perf_counter() - t_c
    for _ in range(burn):
        s.step(True, fetch=False)
    s.step(False, fetch=False)
    c0 = s.sync()
    t0 = time.perf_counter()
    steps = 0
    while True:
        s.step(False, fetch=False)
        steps += 1
        if time.perf_counter() - t0 > budget_s or steps >= max_steps:
            break
    dt = time.perf_counter() - t0
    c1 = s.sync()
    return {k: c1[k] - c0[k] for k in c1} | {"dt": dt, "steps": steps, "create_s": create_s}


def _cpu_worker(args_tuple):
    (wname, wkw, seed, budget_s, burn, response, so_path, *rest) = args_tuple
    trees_per_step, max_steps = (rest + [None, 256])[:2] if rest else (None, 256)
    sys.path.insert(0, ROOT)
    from pymc_bart_amd import workloads

    return _cpu_run(getattr(workloads, wname)(**wkw), seed, budget_s, burn, response, so_path,
                    trees_per_step=trees_per_step, max_steps=max_steps)


def _cpu_host():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), model)
    except OSError:
        pass
    return {"nproc": os.cpu_count(), "cpu_model": model}


def cpu_baseline(wname, wkw, seed, budget_s, response="constant"):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import multiprocessing as mp

    from _oracle import build_oracle_native

    so_path, flags = build_oracle_native()
    cores = min(8, os.cpu_count() or 1)
    burn = CPU_BURN_CFG2 if wname == "cfg2" else 0  # (the GPU rate at this chain age: cpu_baseline.matched_age)
    ctx = mp.get_context("spawn")
    with ctx.Pool(1) as pool:
        one = pool.map(_cpu_worker, [(wname, wkw, seed, budget_s, burn, response, so_path)])[0]
    many = None
    if cores > 1:
        with ctx.Pool(cores) as pool:
            many = pool.map(_cpu_worker, [(wname, wkw, seed + c, budget_s, burn, response, so_path)
                                          for c in range(cores)])
    out = {
        "value": one["particle_steps"] / one["dt"], "unit": "particle-steps/s", "cores": 1, "kind": "port",
        "sample": f"{one['steps']} tune=0 asteps ({one['tree_updates']} tree updates, {one['dt']:.1f} s) of the same "
                  f"{wname} data after {burn} tune=1 + 1 tune=0 warm-up asteps; oracle/ (C restatement), 1 chain on 1 core",
        "burnin_asteps_tune1": burn,
        "compiler": flags,
        "tree_updates_per_s": one["tree_updates"] / one["dt"],
        "rows_touched_per_particle_step": one["rows_touched"] / max(one["particle_steps"], 1),
        "rows_touched_per_s": one["rows_touched"] / one["dt"],
        "host": _cpu_host(),
    }
    if many:
        out["all_cores"] = {
            "value": sum(r["particle_steps"] / r["dt"] for r in many), "unit": "particle-steps/s",
            "cores": cores, "chains": cores,
            "sample": f"{cores} independent chains as {cores} processes, {budget_s:.0f} s each, same protocol",
            "per_chain_min": min(r["particle_steps"] / r["dt"] for r in many),
            "per_chain_max": max(r["particle_steps"] / r["dt"] for r in many),
        }
    return out


def cpu_baseline_short(w, wname, seed, budget_s, all_cores=True):
    """The bounded CPU sample of a large configuration (cfg4 / cfg5): the oracle in THIS process on the data
    already generated, one tree update per step (a 10 %-of-m astep costs the oracle tens of seconds there),
    one core.  Runs after every GPU leg of the workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle import build_oracle_native

    so_path, flags = build_oracle_native()
    r = _cpu_run(w, seed, budget_s, 0, "constant", so_path, trees_per_step=1, max_steps=64)
    many = None
    cores = min(8, os.cpu_count() or 1)
    if all_cores and cores > 1:
        # C chains as C processes, each on its own copy of the data (regenerated in the child: 0.8 / 0.4 GB), the
        # same one-tree-per-step sample -- the "8 chains on 8 cores" row of BASELINE.md's plan for this workload
        import multiprocessing as mp

        with mp.get_context("spawn").Pool(cores) as pool:
            many = pool.map(_cpu_worker, [(wname, dict(seed=3415), seed + c, budget_s, 0, "constant", so_path, 1, 64)
                                          for c in range(cores)])
    extra = {}
    if many:
        extra["all_cores"] = {"value": sum(q["particle_steps"] / q["dt"] for q in many), "unit": "particle-steps/s",
                              "cores": cores, "chains": cores,
                              "sample": f"{cores} independent chains as {cores} processes, same one-tree-per-step sample"}
    return extra | {
        "value": r["particle_steps"] / r["dt"], "unit": "particle-steps/s", "cores": 1, "kind": "port",
        "sample": f"{r['tree_updates']} tune=0 tree updates ({r['dt']:.1f} s, one tree per step) of the same {wname} "
                  f"data from the start of a chain after 1 warm-up tree update; oracle/ (C restatement), 1 chain on 1 core",
        "tree_updates": r["tree_updates"],
        "compiler": flags,
        "rows_touched_per_particle_step": r["rows_touched"] / max(r["particle_steps"], 1),
        "rows_touched_per_s": r["rows_touched"] / r["dt"],
        "host": _cpu_host(),
    }


def cpu_cfg1(budget_s=3.0):
    """BASELINE.json configs[0] -- Friedman n=500, p=5, m=50, 10 particles, one chain on the CPU restatement (one core).
    The GPU never touches it: BASELINE.md's table has a row for it, and it is the smallest case both backends are
    held to (tests: `cfg1_friedman`)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle import build_oracle_native
    from pymc_bart_amd import workloads

    so_path, flags = build_oracle_native()
    w = workloads.cfg1(3415)
    r = _cpu_run(w, 3415, budget_s, 10, "constant", so_path, max_steps=100_000)
    return {"metric": "particle-steps/sec (cfg1: Friedman n=500, p=5, m=50, 10 particles)", "backend": "oracle-cpu",
            "value": r["particle_steps"] / r["dt"], "unit": "particle-steps/s", "cores": 1, "kind": "port",
            "tree_updates_per_s": r["tree_updates"] / r["dt"], "compiler": flags,
            "sample": f"{r['steps']} tune=0 asteps ({r['dt']:.1f} s) after 10 tune=1 asteps"}


def kernel_source_sha256(root=None):
    """Identity of the device code of this tree: sha256 over the sources the HIP library is built from
    (pymc_bart_amd/csrc/*.hip, *.h, include/*.h) and the compiler flags.  (Not a hash of the .so: two hipcc builds of
    the same sources differ in their bytes.)"""
    import glob
    import hashlib

    root = ROOT if root is None else root
    files = sorted(glob.glob(os.path.join(root, "pymc_bart_amd", "csrc", "*.h"))
                   + glob.glob(os.path.join(root, "pymc_bart_amd", "csrc", "*.hip"))
                   + glob.glob(os.path.join(root, "include", "*.h")))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, root).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    try:
        sys.path.insert(0, root)
        import __graft_entry__ as g

        h.update(" ".join(x for x in g.HIPCC_FLAGS if not x.startswith("-I")).encode())
    except Exception:  # noqa: BLE001 - the flags are a refinement: the sources alone still identify a kernel change
        pass
    return h.hexdigest()


# ---------------------------------------------------------------------------------------------
class DryRunSampler:
    """--dry-run only: stands in for a chain so that the launcher / process group / aggregation / JSON
    plumbing can be exercised on a box without a GPU.  It produces no sampling work."""

    def __init__(self, rank):
        self.c = {"particle_steps": 0, "tree_updates": 0, "rows_touched": 0, "rounds": 0, "saturations": 0,
                  "slots": 0, "partitions": 0}
        self.rank = rank

    def step_async(self, tune, k):
        time.sleep(0.0005 * k)
        self.c["particle_steps"] += 100 * k
        self.c["tree_updates"] += k
        self.c["rows_touched"] += 1000 * k

    def sync(self):
        return dict(self.c)


class DryRunStep:
    """--dry-run only: the astep-shaped face of :class:`DryRunSampler`."""

    def __init__(self, sampler):
        self.sampler = sampler
        self.tune = True

    def astep(self, _q=None):
        self.sampler.step_async(self.tune, 1)

    @property
    def counters(self):
        return self.sampler.sync()


def median_block(blocks):
    """blocks: list of (seconds, units dict).  Returns (median block by rate, min rate, max rate)."""
    rates = [b[1]["particle_steps"] / b[0] for b in blocks]
    order = np.argsort(rates)
    mid = blocks[int(order[len(order) // 2])]
    return mid, float(min(rates)), float(max(rates))


def run_all(ss, tn, k):
    """k asteps of every chain on the resident path: started asynchronously (each handle's worker thread
    feeds its own stream), then waited for.  Returns the summed counters."""
    if k > 0:
        for q in ss:
            q.step_async(tn, k)
    tot = {}
    for q in ss:
        for key, v in q.sync().items():
            tot[key] = tot.get(key, 0) + v
    return tot


def resident_blocks(ss, tn, steps, repeats, barrier):
    out = []
    a = run_all(ss, tn, 0)
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        b = run_all(ss, tn, steps)
        barrier()
        el = time.perf_counter() - t0
        out.append((el, {key: b[key] - a[key] for key in b}))
        a = b
    return out


def astep_blocks(step, steps, repeats, barrier, min_seconds=0.0, agree=None, max_blocks=400):
    """Blocks of exactly ``steps`` calls of ``step.astep`` (the call PyMC makes), each bracketed by
    barrier + synchronize.  ``repeats`` > 0: that many blocks.  Otherwise blocks are added until
    ``min_seconds`` are covered; ``agree`` (max over ranks of a float) keeps every rank on the same count."""
    out = []
    a = dict(step.counters)

    def one():
        nonlocal a
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step.astep(None)
        barrier()
        el = time.perf_counter() - t0
        b = dict(step.counters)
        out.append((el, {key: b[key] - a[key] for key in b}))
        a = b
        return el

    if repeats > 0:
        for _ in range(repeats):
            one()
        return out
    first = one()
    first = agree(first) if agree is not None else first
    more = int(np.ceil(min_seconds / max(first, 1e-6))) - 1
    for _ in range(max(4, min(more, max_blocks - 1))):
        one()
    return out


def kernel_profile(s, tune, steps):
    """A further block of ``steps`` asteps with HIP events attached to every dispatch (hipExtLaunchKernelGGL:
    the start / stop stamps of the kernel's own packet, on the stream the kernels are launched on)."""
    s.profile(True)
    cp0 = s.sync()
    s.step_async(tune, steps)
    cp1 = s.sync()
    ms_rows, launches = s.profile(False)
    kern = s.profile_kernels()
    clk_ms, clk_launches = s.profile_clock()
    d = {k: cp1[k] - cp0[k] for k in cp1}
    return {"kern": kern, "ms_rows": ms_rows, "launches": launches, "clk_ms": clk_ms, "clk_launches": clk_launches,
            "tree_updates": d["tree_updates"], "rows_touched": d["rows_touched"], "partitions": d["partitions"],
            "slots": d["slots"]}


def load_pmc(wname, current=None):
    """Per-launch counter averages of the hot kernels from the separate ``rocprofv3 --pmc`` passes
    (``tools/pmc_collect.sh``; the newest round's file that exists).  Returns (counters, file, stale): the counters
    are read from a COMMITTED file, not measured in this run, so each file names the kernel sources it was taken on
    (``kernel_source_sha256``); ``stale`` is True when those are not this tree's -- a kernel changed and
    nobody re-ran the counter passes (round-5 VERDICT, weak #6).  ``current``: this tree's hash (tests pass one)."""
    for rnd in (ROUND, "r05", "r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_{wname}.json")
        if os.path.exists(path):
            with open(path) as fh:
                d = json.load(fh)
            have = kernel_source_sha256() if current is None else current
            stale = d.get("kernel_source_sha256") != have
            return d, f"profiles/{rnd}_pmc_{wname}.json", bool(stale)
    return {}, None, False


def rooflines(wname, w, X_shape, prof, response="constant", default_cfg=True):
    """``roofline`` (the dominant kernel of the workload), ``roofline_rows`` (the row pass, HBM) and
    ``roofline_kernels`` (every kernel of the slot) from a :func:`kernel_profile` block."""
    from pymc_bart_amd import workloads

    n, p = X_shape
    K_out = w.get("K", 1)
    kern = prof["kern"]
    tot_ms = sum(k["ms"] for k in kern.values()) or 1.0
    out = {"roofline_kernels": {
        name: {"pct": 100.0 * k["ms"] / tot_ms, "avg_us": k["ms"] * 1e3 / k["launches"],
               "launches": k["launches"], "workgroups": k["workgroups"]}
        for name, k in kern.items()}}
    if "k_ctrl" in out["roofline_kernels"]:
        # the control kernel has no bandwidth or FLOP roofline: one workgroup per particle walks a chain of
        # dependent loads and scalar decisions (profiles/r03_experiments.md section 5: ~1.6 us until its first data,
        # ~3.8 us of dependent work); its floor is latency, and it is the other half of a cfg2 slot
        out["roofline_kernels"]["k_ctrl"]["bound"] = "latency"
    pmc, pmc_src, pmc_stale = load_pmc(wname) if default_cfg else ({}, None, False)
    ms_rows, launches = prof["ms_rows"], prof["launches"]
    tu, rt, parts = prof["tree_updates"], prof["rows_touched"], prof["partitions"]
    rows = None
    if ms_rows > 0:
        alg = workloads.bytes_per_tree_update(n, rt / max(tu, 1), K=K_out) * tu
        ach = alg / (ms_rows * 1e-3) / 1e9
        nact = parts / max(launches, 1)
        nchunks = (n + 1023) // 1024
        G = max(1, -(-int(round(nact * nchunks)) // 640))
        ngroups = max(1.0, nact / G)
        # what THIS layout moves per launch: every active particle streams its n labels in and out
        # (1 B each) and the split column (8 B per row; 2 B of order keys when the matrix exceeds
        # the Infinity Cache); each particle group re-reads {sum_trees, r} (16 B per row); a tree-boundary
        # pass adds the INIT/FINAL streams (~26 B read per row and group, 25 B written per row)
        shadow = response == "constant" and p * (nchunks * 1024) * 8 >= (192 << 20)
        xbytes = 2.0 if shadow else 8.0
        impl = parts * (2.0 + xbytes) * n + launches * ngroups * 16.0 * n + tu * (ngroups * 26.0 + 25.0) * n
        avg_us = ms_rows * 1e3 / max(launches, 1)
        traffic = (pmc.get("k_rows") or pmc.get("k_rows_mk") or {}).get("hbm_bytes_per_launch_corrected")
        rows = {
            "bound": "hbm", "kernel": "k_rows" if K_out == 1 else "k_rows_mk", "achieved": ach, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": pmc_src,
            "traffic_stale": pmc_stale if traffic else None,
            "launches": launches, "avg_launch_us": avg_us,
            "avg_kernel_us_device_clock": (prof["clk_ms"] * 1e3 / prof["clk_launches"]) if prof["clk_launches"] else None,
            "frac_device_clock": (alg / (prof["clk_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if prof["clk_ms"] > 0 else None,
            "algorithmic_bytes_per_launch": alg / max(launches, 1),
            "implementation_bytes_per_launch_model": impl / max(launches, 1),
            "implementation_GBps_model": impl / (ms_rows * 1e-3) / 1e9,
            "note": NOTES + "#roofline-of-the-row-pass",
        }
        if traffic:
            rows["measured_hbm_GBps"] = traffic / (avg_us * 1e-6) / 1e9
            rows["measured_hbm_frac"] = rows["measured_hbm_GBps"] / HBM_PEAK_GBS
    if wname in ("cfg4", "cfg5") and "k_loglik" in kern:
        kl = kern["k_loglik"]
        insts = pmc.get("k_loglik", {}).get("valu_wave_insts_per_launch")
        ginst = (insts * kl["launches"] / (kl["ms"] * 1e-3) / 1e9) if insts else None
        # the row pass is NOT the dominant kernel of these workloads and the index-list byte model exceeds what
        # this layout moves: the workload's `roofline` is the dominant kernel's; the row pass is graded on its
        # MEASURED HBM bytes (round-2 VERDICT)
        out["roofline"] = {
            "bound": "valu-f64-issue", "kernel": "k_loglik", "pct_of_gpu_time": 100.0 * kl["ms"] / tot_ms,
            "achieved": ginst, "peak": VALU_F64_PEAK_GINST, "unit": "G wave-instructions/s",
            "frac": (ginst / VALU_F64_PEAK_GINST) if ginst else None,
            "avg_launch_us": kl["ms"] * 1e3 / kl["launches"], "launches": kl["launches"],
            "valu_wave_insts_per_launch": insts, "insts_source": pmc_src, "traffic_stale": pmc_stale if insts else None,
            "note": NOTES + "#roofline-of-the-likelihood-pass",
        }
        if rows is not None:
            if rows.get("measured_hbm_GBps"):
                rows["model_achieved"], rows["model_frac"] = rows["achieved"], rows["frac"]
                rows["achieved"], rows["frac"] = rows["measured_hbm_GBps"], rows["measured_hbm_frac"]
                rows["note"] = NOTES + "#row-pass-graded-on-measured-bytes"
            out["roofline_rows"] = rows
    elif rows is not None:
        out["roofline"] = rows
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))
    t_start = time.perf_counter()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    dry = args.dry_run
    if not dry and torch.cuda.device_count() < max(1, local_rank + 1):
        raise SystemExit(f"bench.py: rank {rank} has no GPU {local_rank} "
                         f"({torch.cuda.device_count()} visible); there is no CPU fallback")
    dist = None
    ranks_reported = 1
    if "RANK" in os.environ:  # launched by torch.distributed.run (any world size)
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if dry or args.backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
    elif not dry:
        torch.cuda.set_device(0)
    cdev = "cpu" if (dry or args.backend == "gloo") else "cuda"

    def barrier():
        if dist is not None:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    def allreduce(vals, op):
        if dist is None:
            return [float(v) for v in vals]
        t = torch.tensor(list(vals), dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=getattr(dist.ReduceOp, op))
        return [float(x) for x in t.tolist()]

    if dist is not None:  # the number of ranks the collective library itself sees
        ranks_reported = int(round(allreduce([1.0], "SUM")[0]))

    from pymc_bart_amd import workloads

    seed = 3415 + rank  # independent chains: SURVEY.md 8e
    if args.workload == "cfg4":
        wname, wkw = "cfg4", dict(seed=3415, n=args.n if args.n != 100_000 else 1_000_000,
                                  p=args.p if args.p != 50 else 100, m=args.m, num_particles=args.particles)
    elif args.workload == "cfg5":
        wname, wkw = "cfg5", dict(seed=3415, num_particles=args.particles, K=args.outputs)
    else:
        wname, wkw = "cfg2", dict(seed=3415, n=args.n, p=args.p, m=args.m, num_particles=args.particles)
    tune = bool(args.tune)
    default_cfg = ((args.workload != "cfg2") or (args.n, args.p, args.m, args.particles) == (100_000, 50, 200, 40)) \
        and (args.workload != "cfg5" or args.outputs == 4)

    def make_chain(wn, kw, sd, be, batch=(0.1, 0.1), wl=None):
        """One chain of workload ``wn`` as the step method itself: its sampler is the resident path."""
        from pymc_bart_amd.pgbart import (PGBART, BARTOp, BernoulliLikelihood, CategoricalLikelihood,
                                          NormalLikelihood)
        import warnings

        wl = getattr(workloads, wn)(**kw) if wl is None else wl
        lik = {"normal": lambda: NormalLikelihood(1.0),  # sigma fixed at 1
               "bernoulli_probit": lambda: BernoulliLikelihood("probit"),
               "categorical": lambda: CategoricalLikelihood(wl.get("K", 1))}[wl["family"]]()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # response=linear is flagged experimental, as upstream
            op = BARTOp(wl["X"], wl["Y"], m=wl["m"], response=args.response)
        st = PGBART([op], num_particles=wl["num_particles"], likelihood=lik, observed=wl["Y"], random_seed=sd,
                    backend=be, batch=batch)
        return wl, st

    if dry:
        w = {"name": "dry-run (no sampling work)", "family": "normal", "m": args.m}
        n, X = args.n, None
        samplers = [DryRunSampler(rank)]
        step = DryRunStep(samplers[0])
        batch_trees = max(1, args.m // 10)
        be = None
    else:
        from pymc_bart_amd.sampler import PyBartSettings, PySampler, default_backend

        be = default_backend(local_rank)
        w, step = make_chain(wname, wkw, seed, be)
        X, Y = w["X"], w["Y"]
        n = X.shape[0]
        if args.response != "constant":
            w["name"] += f", response={args.response}"
        samplers = [step.sampler]
        batch_trees = step.settings.batch_sizes()[0 if tune else 1]

        def extra_chains(count, first_chain):
            """More independent chains on this GPU, each on its own HIP stream."""
            out = []
            for c in range(count):  # (every sampler takes a stream of its own: TorchHipMemory.sampler_stream)
                stc = PyBartSettings.from_data(X, Y, m=w["m"], num_particles=w["num_particles"],
                                               seed=seed + 1000 * (first_chain + c), family=w["family"],
                                               n_outputs=w.get("K", 1), response=args.response)
                sc = PySampler(stc, step._X, Y, np.zeros(X.shape[1], np.int32), np.ones(X.shape[1]), backend=be)
                sc.set_likelihood([1.0] if w["family"] == "normal" else [])
                out.append(sc)
            torch.cuda.synchronize()
            return out

        samplers += extra_chains(args.chains_per_gpu - 1, 1)
        for q in samplers:
            q.set_likelihood([1.0] if w["family"] == "normal" else [])

    # ---- burn-in (tune=1) and warm-up (untimed)
    t_burn = time.perf_counter()
    if args.burnin > 0:
        run_all(samplers, True, args.burnin)
    step.tune = tune
    for _ in range(args.warmup):
        step.astep(None)
    if len(samplers) > 1 and args.warmup > 0:
        run_all(samplers[1:], tune, args.warmup)
    burn_s = time.perf_counter() - t_burn

    def aggregate(blocks):
        """Whole-job aggregate per block: max time over ranks, sum of units over ranks."""
        agg = []
        for el, dc in blocks:
            dt_max = allreduce([el], "MAX")[0]
            ps, tu, rt = allreduce([dc["particle_steps"], dc["tree_updates"], dc["rows_touched"]], "SUM")
            agg.append((dt_max, {"particle_steps": ps, "tree_updates": tu, "rows_touched": rt}))
        return agg

    # ---- headline: blocks of exactly `steps` PGBART.astep calls (one chain per GPU: the step method is the
    #      chain).  With several chains per GPU the host-synchronous astep cannot overlap them: there the
    #      headline stays the resident path, as before.
    multi = args.chains_per_gpu > 1
    if multi:
        blocks = resident_blocks(samplers, tune, args.steps, max(1, args.repeats or 5), barrier)
    else:
        blocks = astep_blocks(step, args.steps, args.repeats, barrier, args.min_seconds,
                              agree=lambda v: allreduce([v], "MAX")[0])
    agg = aggregate(blocks)
    (dt_med, u_med), v_min, v_max = median_block(agg)
    per_rank_ms = None
    if dist is not None:
        mine = torch.tensor([float(np.median([b[0] for b in blocks])) * 1e3 / args.steps], dtype=torch.float64,
                            device=cdev)
        outs = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(outs, mine)
        per_rank_ms = [float(o.item()) for o in outs]
    if not dry and not multi:
        step._batches.clear()  # (the history of a long timed run is not needed afterwards)

    line = {
        "metric": METRIC[wname] if default_cfg else f"particle-steps/sec ({w['name']})",
        "value": u_med["particle_steps"] / dt_med,
        "unit": "particle-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt_med * 1e3 / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic" if not dry else "dry-run",
        "config": {
            "workload": w["name"] + f", sigma=1 fixed, tune={int(tune)}, {batch_trees} trees per step",
            "path": "resident (pgb_step_async), several chains per GPU" if multi else
                    "PGBART.astep: host outputs + tree export + stats + history per step",
            "chains": world * args.chains_per_gpu, "chains_per_gpu": args.chains_per_gpu,
            "parallelism": f"chains{world * args.chains_per_gpu}",
            "burnin_asteps_tune1": args.burnin,
        },
        "repeats": len(agg),
        "timed_seconds": float(sum(b[0] for b in agg)),
        "value_min": v_min, "value_max": v_max,
        "ranks_reported_by_collective": ranks_reported,
        "tree_updates_per_s": u_med["tree_updates"] / dt_med,
        "rows_touched_per_tree": u_med["rows_touched"] / max(u_med["tree_updates"], 1.0),
        "particle_steps_per_tree": u_med["particle_steps"] / max(u_med["tree_updates"], 1.0),
        "burnin_seconds": burn_s,
    }
    if per_rank_ms is not None:
        line["per_rank_ms_per_step"] = per_rank_ms
    if dry:
        line["metric"] = "DRY RUN -- plumbing only, not a measurement"
        line["value"] = None

    s = samplers[0]
    solo = world == 1 and dist is None and args.chains_per_gpu == 1 and not dry
    K_out = w.get("K", 1)
    if not dry:
        line["algorithmic_GBps_whole_step"] = workloads.bytes_per_tree_update(
            n, line["rows_touched_per_tree"], K=K_out) * u_med["tree_updates"] / dt_med / 1e9

    # ---- the device-resident path (no host outputs): what round 1 / 2 quoted as the headline
    if not multi and not args.no_extras:
        (el_r, u_r), r_min, r_max = median_block(aggregate(resident_blocks(samplers, tune, args.steps, 5, barrier)))
        if not dry:
            line["resident_path"] = {
                "value": u_r["particle_steps"] / el_r, "unit": "particle-steps/s", "ms_per_step": el_r * 1e3 / args.steps,
                "value_min": r_min, "value_max": r_max, "repeats": 5,
                "astep_fraction_of_resident": line["value"] / (u_r["particle_steps"] / el_r),
                "note": NOTES + "#resident-path",
            }

    # ---- roofline of the dominant kernel + per-kernel shares: a further block with events attached
    if not dry and not args.no_roofline and rank == 0:
        line.update(rooflines(wname, w, X.shape, kernel_profile(s, tune, args.steps), args.response, default_cfg))
        if line.get("roofline"):
            # SURVEY.md 8(d)'s own definition, over the WALL clock of the timed asteps (host outputs and all):
            # sum of B_tree / seconds / 8 TB/s -- next to `frac`, which is the dominant kernel's
            line["roofline"]["whole_step_frac"] = line["algorithmic_GBps_whole_step"] / HBM_PEAK_GBS

    # ---- tune=1 block (reported separately: SURVEY.md 8d)
    if solo and not args.no_extras and not tune:
        run_all(samplers, True, 2)
        (el_t, u_t), t_min, t_max = median_block(resident_blocks(samplers, True, args.steps, 3, barrier))
        line["tune1"] = {"value": u_t["particle_steps"] / el_t, "unit": "particle-steps/s", "path": "resident",
                         "ms_per_step": el_t * 1e3 / args.steps, "value_min": t_min, "value_max": t_max,
                         "tree_updates_per_s": u_t["tree_updates"] / el_t}
        # ... and through PGBART.astep itself, as the headline is ("tune=False and tune=True reported
        # separately"; a tuning astep publishes no history but pays the running-sd pass and the weight update)
        step.tune = True
        (el_a, u_a), a_min, a_max = median_block(astep_blocks(step, args.steps, 5, barrier))
        step.tune = tune
        line["tune1"]["astep"] = {"value": u_a["particle_steps"] / el_a, "unit": "particle-steps/s", "path": "PGBART.astep",
                                  "ms_per_step": el_a * 1e3 / args.steps, "value_min": a_min, "value_max": a_max,
                                  "tree_updates_per_s": u_a["tree_updates"] / el_a, "repeats": 5}

    # ---- end-of-run gather (the only collective; outside the timed region): what a user's run hands to rank 0 --
    #      `--gather-draws` kept draws of this chain (100 x n doubles = 80 MB per rank at cfg2), its tree history as
    #      packed records and its variable-inclusion stats -- through chains.gather_chains itself
    if dist is not None:
        line.update(end_of_run_gather(step, dist, rank, world, dry, args, seed, barrier))

    # ---- informational: PyMC's default of 4 chains, run concurrently on ONE GPU (never the headline)
    if solo and not args.no_multichain:
        ss4 = samplers + extra_chains(3, 1)
        run_all(ss4[1:], True, min(args.burnin, 20))
        run_all(ss4, False, 2)
        (el4, u4), m_min, m_max = median_block(resident_blocks(ss4, False, args.steps, 3, barrier))
        line["concurrent_chains"] = {"chains_per_gpu": 4, "value": u4["particle_steps"] / el4, "path": "resident",
                                     "unit": "particle-steps/s", "ms_per_step": el4 * 1e3 / args.steps,
                                     "value_min": m_min, "value_max": m_max,
                                     "note": "4 independent chains on one GPU, one HIP stream each"}
        del ss4

    # ---- the other single-GPU configurations of BASELINE.json, under the same clock: GPU legs only here,
    #      back to back with the headline's; their CPU samples follow after every GPU leg of the run
    pending_cpu = []
    if solo and args.workload == "cfg2" and default_cfg and not args.no_extras and not args.no_workloads and not tune:
        del samplers, s
        step.sampler = None
        line["workloads"] = {}
        for wn in ("cfg4", "cfg5"):
            d, wl = workload_leg(wn, make_chain, be, args, torch)
            line["workloads"][wn] = d
            pending_cpu.append((wn, d, wl))
    # ---- GPU rate at the chain age of the CPU sample (10 tune asteps of burn-in instead of 100: the CPU leg
    #      cannot afford 100) so that the ratio compares like with like
    matched = None
    if rank == 0 and world == 1 and solo and not args.no_cpu_baseline and default_cfg and wname == "cfg2":
        _, stm = make_chain(wname, wkw, seed, be)
        sm = stm.sampler
        run_all([sm], True, CPU_BURN_CFG2)
        run_all([sm], False, 1)
        (el_m, u_m), _, _ = median_block(resident_blocks([sm], False, args.steps, 3, barrier))
        matched = {"gpu_value": u_m["particle_steps"] / el_m, "path": "resident", "burnin_asteps_tune1": CPU_BURN_CFG2,
                   "rows_touched_per_particle_step": u_m["rows_touched"] / max(u_m["particle_steps"], 1)}
        del sm
        stm.sampler = None
    gpu_done_s = time.perf_counter() - t_start

    # ---- CPU baselines, after every GPU leg
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not dry:
        cpu = cpu_baseline(wname, wkw, seed, args.cpu_budget, response=args.response)
        line["cpu_baseline"] = cpu
        line["speedup_vs_cpu_baseline"] = line["value"] / cpu["value"]
        line["speedup_vs_cpu_rows_touched_per_s"] = (line["tree_updates_per_s"] * line["rows_touched_per_tree"]
                                                     / cpu["rows_touched_per_s"])
        if matched:
            matched["speedup_vs_cpu_baseline"] = matched["gpu_value"] / cpu["value"]
            line["cpu_baseline"]["matched_age"] = matched
        if "all_cores" in cpu:
            line["speedup_vs_cpu_all_cores"] = line["value"] / cpu["all_cores"]["value"]
        for wn, d, wl in pending_cpu:
            d["cpu_baseline"] = cpu_baseline_short(wl, wn, 3415, min(args.cpu_budget, 8.0))
            d["speedup_vs_cpu_baseline"] = d["value"] / d["cpu_baseline"]["value"]
            if d.get("chain_start"):
                d["cpu_baseline"]["matched_age"] = dict(
                    d.pop("chain_start"), speedup_vs_cpu_baseline=None)
                ma = d["cpu_baseline"]["matched_age"]
                ma["speedup_vs_cpu_baseline"] = ma["gpu_value"] / d["cpu_baseline"]["value"]
        if default_cfg and wname == "cfg2" and not args.no_extras:
            line["cfg1"] = cpu_cfg1()
    line["gpu_legs_seconds"] = gpu_done_s

    # ---- the last object of the line: every headline figure again, compact (a kept log tail holds it)
    def brief(d):
        rf, rr = d.get("roofline") or {}, d.get("roofline_rows") or {}
        kk = d.get("roofline_kernels") or {}
        b = {"value": _r(d.get("value")), "ms": _r(d.get("ms_per_step")),
             "resident": _r((d.get("resident_path") or {}).get("value")),
             "kernel": rf.get("kernel"), "frac": _r(rf.get("frac")),
             "step_frac": _r(rf.get("whole_step_frac")), "tu_s": _r(d.get("tree_updates_per_s")),
             "stale": True if (rf.get("traffic_stale") or rr.get("traffic_stale")) else None,
             "cpu8": _r(((d.get("cpu_baseline") or {}).get("all_cores") or {}).get("value")),
             "tune1": _r(((d.get("tune1") or {}).get("astep") or {}).get("value")),
             "rows_hbm": _r(rr.get("frac") if rr else (rf.get("measured_hbm_frac") if rf.get("bound") == "hbm" else None)),
             "us": {k: _r(v["avg_us"]) for k, v in kk.items()},
             "cpu": _r((d.get("cpu_baseline") or {}).get("value")),
             "x_cpu": _r(d.get("speedup_vs_cpu_baseline")),
             "x_same_age": _r(((d.get("cpu_baseline") or {}).get("matched_age") or {}).get("speedup_vs_cpu_baseline"))}
        return {k: v for k, v in b.items() if v is not None}

    summ = {"cfg2" if default_cfg and wname == "cfg2" else wname: brief(line)}
    if "concurrent_chains" in line:
        summ[next(iter(summ))]["chains4"] = _r(line["concurrent_chains"]["value"])
    for wn, d in (line.get("workloads") or {}).items():
        summ[wn] = brief(d)
    if "cfg1" in line:
        summ["cfg1"] = {"cpu": _r(line["cfg1"]["value"]), "tu_s": _r(line["cfg1"]["tree_updates_per_s"])}
    line["summary"] = summ

    if rank == 0:
        print(json.dumps(line))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def end_of_run_gather(step, dist, rank, world, dry, args, seed, barrier):
    """Draw ``--gather-draws`` more posterior draws on every rank (PGBART.astep, tune=0: the draws, their trees,
    their stats), then ONE ``chains.gather_chains`` to rank 0, timed.  Rank 0 checks what arrived: one entry per
    rank, in rank order, keyed 3415 + rank, no two chains alike, every history as long as its draws."""
    from pymc_bart_amd.chains import gather_chains
    from pymc_bart_amd.utils import _decode_vi

    G = max(1, int(args.gather_draws))
    if dry:
        width, p = 8, 4
        mu = np.arange(G, dtype=np.float64)[:, None] + 1000.0 * rank + np.zeros((G, width))
        vi_stats = ["AAAAAA=="] * G
        vi = np.zeros((G, p), np.int64)
        history = (None, [b"dry"] * G)
        counters = step.counters
    else:
        width = int(np.prod(step.shape))
        G = max(1, min(G, (256 << 20) // (8 * width)))  # at most 256 MiB of draws per rank (cfg5: 32 draws of 4 x 250k)
        p = step.num_variates
        step.tune = False
        step.reset_history()  # the history that travels is the history of the draws that travel
        mu = np.empty((G, width))
        vi_stats = []
        for d in range(G):
            m_d, stats = step.astep(None)
            mu[d] = np.asarray(m_d).reshape(-1)
            vi_stats.append(stats[0]["variable_inclusion"])
        vi = np.array([_decode_vi(v, p) for v in vi_stats], dtype=np.int64).reshape(-1, p)
        history = (step.history[0], list(step.history[1]))
        counters = step.counters
    result = {"chain": rank, "seed": seed, "mu": mu, "sigma": np.ones(G), "variable_inclusion": vi_stats,
              "vi_counts": vi, "history": history, "counters": counters}
    tm = {}
    barrier()
    g0 = time.perf_counter()
    got = gather_chains(result, dist, dst=0, force_collective=True, timings=tm)
    barrier()
    gather_s = time.perf_counter() - g0
    out = {"gather_ms": gather_s * 1e3, "gather_draws": G, "gather_bytes_per_rank": tm["dense_bytes"],
           "gather_stages_ms": {k: tm[k] for k in ("h2d_ms", "collective_ms", "object_ms", "d2h_ms")},
           # the collective alone, device to device: the shards of all ranks land on rank 0
           "gather_collective_GBps": world * tm["dense_bytes"] / max(tm["collective_ms"] * 1e-3, 1e-9) / 1e9}
    if rank == 0:
        assert len(got) == world, (len(got), world)
        for r, item in enumerate(got):
            assert item["chain"] == r and item["seed"] == 3415 + r, (r, item["chain"], item["seed"])
            assert item["mu"].shape == (G, width) and len(item["history"][1]) == G == len(item["variable_inclusion"])
        assert np.array_equal(got[0]["mu"], mu)  # rank 0's own shard came back as it went
        for r in range(1, world):
            assert not np.array_equal(got[r]["mu"], got[0]["mu"]), "chains must be independent"
            assert all(not np.array_equal(got[r]["mu"], got[q]["mu"]) for q in range(1, r)), "chains must be independent"
        out["gather_history_bytes_per_rank"] = int(sum(len(getattr(b, "raw", b)) for b in history[1]))
        out["gather_chains_checked"] = world
    return out


def _r(x, sig=4):
    """Round to ``sig`` significant digits (the compact summary)."""
    if x is None:
        return None
    x = float(x)
    if x == 0.0 or not np.isfinite(x):
        return x
    return float(f"{x:.{sig}g}")


CPU_BURN_CFG2 = 10  # tune=1 asteps before the cfg2 CPU sample (50 s per 100 on one core): see matched_age


def workload_leg(wn, make_chain, be, args, torch):
    """GPU legs of one further BASELINE configuration, same protocol as the headline: burn-in with tune=1,
    warm-up, blocks of PGBART.astep until >= --min-seconds are timed, the resident path, the per-kernel profile,
    and the rate at the START of a chain (one tree per step: what the bounded CPU sample of this workload
    runs).  Returns (leg, workload data); the CPU sample itself is taken after every GPU leg of the run."""
    def sync():
        torch.cuda.synchronize()

    t0 = time.perf_counter()
    w, st = make_chain(wn, dict(seed=3415), 3415, be)
    s = st.sampler
    setup_s = time.perf_counter() - t0
    steps = max(4, args.steps // 2)
    run_all([s], True, args.burnin)
    st.tune = False
    for _ in range(2):
        st.astep(None)
    blocks = astep_blocks(st, steps, 0, sync, min_seconds=args.min_seconds, max_blocks=200)
    (el, u), vmin, vmax = median_block(blocks)
    st._batches.clear()
    d = {
        "metric": METRIC[wn], "value": u["particle_steps"] / el, "unit": "particle-steps/s", "steps": steps,
        "ms_per_step": el * 1e3 / steps, "repeats": len(blocks), "timed_seconds": float(sum(b[0] for b in blocks)),
        "value_min": vmin, "value_max": vmax, "dtype": "f64", "data": "synthetic",
        "config": {"workload": w["name"] + f", tune=0, {st.settings.batch_sizes()[1]} trees per step",
                   "path": "PGBART.astep", "burnin_asteps_tune1": args.burnin},
        "tree_updates_per_s": u["tree_updates"] / el,
        "rows_touched_per_tree": u["rows_touched"] / max(u["tree_updates"], 1),
        "particle_steps_per_tree": u["particle_steps"] / max(u["tree_updates"], 1),
        "setup_seconds": setup_s,
    }
    (el_r, u_r), r_min, r_max = median_block(resident_blocks([s], False, steps, 3, sync))
    d["resident_path"] = {"value": u_r["particle_steps"] / el_r, "unit": "particle-steps/s",
                          "ms_per_step": el_r * 1e3 / steps, "value_min": r_min, "value_max": r_max,
                          "astep_fraction_of_resident": d["value"] / (u_r["particle_steps"] / el_r)}
    if not args.no_roofline:
        d.update(rooflines(wn, w, w["X"].shape, kernel_profile(s, False, steps)))
        from pymc_bart_amd import workloads as _wl

        d["algorithmic_GBps_whole_step"] = _wl.bytes_per_tree_update(
            w["X"].shape[0], d["rows_touched_per_tree"], K=w.get("K", 1)) * d["tree_updates_per_s"] / 1e9
        if d.get("roofline"):
            d["roofline"]["whole_step_frac"] = d["algorithmic_GBps_whole_step"] / HBM_PEAK_GBS
    del s
    st.sampler = None  # free the chain's HBM before the next chain
    if not args.no_cpu_baseline:
        # the chain age of the CPU sample: a fresh chain, one tree per step, one warm-up tree update
        _, st1 = make_chain(wn, dict(seed=3415), 3415, be, batch=(1, 1), wl=w)
        s1 = st1.sampler
        run_all([s1], False, 1)
        cs_blocks = resident_blocks([s1], False, 8, 3, sync)
        (el_c, u_c), _, _ = median_block(cs_blocks)
        d["chain_start"] = {"gpu_value": u_c["particle_steps"] / el_c, "path": "resident, one tree per step",
                            # tree updates the chain has seen when the timed blocks end (1 warm-up + what ran)
                            "tree_updates": 1 + int(sum(b[1]["tree_updates"] for b in cs_blocks)),
                            "tree_updates_in_reported_block": int(u_c["tree_updates"]),
                            "rows_touched_per_particle_step": u_c["rows_touched"] / max(u_c["particle_steps"], 1)}
        del s1
        st1.sampler = None
    return d, w


if __name__ == "__main__":
    main()
