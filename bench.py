#!/usr/bin/env python3
"""bench.py -- particle-steps/sec of the PGBART hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one PGBART.astep (a batch of 10% of the m trees re-sampled by particle Gibbs) on the
configuration BASELINE.json's metric is quoted on: n=100k, p=50, m=200 trees, 40 particles, Gaussian
likelihood (cfg2), synthetic data resident in HBM.

N > 1: ``python bench.py --gpus N`` launches its own N ranks (``torch.distributed.run``, one process
per GPU, RCCL) BEFORE anything touches a GPU; launched under ``torch.distributed.run`` by somebody
else (RANK set) it is one of those ranks.  N independent chains run one per GPU (weak scaling, no
data-path collective); the draws are gathered over RCCL after the timed region.  Rank 0 prints ONE
JSON line.

Protocol (VERDICT r1 "steady state"): burn in ``--burnin`` asteps with tune=1 (default
100 = 10 sweeps over the m trees), switch to tune=0, W untimed warm-up asteps, then ``--repeats``
blocks of EXACTLY K asteps, each bracketed by barrier + synchronize; ``value`` is the median block
(min / max alongside).  Further legs at N=1: ``astep_path`` (PGBART.astep itself: sum_trees to the
host, the step's trees exported, stats encoded), ``tune1``, the per-kernel profile behind
``roofline`` / ``roofline_kernels``, 4 chains on the one GPU, and the CPU baseline (1 core and
8 chains on 8 cores).
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
ROUND = "r02"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps asteps; value = median")
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
    ap.add_argument("--response", default="constant", choices=["constant", "linear", "mix"])
    ap.add_argument("--no-multichain", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only (skip astep_path / tune1 legs)")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of timed CPU work per baseline leg")
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
def _cpu_worker(args_tuple):
    (wname, wkw, seed, budget_s, burn, response, so_path) = args_tuple
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle import NumpyMemory
    from pymc_bart_amd import _abi, workloads
    from pymc_bart_amd.sampler import Backend, PyBartSettings, PySampler

    w = getattr(workloads, wname)(**wkw)
    X, Y = w["X"], w["Y"]
    st = PyBartSettings.from_data(X, Y, m=w["m"], num_particles=w["num_particles"], seed=seed,
                                  family=w["family"], n_outputs=w.get("K", 1), response=response)
    be = Backend(lib=_abi.PGBLibrary(so_path), mem=NumpyMemory())
    s = PySampler(st, X, Y, np.zeros(X.shape[1], np.int32), np.ones(X.shape[1]), backend=be)
    s.set_likelihood([1.0] if w["family"] == "normal" else [])
    for _ in range(burn):
        s.step(True, fetch=False)
    s.step(False, fetch=False)
    c0 = s.sync()
    t0 = time.perf_counter()
    steps = 0
    while True:
        s.step(False, fetch=False)
        steps += 1
        if time.perf_counter() - t0 > budget_s or steps >= 256:
            break
    dt = time.perf_counter() - t0
    c1 = s.sync()
    return {k: c1[k] - c0[k] for k in c1} | {"dt": dt, "steps": steps}


def cpu_baseline(wname, wkw, seed, budget_s, response="constant"):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import multiprocessing as mp

    from _oracle import build_oracle_native

    so_path, flags = build_oracle_native()
    cores = min(8, os.cpu_count() or 1)
    burn = 10 if wname == "cfg2" else 0  # one tune=1 sweep first (a fraction of the GPU burn-in: CPU time)
    ctx = mp.get_context("spawn")
    with ctx.Pool(1) as pool:
        one = pool.map(_cpu_worker, [(wname, wkw, seed, budget_s, burn, response, so_path)])[0]
    many = None
    if cores > 1:
        with ctx.Pool(cores) as pool:
            many = pool.map(_cpu_worker, [(wname, wkw, seed + c, budget_s, burn, response, so_path)
                                          for c in range(cores)])
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), model)
    except OSError:
        pass
    out = {
        "value": one["particle_steps"] / one["dt"], "unit": "particle-steps/s", "cores": 1, "kind": "port",
        "sample": f"{one['steps']} tune=0 asteps ({one['tree_updates']} tree updates, {one['dt']:.1f} s) of the "
                  f"same {wname} data after {burn} tune=1 + 1 tune=0 warm-up asteps; restated CPU baseline "
                  "(oracle/), not the reference binary (bartrs is not installable here); one chain on one "
                  "core, as upstream runs a chain (chains are processes)",
        "compiler": flags,
        "tree_updates_per_s": one["tree_updates"] / one["dt"],
        "rows_touched_per_particle_step": one["rows_touched"] / max(one["particle_steps"], 1),
        "host": {"nproc": os.cpu_count(), "cpu_model": model},
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


def median_block(blocks):
    """blocks: list of (seconds, units dict).  Returns (median block by rate, min rate, max rate)."""
    rates = [b[1]["particle_steps"] / b[0] for b in blocks]
    order = np.argsort(rates)
    mid = blocks[int(order[len(order) // 2])]
    return mid, float(min(rates)), float(max(rates))


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))

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
        wname, wkw = "cfg5", dict(seed=3415, num_particles=args.particles)
    else:
        wname, wkw = "cfg2", dict(seed=3415, n=args.n, p=args.p, m=args.m, num_particles=args.particles)
    tune = bool(args.tune)

    if dry:
        w = {"name": "dry-run (no sampling work)", "family": "normal", "m": args.m}
        n = args.n
        samplers = [DryRunSampler(rank)]
        step = None
        batch_trees = max(1, args.m // 10)
    else:
        from pymc_bart_amd.pgbart import (PGBART, BARTOp, BernoulliLikelihood, CategoricalLikelihood,
                                          NormalLikelihood)
        from pymc_bart_amd.sampler import PyBartSettings, PySampler, default_backend

        w = getattr(workloads, wname)(**wkw)
        X, Y = w["X"], w["Y"]
        n = X.shape[0]
        if args.response != "constant":
            w["name"] += f", response={args.response}"
        be = default_backend(local_rank)
        lik = {"normal": lambda: NormalLikelihood(1.0),  # sigma fixed at 1
               "bernoulli_probit": lambda: BernoulliLikelihood("probit"),
               "categorical": lambda: CategoricalLikelihood(w.get("K", 1))}[w["family"]]()
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # response=linear is flagged experimental, as upstream
            op = BARTOp(X, Y, m=w["m"], response=args.response)
        # the step method itself: its sampler is the resident path, its astep the host path
        step = PGBART([op], num_particles=w["num_particles"], likelihood=lik, observed=Y, random_seed=seed,
                      backend=be)
        samplers = [step.sampler]
        batch_trees = step.settings.batch_sizes()[0 if tune else 1]

        def extra_chains(count, first_chain):
            """More independent chains on this GPU, each on its own HIP stream."""
            out = []
            for c in range(count):
                with torch.cuda.stream(torch.cuda.Stream()):
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

    def run_all(ss, tn, k):
        """k asteps of every chain: started asynchronously (each handle's worker thread feeds its own
        stream), then waited for.  Returns the summed counters."""
        if k > 0:
            for q in ss:
                q.step_async(tn, k)
        tot = {}
        for q in ss:
            for key, v in q.sync().items():
                tot[key] = tot.get(key, 0) + v
        return tot

    def timed_blocks(ss, tn, steps, repeats):
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

    # ---- burn-in (tune=1) and warm-up (untimed)
    t_burn = time.perf_counter()
    if args.burnin > 0:
        run_all(samplers, True, args.burnin)
    if args.warmup > 0:
        run_all(samplers, tune, args.warmup)
    burn_s = time.perf_counter() - t_burn

    # ---- headline: `repeats` blocks of exactly `steps` asteps
    blocks = timed_blocks(samplers, tune, args.steps, max(1, args.repeats))
    # whole-job aggregate per block: max time over ranks, sum of units over ranks
    agg = []
    for el, dc in blocks:
        dt_max = allreduce([el], "MAX")[0]
        ps, tu, rt = allreduce([dc["particle_steps"], dc["tree_updates"], dc["rows_touched"]], "SUM")
        agg.append((dt_max, {"particle_steps": ps, "tree_updates": tu, "rows_touched": rt}))
    (dt_med, u_med), v_min, v_max = median_block(agg)
    per_rank_ms = None
    if dist is not None:
        mine = torch.tensor([blocks[len(blocks) // 2][0] * 1e3 / args.steps], dtype=torch.float64, device=cdev)
        outs = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(outs, mine)
        per_rank_ms = [float(o.item()) for o in outs]

    line = {
        "metric": "particle-steps/sec (n=100k, p=50, m=200, 40 particles)",
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
            "chains": world * args.chains_per_gpu, "chains_per_gpu": args.chains_per_gpu,
            "parallelism": f"chains{world * args.chains_per_gpu}",
            "burnin_asteps_tune1": args.burnin,
        },
        "repeats": len(agg),
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

    # ---- roofline of the dominant kernel + per-kernel shares: a further block with events attached
    if not dry and not args.no_roofline and rank == 0:
        s.profile(True)
        cp0 = s.sync()
        s.step_async(tune, args.steps)
        cp1 = s.sync()
        ms_rows, launches = s.profile(False)
        kern = s.profile_kernels()
        clk_ms, clk_launches = s.profile_clock()
        tu = cp1["tree_updates"] - cp0["tree_updates"]
        rt = cp1["rows_touched"] - cp0["rows_touched"]
        parts = cp1["partitions"] - cp0["partitions"]
        tot_ms = sum(k["ms"] for k in kern.values()) or 1.0
        line["roofline_kernels"] = {
            name: {"pct": 100.0 * k["ms"] / tot_ms, "avg_us": k["ms"] * 1e3 / k["launches"],
                   "launches": k["launches"], "workgroups": k["workgroups"]}
            for name, k in kern.items()}
        dom = "k_rows"
        pmc = {}
        pmc_path = os.path.join(ROOT, "profiles", f"{ROUND}_pmc_{args.workload}.json")
        default_cfg = (args.workload != "cfg2") or (args.n, args.p, args.m, args.particles) == (100_000, 50, 200, 40)
        if os.path.exists(pmc_path) and default_cfg:
            pmc = json.load(open(pmc_path))
        if ms_rows > 0:
            alg = workloads.bytes_per_tree_update(n, rt / max(tu, 1), K=K_out) * tu
            ach = alg / (ms_rows * 1e-3) / 1e9
            nact = parts / max(launches, 1)
            nchunks = (n + 1023) // 1024
            G = max(1, -(-int(round(nact * nchunks)) // 640))
            ngroups = max(1.0, nact / G)
            # what THIS layout moves per launch: every active particle streams its n labels in and out
            # (1 B each) and the split column (8 B per row); each particle group re-reads {sum_trees, r}
            # (16 B per row); a tree-boundary pass adds the INIT/FINAL streams (~26 B read per row and
            # group, 25 B written per row)
            # (a matrix beyond the Infinity Cache is partitioned on its float32 shadow: 4 B per row)
            shadow = K_out <= 4 and args.response == "constant" and X.shape[1] * (nchunks * 1024) * 8 >= (192 << 20)
            xbytes = 4.0 if shadow else 8.0
            impl = parts * (2.0 + xbytes) * n + launches * ngroups * 16.0 * n + tu * (ngroups * 26.0 + 25.0) * n
            line["roofline"] = {
                "bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS,
                "traffic": (pmc.get(dom) or pmc.get("k_rows_mk") or {}).get("hbm_bytes_per_launch_corrected"),
                "traffic_source": (f"profiles/{ROUND}_pmc_{args.workload}.json: " + pmc.get("command", "")) if pmc else None,
                "launches": launches, "avg_launch_us": ms_rows * 1e3 / max(launches, 1),
                "avg_kernel_us_device_clock": (clk_ms * 1e3 / clk_launches) if clk_launches else None,
                "frac_device_clock": (alg / (clk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if clk_ms > 0 else None,
                "algorithmic_bytes_per_launch": alg / max(launches, 1),
                "implementation_bytes_per_launch_model": impl / max(launches, 1),
                "implementation_GBps_model": impl / (ms_rows * 1e-3) / 1e9,
                "note": "achieved = ALGORITHMIC bytes (sum over tree updates of 48 n + 40 rows_touched, "
                        "SURVEY.md 8d: the traffic of the reference's index-list layout) / total time of the "
                        "dominant kernel from HIP events attached to each dispatch; it is a work rate, not HBM "
                        "utilisation -- at cfg2 the working set lives in L2 / Infinity Cache and `traffic` (PMC, "
                        "per launch) is far below it.  implementation_* is a byte model of what this layout "
                        "actually streams per launch (labels 1 B in + 1 B out and the split column, 8 B per row -- "
                        "4 B from the float32 shadow when the matrix exceeds the Infinity Cache -- of every ACTIVE "
                        "particle, 16 B per row and particle group), most of it served by L2 at cfg2.",
            }
        if args.workload in ("cfg4", "cfg5") and "k_loglik" in kern:
            kl = kern["k_loglik"]
            insts = pmc.get("k_loglik", {}).get("valu_wave_insts_per_launch")
            ginst = (insts * kl["launches"] / (kl["ms"] * 1e-3) / 1e9) if insts else None
            # for these workloads the row pass is NOT the dominant kernel and its algorithmic-byte rate can
            # exceed the HBM peak (the layout moves far fewer bytes than the reference's index lists): the
            # line's `roofline` is the dominant kernel's, the row pass keeps its figures under `roofline_rows`
            if "roofline" in line:
                line["roofline_rows"] = line.pop("roofline")
            line["roofline"] = {
                "bound": "valu-f64-issue", "kernel": "k_loglik", "pct_of_gpu_time": 100.0 * kl["ms"] / tot_ms,
                "achieved": ginst, "peak": VALU_F64_PEAK_GINST, "unit": "G wave-instructions/s",
                "frac": (ginst / VALU_F64_PEAK_GINST) if ginst else None,
                "note": "the per-row log-likelihood pass is the dominant kernel of this workload and is bound by "
                        "fp64 VALU issue (peak = 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 fp64 "
                        "instruction); instructions per launch from the SQ_INSTS_VALU pass under profiles/",
            }

    # ---- PGBART.astep itself: host outputs, tree export, stats (the path pm.sample drives)
    if solo and not args.no_extras and not tune:
        step.tune = False
        for _ in range(2):
            step.astep(None)
        blocks_a = []
        a = s.sync()
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step.astep(None)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            b = s.sync()
            blocks_a.append((el, {k: b[k] - a[k] for k in b}))
            a = b
        (el_a, u_a), a_min, a_max = median_block(blocks_a)
        line["astep_path"] = {
            "value": u_a["particle_steps"] / el_a, "unit": "particle-steps/s", "ms_per_step": el_a * 1e3 / args.steps,
            "value_min": a_min, "value_max": a_max, "fraction_of_resident": (u_a["particle_steps"] / el_a) / line["value"],
            "note": "PGBART.astep(q) per step: synchronous pgb_step_host (sum_trees DMA'd to pinned host memory, "
                    "the step's trees + vi + counters through the mapped block, ONE stream sync), TreeArrays "
                    "built, history published, variable_inclusion encoded",
        }
        step._batches.clear()

    # ---- tune=1 block (reported separately: SURVEY.md 8d)
    if solo and not args.no_extras and not tune:
        run_all(samplers, True, 2)
        (el_t, u_t), t_min, t_max = median_block(timed_blocks(samplers, True, args.steps, 3))
        line["tune1"] = {"value": u_t["particle_steps"] / el_t, "unit": "particle-steps/s",
                         "ms_per_step": el_t * 1e3 / args.steps, "value_min": t_min, "value_max": t_max,
                         "tree_updates_per_s": u_t["tree_updates"] / el_t}

    # ---- end-of-run gather of the draws (the only collective; outside the timed region)
    if dist is not None:
        if dry:
            draw = torch.full((8,), float(rank), dtype=torch.float64)
        else:
            s.step(tune, fetch=False)  # one synchronous step: the device buffer holds this chain's last draw
            draw = s.sum_trees_device().clone()  # (K*n,)
        outs = [torch.empty_like(draw) for _ in range(world)]
        barrier()
        g0 = time.perf_counter()
        dist.all_gather(outs, draw)  # direct all-gather over xGMI: every rank's shard moves in parallel
        if not dry:
            torch.cuda.synchronize()
        line["gather_ms"] = (time.perf_counter() - g0) * 1e3
        line["gather_bytes_per_rank"] = int(draw.numel() * 8)
        if rank == 0 and world > 1:
            assert not torch.equal(outs[0], outs[1]), "chains must be independent"

    # ---- informational: PyMC's default of 4 chains, run concurrently on ONE GPU (never the headline)
    if solo and not args.no_multichain:
        ss4 = samplers + extra_chains(3, 1)
        run_all(ss4[1:], True, min(args.burnin, 20))
        run_all(ss4, False, 2)
        (el4, u4), m_min, m_max = median_block(timed_blocks(ss4, False, args.steps, 3))
        line["concurrent_chains"] = {"chains_per_gpu": 4, "value": u4["particle_steps"] / el4,
                                     "unit": "particle-steps/s", "ms_per_step": el4 * 1e3 / args.steps,
                                     "value_min": m_min, "value_max": m_max,
                                     "note": "4 independent chains on one GPU, one HIP stream each"}
        del ss4

    if rank == 0 and world == 1 and not args.no_cpu_baseline and not dry:
        cpu = cpu_baseline(wname, wkw, seed, args.cpu_budget, response=args.response)
        line["cpu_baseline"] = cpu
        line["speedup_vs_cpu_baseline"] = line["value"] / cpu["value"]
        if "all_cores" in cpu:
            line["speedup_vs_cpu_all_cores"] = line["value"] / cpu["all_cores"]["value"]

    if rank == 0:
        print(json.dumps(line))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
