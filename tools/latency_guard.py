#!/usr/bin/env python3
"""Kernel-latency guard (GPU box).  The occupancy guard sees registers, not instruction scheduling: in round 3 a
one-line change of the leaf_sd statement in k_ctrl re-ordered the kernel's first loads and cost 0.34 us per launch
(5 % of the control kernel, 2 % of the headline) with identical register counts.  This runs the cfg2 headline
protocol briefly and compares the event-timed averages of the slot kernels with profiles/latency_budget.json
(measured value x 1.04); exit code 1 on a regression.  `--write` records the current values.
usage: python tools/latency_guard.py [--write]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUDGET = os.path.join(ROOT, "profiles", "latency_budget.json")


def measure():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--repeats", "20",
           "--no-extras", "--no-cpu-baseline", "--no-multichain", "--no-workloads"]
    runs = []
    for _ in range(3):
        out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=600).stdout
        d = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
        runs.append({k: v["avg_us"] for k, v in d["roofline_kernels"].items()})
    return {k: sorted(r[k] for r in runs)[1] for k in runs[0]}       # median of three runs


def main():
    cur = measure()
    if "--write" in sys.argv:
        json.dump({"protocol": "bench.py --steps 20 --warmup 5 --repeats 20 (cfg2), median of 3 runs, HIP events",
                   "avg_us": cur, "tolerance": 1.04}, open(BUDGET, "w"), indent=1)
        print("wrote", BUDGET, cur)
        return 0
    b = json.load(open(BUDGET))
    bad = [f"{k}: {cur[k]:.2f} us > {b['avg_us'][k]:.2f} us x {b['tolerance']}" for k in b["avg_us"]
           if k in cur and cur[k] > b["avg_us"][k] * b["tolerance"]]
    print("latency guard:", {k: round(v, 2) for k, v in cur.items()}, "budget", {k: round(v, 2) for k, v in b["avg_us"].items()})
    for line in bad:
        print("LATENCY REGRESSION:", line)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
