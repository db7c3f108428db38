#!/usr/bin/env python3
"""Parity hunt on a GPU box: random configurations (tests/_cases.random_case) through the HIP library and
the CPU oracle, bit-for-bit digests compared.
usage: python tools/fuzz_hunt.py FIRST_SEED COUNT [SECONDS] [large] [compat] [mk] [migrate]
(large: n = 50k .. 1M, few trees; compat: the upstream-semantics switches on, PGB_COMPAT_* = 1 + seed % 3;
mk: only the configurations with K-vector leaves -- the seeds of the others are skipped;
migrate: the GPU chain is moved to the oracle and back through the chain image -- pgb_checkpoint_save / _load,
include/pgbart_image.h -- at random steps, and must still be the chain the oracle runs alone)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _cases import digest, random_case, run_case  # noqa: E402
from _oracle import oracle_backend  # noqa: E402
from pymc_bart_amd.sampler import default_backend  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
large = "large" in sys.argv[4:]
compat_on = "compat" in sys.argv[4:]
mk_only = "mk" in sys.argv[4:]
migrate = "migrate" in sys.argv[4:]
import numpy as np  # noqa: E402
hip, orc = default_backend(0), oracle_backend()
t0, bad, done = time.time(), [], 0
fam = {}
for seed in range(first, first + count):
    if time.time() - t0 > budget:
        break
    c = random_case(seed, large, compat=(1 + seed % 3) if compat_on else 0)
    if mk_only and int(c["K"]) < 2:
        continue
    cuts = ()
    if migrate:  # 1..4 cuts, the chain alternating between the backends (the first move is to the oracle)
        r = np.random.default_rng(seed)
        at = sorted(set(int(x) for x in r.integers(1, max(2, c["steps"]), size=int(r.integers(1, 5)))))
        cuts = {a: (orc if i % 2 == 0 else hip) for i, a in enumerate(at)}
    g, o = digest(run_case(c, hip, checkpoint_at=cuts)), digest(run_case(c, orc))
    done += 1
    key = (c["family"], int(c["K"]), str(c.get("response", "constant")))
    fam[key] = fam.get(key, 0) + 1
    if g != o:
        bad.append(seed)
        print("MISMATCH", seed, c["family"], c["X"].shape, c["m"], c["P"], c["K"], c["rules"].tolist(), flush=True)
print(f"fuzz: seeds {first}..{first + done - 1}: {done} configurations, {len(bad)} mismatches {bad}, {time.time() - t0:.0f} s")
print("by (family, K, response):", sorted(fam.items(), key=lambda kv: -kv[1])[:12])
sys.exit(1 if bad else 0)
