#!/usr/bin/env python3
"""Generate tests/golden/oracle_runs.json: exact fingerprints of the CPU oracle on the seeded
parity cases of tests/_cases.py.  The HIP backend must reproduce them bit for bit
(tests/test_parity_gpu.py), and the oracle itself is regression-pinned against them on CPU
(tests/test_oracle_golden.py)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from _cases import CASES, digest, make_case, run_case  # noqa: E402
from _oracle import oracle_backend  # noqa: E402

out = {}
for name in CASES:
    res = run_case(make_case(name), oracle_backend())
    out[name] = digest(res)
    print(name, out[name]["sha256"][:16], out[name]["counters"])
json.dump(out, open(os.path.join(HERE, "oracle_runs.json"), "w"), indent=1)
