"""Regression pin of the CPU oracle against its committed fingerprints (tests/golden/oracle_runs.json).
The GPU parity tests compare the HIP backend with the same file."""
import json
import os

import pytest

from _cases import CASES, digest, make_case, run_case

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_runs.json")))


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden_fingerprint(oracle, name):
    got = digest(run_case(make_case(name), oracle))
    assert got == GOLD[name]
