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


@pytest.mark.parametrize("name", ["nan_onehot_prior", "ragged_1025", "categorical_k4_cfg5_small",
                                  "logit_nan_onehot", "mix_response", "meanscale_k2_linear"])
def test_checkpoint_resume_does_not_change_the_chain(oracle, name):
    """pgb_checkpoint_save -> destroy -> create -> pgb_checkpoint_load, in tuning and in the draws:
    the resumed chain reproduces the uninterrupted chain's committed fingerprint."""
    c = make_case(name)
    cuts = (1, c["steps"] // 2 - 1, c["steps"] // 2 + 2)
    assert digest(run_case(c, oracle, checkpoint_at=cuts)) == GOLD[name]
