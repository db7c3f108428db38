"""Kernel-level regression guard (round-2 VERDICT, weak #9 / next #8): every kernel instance of the built
gfx950 library keeps the occupancy (resident workgroups per CU) committed in
``profiles/occupancy_budget.json``.  Runs on the build box: it reads the code object's metadata, no GPU."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import occupancy_guard as og  # noqa: E402


def test_occupancy_arithmetic_of_gfx950():
    # one wave per SIMD and 256-thread workgroup; 512 VGPRs per lane, granule 8, at most 8 waves per SIMD
    assert og.workgroups_per_cu(64, 0, 0)["wgs_per_cu"] == 8
    assert og.workgroups_per_cu(88, 0, 0)["wgs_per_cu"] == 5
    assert og.workgroups_per_cu(128, 0, 0)["wgs_per_cu"] == 4
    assert og.workgroups_per_cu(129, 0, 0)["wgs_per_cu"] == 3       # the 128 edge round 2 fell over
    assert og.workgroups_per_cu(168, 0, 0)["wgs_per_cu"] == 3
    assert og.workgroups_per_cu(169, 0, 0)["wgs_per_cu"] == 2       # ... and the 168 edge
    assert og.workgroups_per_cu(100, 32, 0)["wgs_per_cu"] == 3      # AGPRs share the file
    r = og.workgroups_per_cu(64, 0, 64 * 1024)
    assert r["wgs_per_cu"] == 2 and r["limited_by"] == "lds"        # 160 KiB of LDS per CU
    assert og.next_edge(112) == 128 and og.next_edge(136) == 168 and og.next_edge(176) == 256


@pytest.mark.skipif(not os.path.exists(og.SO), reason="libpgbart_hip.so has not been built")
def test_every_kernel_instance_keeps_its_budgeted_occupancy():
    rows = og.table()
    budget = json.load(open(og.BUDGET))
    assert len(rows) >= 40 and {r["kernel"] for r in rows} == set(budget["kernels"])
    bad = og.check(rows, budget)
    assert not bad, "\n".join(bad)
    by = {r["kernel"]: r for r in rows}
    # the instances the three BASELINE configs launch in their slots
    assert by["k_rows<false, true, false, false>"]["wgs_per_cu"] >= 4      # cfg2 row pass
    assert by["k_rows<false, false, false, true>"]["wgs_per_cu"] >= 4      # cfg4 row pass (16-bit order keys)
    assert by["k_loglik<1, 1, false>"]["wgs_per_cu"] >= 5                   # cfg4 probit likelihood
    assert by["k_rows_mk<4, false, true>"]["wgs_per_cu"] >= 3              # cfg5 row pass (round 4: 3 without spills beat 4 with)
    assert by["k_rows_mk<4, false, true>"]["scratch_bytes"] == 0
    assert by["k_loglik<4, 3, false>"]["wgs_per_cu"] >= 3                   # cfg5 softmax likelihood (factorised, round 5)
    assert by["k_loglik<4, 3, false>"]["scratch_bytes"] == 0


def test_a_regression_is_reported():
    rows = [{"kernel": "k_x", "vgpr": 130, "agpr": 0, "vgpr_alloc": 136, "lds_bytes": 0, "wgs_per_cu": 3,
             "limited_by": "vgpr", "scratch_bytes": 16, "vgpr_spills": 0, "sgpr_spills": 0}]
    budget = {"kernels": {"k_x": {"min_wgs_per_cu": 4, "max_scratch_bytes": 0, "max_vgpr_spills": 0}}}
    bad = og.check(rows, budget)
    assert len(bad) == 2 and "3 workgroups/CU" in bad[0] and "scratch" in bad[1]
    rows[0].update(wgs_per_cu=4, scratch_bytes=60, vgpr_spills=13)      # an instance that already spills may drift a little
    assert not og.check(rows, {"kernels": {"k_x": {"min_wgs_per_cu": 4, "max_scratch_bytes": 44, "max_vgpr_spills": 10}}})
    rows[0].update(scratch_bytes=200)
    assert og.check(rows, {"kernels": {"k_x": {"min_wgs_per_cu": 4, "max_scratch_bytes": 44, "max_vgpr_spills": 10}}})
    assert "not in the budget" in og.check(rows, {"kernels": {}})[0]
