#!/usr/bin/env python3
"""Occupancy guard: registers / LDS / scratch of every kernel instance in libpgbart_hip.so, checked
against the budget committed in ``profiles/occupancy_budget.json``.

Why: twice in round 2 a source change pushed a kernel instance over a VGPR edge (128 -> 3 waves per SIMD,
168 -> 2) and cost 25 % before a late profile run noticed.  This runs on the build box (no GPU): it
unbundles the gfx950 code object, reads the ``NT_AMDGPU_METADATA`` note and derives, for the 256-thread
workgroups every kernel here uses (one wave per SIMD and workgroup), how many workgroups a CU keeps
resident: min over VGPRs (512 per SIMD lane, granule 8, at most 8 waves), LDS (160 KiB per CU) and the
hardware's 8 waves per SIMD.

  python tools/occupancy_guard.py            # table + check against the budget (exit 1 on a violation)
  python tools/occupancy_guard.py --write    # (re)write the budget from the current build

``__graft_entry__.build()`` runs the check after compiling; ``tests/test_occupancy_guard.py`` runs it too.
"""

from __future__ import annotations

import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "pymc_bart_amd", "csrc", "libpgbart_hip.so")
BUDGET = os.path.join(ROOT, "profiles", "occupancy_budget.json")
LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")

VGPR_FILE = 512          # per SIMD lane, unified VGPR + AGPR (gfx90a and later, gfx950 included)
VGPR_GRANULE = 8
MAX_WAVES_PER_SIMD = 8
LDS_PER_CU = 160 * 1024  # /opt/skills/guides/MI355X_MICROARCH.md
SIMDS_PER_CU = 4


def code_object_metadata(so_path: str) -> list[dict]:
    """The ``amdhsa.kernels`` list of the gfx950 code object bundled in ``so_path``."""
    import yaml

    tmp = tempfile.mkdtemp(prefix="pgb_occ_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(so_path, local)
        subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", local],
                              stdout=subprocess.DEVNULL, cwd=tmp)
        cos = [f for f in os.listdir(tmp) if "gfx950" in f]
        if len(cos) != 1:
            raise RuntimeError(f"expected one gfx950 code object in {so_path}, found {cos}")
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, cos[0])],
                                        text=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    start = notes.index("---\n") + 4
    end = notes.index("\n...", start) if "\n..." in notes[start:] else len(notes)
    meta = yaml.safe_load(notes[start:end])
    return meta["amdhsa.kernels"]


def demangle(names: list[str]) -> list[str]:
    tool = shutil.which("c++filt") or os.path.join(LLVM, "llvm-cxxfilt")
    out = subprocess.check_output([tool], input="\n".join(names), text=True)
    res = []
    for ln in out.strip().split("\n"):
        ln = ln.split("(")[0]                       # drop the argument list
        if ln.startswith("void "):
            ln = ln[5:]
        res.append(ln.strip())
    return res


def workgroups_per_cu(vgpr: int, agpr: int, lds: int, threads: int = 256) -> dict:
    waves_per_wg = (threads + 63) // 64
    # (.vgpr_count of a kernel that uses accumulation registers is already the unified total: k_probe reports
    #  512 with .agpr_count 256)
    unified = vgpr if agpr and vgpr > 256 else ((vgpr + 3) // 4) * 4 + agpr
    total = -(-unified // VGPR_GRANULE) * VGPR_GRANULE
    waves_simd = max(1, min(MAX_WAVES_PER_SIMD, VGPR_FILE // max(total, VGPR_GRANULE)))
    by_vgpr = waves_simd * SIMDS_PER_CU // waves_per_wg
    by_lds = LDS_PER_CU // lds if lds > 0 else 1 << 30
    by_waves = MAX_WAVES_PER_SIMD * SIMDS_PER_CU // waves_per_wg
    return {"vgpr_alloc": total, "waves_per_simd_by_vgpr": waves_simd, "wgs_per_cu": min(by_vgpr, by_lds, by_waves),
            "limited_by": "vgpr" if by_vgpr <= min(by_lds, by_waves) else ("lds" if by_lds <= by_waves else "waves")}


def next_edge(total: int) -> int:
    """The largest allocation that keeps the current number of waves per SIMD."""
    waves = max(1, min(MAX_WAVES_PER_SIMD, VGPR_FILE // max(total, VGPR_GRANULE)))
    return (VGPR_FILE // waves) // VGPR_GRANULE * VGPR_GRANULE


def table(so_path: str = SO) -> list[dict]:
    kernels = code_object_metadata(so_path)
    names = demangle([k[".name"] for k in kernels])
    rows = []
    for k, name in zip(kernels, names):
        if not name.startswith("k_"):  # library kernels (rocPRIM's radix sort, used once in pgb_set_data) are not ours to budget
            continue
        vg, ag = int(k[".vgpr_count"]), int(k.get(".agpr_count", 0))
        lds = int(k[".group_segment_fixed_size"])
        threads = 64 if name.startswith("k_predict") else 256  # k_predict: one wave per workgroup (+ dynamic LDS)
        occ = workgroups_per_cu(vg, ag, lds, threads)
        rows.append({"kernel": name, "vgpr": vg, "agpr": ag, "sgpr": int(k[".sgpr_count"]), "lds_bytes": lds,
                     "scratch_bytes": int(k[".private_segment_fixed_size"]),
                     "vgpr_spills": int(k.get(".vgpr_spill_count", 0)), "sgpr_spills": int(k.get(".sgpr_spill_count", 0)),
                     "threads": threads, **occ, "vgpr_edge": next_edge(occ["vgpr_alloc"])})
    rows.sort(key=lambda r: r["kernel"])
    return rows


def format_table(rows: list[dict]) -> str:
    head = f"{'kernel instance':<58} {'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'LDS B':>7} {'scratch':>7} {'spill':>5} " \
           f"{'WG/CU':>5} {'by':>5} {'edge':>5}"
    lines = [head, "-" * len(head)]
    for r in rows:
        lines.append(f"{r['kernel'][:58]:<58} {r['vgpr']:>5} {r['agpr']:>5} {r['sgpr']:>5} {r['lds_bytes']:>7} "
                     f"{r['scratch_bytes']:>7} {r['vgpr_spills'] + r['sgpr_spills']:>5} {r['wgs_per_cu']:>5} "
                     f"{r['limited_by']:>5} {r['vgpr_edge']:>5}")
    return "\n".join(lines)


def check(rows: list[dict], budget: dict) -> list[str]:
    """Violations of the committed budget: fewer resident workgroups per CU than budgeted, scratch or spills
    where none were budgeted, an instance the budget does not know (add it with --write)."""
    bad = []
    for r in rows:
        b = budget["kernels"].get(r["kernel"])
        if b is None:
            bad.append(f"{r['kernel']}: not in the budget (run tools/occupancy_guard.py --write and commit it)")
            continue
        if r["wgs_per_cu"] < b["min_wgs_per_cu"]:
            bad.append(f"{r['kernel']}: {r['wgs_per_cu']} workgroups/CU (VGPR {r['vgpr']}+{r['agpr']} -> {r['vgpr_alloc']}, "
                       f"LDS {r['lds_bytes']} B; limited by {r['limited_by']}) < budget {b['min_wgs_per_cu']}")
        # scratch / VGPR spills: an instance that had none must stay without; one that already spills (the generic
        # K <= 8 and linear-response instances) may move by half its budget + a few registers before it is news
        sb, vb = b.get("max_scratch_bytes", 0), b.get("max_vgpr_spills", 0)
        if r["scratch_bytes"] > (sb * 3 // 2 + 32 if sb else 0):
            bad.append(f"{r['kernel']}: {r['scratch_bytes']} B of scratch > budget {sb}")
        if r["vgpr_spills"] > (vb * 3 // 2 + 4 if vb else 0):  # (SGPR spills go to VGPR lanes: cheap, not budgeted)
            bad.append(f"{r['kernel']}: {r['vgpr_spills']} VGPR spills > budget {vb}")
    return bad


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=SO)
    ap.add_argument("--write", action="store_true", help="write the budget from the current build")
    ap.add_argument("--table", default=None, help="also write the table to this file")
    args = ap.parse_args(argv)
    rows = table(args.so)
    txt = format_table(rows)
    print(txt)
    if args.table:
        with open(args.table, "w") as fh:
            fh.write(txt + "\n")
    if args.write:
        budget = {"note": "minimum resident 256-thread workgroups per CU of every kernel instance (tools/occupancy_guard.py); "
                          "a build below it fails __graft_entry__.build() and tests/test_occupancy_guard.py",
                  "kernels": {r["kernel"]: {"min_wgs_per_cu": r["wgs_per_cu"], "max_scratch_bytes": r["scratch_bytes"],
                                            "max_vgpr_spills": r["vgpr_spills"],
                                            "vgpr_at_write": r["vgpr"], "vgpr_edge": r["vgpr_edge"]} for r in rows}}
        with open(BUDGET, "w") as fh:
            json.dump(budget, fh, indent=1, sort_keys=True)
            fh.write("\n")
        print(f"wrote {BUDGET}")
        return 0
    if not os.path.exists(BUDGET):
        print(f"no budget at {BUDGET}; run with --write", file=sys.stderr)
        return 1
    bad = check(rows, json.load(open(BUDGET)))
    for b in bad:
        print("OCCUPANCY REGRESSION:", b, file=sys.stderr)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
