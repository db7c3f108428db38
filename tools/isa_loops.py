#!/usr/bin/env python3
"""Loops of one kernel in the built code object: length, fp64 / FMA / memory instruction counts and the
instruction histogram of the longest inner loops.  usage: python tools/isa_loops.py 'k_loglikILi4'  (a substring
of the mangled kernel name).  Works on the CPU box (llvm-objdump of the gfx950 code object)."""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "pymc_bart_amd", "csrc", "libpgbart_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble():
    tmp = tempfile.mkdtemp(dir=os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None)
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(SO, so)
        subprocess.run([OBJDUMP, "--offloading", so], cwd=tmp, capture_output=True)
        co = [f for f in os.listdir(tmp) if "gfx950" in f][0]
        return subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(tmp, co)], capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    want = sys.argv[1]
    text = disassemble().splitlines()
    start = next(i for i, l in enumerate(text) if re.match(r"^[0-9a-f]+ <", l) and want in l)
    end = next((i for i in range(start + 1, len(text)) if re.match(r"^[0-9a-f]+ <", text[i])), len(text))
    ins = []
    for l in text[start:end]:
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):.*?(?:<[^+>]+\+0x([0-9a-f]+)>)?\s*$", l)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), int(m.group(4), 16) if m.group(4) else None))
    base = ins[0][0]
    idx = {a: i for i, (a, _, _) in enumerate(ins)}
    print(text[start], "instructions:", len(ins))
    loops = []
    for i, (a, op, t) in enumerate(ins):
        if (op.startswith("s_cbranch") or op == "s_branch") and t is not None and base + t < a and base + t in idx:
            loops.append((idx[base + t], i))
    for j, i in loops:
        body = [op for _, op, _ in ins[j:i + 1]]
        print(f"loop {j:5d}..{i:5d} len {i - j + 1:5d}  f64 {sum('f64' in b for b in body):4d}  fma/fmac {sum(b.startswith(('v_fma_f64', 'v_fmac_f64')) for b in body):4d}"
              f"  mov_b64 {sum(b.startswith('v_mov_b64') for b in body):4d}  cndmask {sum(b.startswith('v_cndmask') for b in body):4d}"
              f"  mem {sum(b.startswith(('global_', 'ds_', 'scratch_', 'flat_')) for b in body):3d}")
    if len(sys.argv) > 3:
        lo, hi = int(sys.argv[2]), int(sys.argv[3])
        print(collections.Counter(op for _, op, _ in ins[lo:hi + 1]).most_common(30))


if __name__ == "__main__":
    main()
