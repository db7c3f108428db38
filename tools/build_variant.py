#!/usr/bin/env python3
"""Build an experiment variant of the HIP library next to the product build:
  python tools/build_variant.py NAME [-DFOO=1 ...]   ->  build/variants/libpgbart_hip_NAME.so
(same flags as __graft_entry__.build plus the given ones).  Run a bench against it with
PGBART_HIP_LIB=<that path>; variants are git-ignored (build/) but travel to the GPU box."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
vdir = os.path.join(ROOT, "build", "variants")  # (csrc/ holds the product library only)
os.makedirs(vdir, exist_ok=True)
out = os.path.join(vdir, f"libpgbart_hip_{name}.so")
subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *g.HIPCC_FLAGS, *extra, g.HIP_SRC, "-o", out])
print(out)
