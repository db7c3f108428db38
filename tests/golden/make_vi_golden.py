#!/usr/bin/env python3
"""Generate tests/golden/vi_codec.json by RUNNING the reference's own codec.

The reference package cannot be imported here (pymc / pytensor / numba are absent), but the
variable-inclusion codec is pure Python: the two FunctionDefs are extracted from
/root/reference/pymc_bart/utils.py with `ast` and executed in isolation.  Only the resulting
input/output vectors are committed (data, not source).  Run in the build container only --
/root/reference does not exist on the GPU box.
"""
import ast
import base64
import json
import os

import numpy as np

REF = "/root/reference/pymc_bart/utils.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vi_codec.json")


def load_reference_codec():
    src = open(REF).read()
    tree = ast.parse(src)
    wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("_encode_vi", "_decode_vi")]
    assert len(wanted) == 2
    mod = ast.Module(body=wanted, type_ignores=[])
    ns = {"base64": base64}
    exec(compile(mod, REF, "exec"), ns)  # noqa: S102
    return ns["_encode_vi"], ns["_decode_vi"]


def main():
    enc, dec = load_reference_codec()
    rng = np.random.default_rng(3415)
    cases = [
        # the five cases of the reference's own test (tests/test_utils.py:103-109)
        [0, 0, 0], [1] * 10, [4, 0, 1, 0, 2, 0, 3, 0, 0, 0], [100, 50, 0, 1], [1, 2, 4, 8, 16],
        # varint boundaries
        [127, 128, 300, 16384, 0], [0], [127], [128], [16383, 16384, 2097151, 2097152],
        [2**31 - 1, 2**32, 2**40 + 3],
        [],
    ]
    for p in (3, 5, 50, 200):
        cases.append(rng.integers(0, 4, size=p).tolist())
        cases.append(rng.integers(0, 1000, size=p).tolist())
    vectors = []
    for c in cases:
        s = enc(c)
        assert dec(s, len(c)) == list(c)
        vectors.append({"vec": [int(x) for x in c], "b64": s})
    json.dump({"source": "pymc_bart/utils.py:1368-1398 (_decode_vi/_encode_vi) executed via ast",
               "vectors": vectors}, open(OUT, "w"), indent=1)
    print("wrote", OUT, len(vectors), "vectors")


if __name__ == "__main__":
    main()
