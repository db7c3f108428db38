"""Several PROCESSES on one GPU, as `pm.sample(chains=C, cores=C)` runs its chains (reference bart.py:133-135:
one history list shared by the workers; tests/test_bart.py:84-104: two chains): each child builds its own step
method on cuda:0 and samples while the others do.  Every chain must come out exactly as it does alone in this
process -- a chain's draws depend on (seed, chain) only, not on what else shares the GPU or the host."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_three_worker_processes_share_the_gpu_and_reproduce_their_chains(hip):
    from _gpu_chain_child import chain_digest

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    child = os.path.join(ROOT, "tests", "_gpu_chain_child.py")
    procs = [subprocess.Popen([sys.executable, child, str(c)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=env, cwd=ROOT) for c in range(3)]      # all three alive at once
    got = {}
    for c, p in enumerate(procs):
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, (c, out[-1000:], err[-3000:])
        line = [ln for ln in out.splitlines() if ln.startswith("GPU_CHAIN ")]
        assert len(line) == 1, out[-2000:]
        got[c] = json.loads(line[0][len("GPU_CHAIN "):])
    for c in range(3):
        assert got[c]["backend"] == "hip-gfx950" and got[c]["n_batches"] == 15
        alone = chain_digest(c, backend=hip)
        assert got[c]["sha256"] == alone["sha256"], c
    assert len({got[c]["sha256"] for c in range(3)}) == 3               # and the chains differ from each other
