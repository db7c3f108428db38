"""A PyMC worker process in miniature, on the GPU: build the step method, tune, draw, print a digest.
Started several at a time by tests/test_multiprocess_gpu.py (one GPU shared by the processes of
``pm.sample(chains=C, cores=C)``, reference bart.py:133-135 / tests/test_bart.py:84-104)."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def chain_digest(chain: int, tune: int = 25, draws: int = 15, backend=None) -> dict:
    import numpy as np

    from pymc_bart_amd import BARTOp, PGBART

    rng = np.random.default_rng(3)
    X = rng.normal(size=(6000, 6))
    Y = np.sin(X[:, 0]) + 0.5 * X[:, 1] + rng.normal(0, 0.2, 6000)
    op = BARTOp(X, Y, m=20)
    kw = {} if backend is None else {"backend": backend}
    step = PGBART([op], num_particles=10, random_seed=11, chain=chain, **kw)
    h = hashlib.sha256()
    for it in range(tune + draws):
        if it == tune:
            step.stop_tuning()
        mu, stats = step.astep(None)
        h.update(np.ascontiguousarray(mu, np.float64).tobytes())
        h.update(np.ascontiguousarray(stats[0]["variable_inclusion"]).tobytes())
    base, batches = op.all_trees[-1]
    for ta in [base] + list(batches):
        for f in ("var", "split", "value", "count"):
            h.update(np.ascontiguousarray(getattr(ta, f)).tobytes())
    return {"chain": chain, "sha256": h.hexdigest(), "n_batches": len(batches),
            "backend": step.sampler.backend.lib.backend_name}


if __name__ == "__main__":
    print("GPU_CHAIN " + json.dumps(chain_digest(int(sys.argv[1]))), flush=True)
