"""Module-level worker functions for the tests that start real processes (importable under the
``spawn`` start method)."""

from __future__ import annotations


def run_chain_and_exit(op, chain, tune, draws):
    """What a PyMC worker process does with a step method: tune, draw, exit -- nothing else is called."""
    from _oracle import oracle_backend
    from pymc_bart_amd.pgbart import PGBART

    step = PGBART([op], num_particles=4, random_seed=11, chain=chain, backend=oracle_backend())
    for it in range(tune + draws):
        if it == tune:
            step.stop_tuning()
        step.astep(None)
