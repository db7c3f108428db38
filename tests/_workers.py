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


def register_at_barrier(op, chain, draws, barrier):
    """A worker whose first draw -- the one that registers the chain's history entry on the op -- starts at the same
    instant as every other worker's (PGBART._publish under contention)."""
    from _oracle import oracle_backend
    from pymc_bart_amd.pgbart import PGBART

    step = PGBART([op], num_particles=4, random_seed=11, chain=chain, backend=oracle_backend())
    for _ in range(2):
        step.astep(None)
    step.stop_tuning()
    barrier.wait(60)
    for _ in range(draws):
        step.astep(None)
    # nothing that is on the shared list is kept here as well
    assert len(step._batches) == (0 if getattr(step, "_shared", None) is not None else draws)


class HistoryBox:
    """A cross-process history container that is NOT a multiprocessing list: append / len / get / set only (the
    re-assignment fallback of PGBART._publish)."""

    def __init__(self):
        self.items = []

    def append(self, x):
        self.items.append(x)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]

    def __setitem__(self, i, x):
        self.items[i] = x


def box_manager():
    from multiprocessing.managers import BaseManager

    class BoxManager(BaseManager):
        pass

    BoxManager.register("Box", HistoryBox, exposed=("append", "__len__", "__getitem__", "__setitem__"))
    return BoxManager()
