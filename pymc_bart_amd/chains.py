"""Chain driver: a minimal sampling loop around :class:`PGBART` and the one-chain-per-GPU
sharding of independent chains.

The reference runs chains as PyMC worker processes that return tree history through a
``multiprocessing.Manager().list()`` (``bart.py:133-135``, consumed at ``utils.py:124-127``).
Here each rank of a ``torch.distributed`` job owns one chain on its own MI355X; there is no
communication during sampling and ONE gather at the end (RCCL over xGMI when the backend is
"nccl"; "gloo" in the CPU tests).  PyMC itself is not required: the loop below is the small
part of ``pm.sample`` the hot path needs -- call ``step.astep``, Gibbs-update ``sigma`` of the
Normal likelihood, keep the draws.
"""

from __future__ import annotations

import numpy as np

from .pgbart import PGBART, BARTOp, NormalLikelihood
from .utils import _decode_vi


def sample_chain(op: BARTOp, tune: int, draws: int, num_particles: int = 10, random_seed: int = 0,
                 chain: int = 0, batch=(0.1, 0.1), sigma: float | None = None,
                 sigma_prior=(1.0, 1.0), backend=None, keep_draws: bool = True) -> dict:
    """Run one chain.  ``sigma=None``: sigma^2 ~ InvGamma(a0, b0) is Gibbs-updated from the
    residuals after every step (conjugate stand-in for the HalfNormal+NUTS of the reference
    tests); otherwise sigma is held fixed."""
    rng = np.random.default_rng(np.random.SeedSequence([int(random_seed), int(chain), 77]))
    lik = NormalLikelihood("sigma")
    step = PGBART([op], num_particles=num_particles, batch=batch, likelihood=lik,
                  random_seed=random_seed, chain=chain, backend=backend)
    Y = np.asarray(op.Y, np.float64)
    n = Y.shape[0]
    cur_sigma = float(sigma) if sigma is not None else float(Y.std()) or 1.0
    mu_draws = np.empty((draws, n)) if keep_draws else None
    sig_draws = np.empty(draws)
    vi_stats = []
    for it in range(tune + draws):
        if it == tune:
            step.stop_tuning()
        point = {"sigma": cur_sigma}
        mu, stats = step.astep(None, point)
        if sigma is None:
            res = Y - mu
            a0, b0 = sigma_prior
            cur_sigma = float(np.sqrt((b0 + 0.5 * float(res @ res)) / rng.gamma(a0 + 0.5 * n)))
        if it >= tune:
            d = it - tune
            if keep_draws:
                mu_draws[d] = mu
            sig_draws[d] = cur_sigma
            vi_stats.append(stats[0]["variable_inclusion"])
    p = step.num_variates
    vi = np.array([_decode_vi(s, p) for s in vi_stats], dtype=np.int64).reshape(-1, p)
    step.flush_history()
    return {
        "chain": chain,
        "mu": mu_draws,
        "sigma": sig_draws,
        "variable_inclusion": vi_stats,
        "vi_counts": vi,
        "history": step.history,
        "counters": step.counters,
        "step": step,
    }


def sample_chains(op: BARTOp, chains: int, tune: int, draws: int, **kw) -> list[dict]:
    """Run ``chains`` independent chains CONCURRENTLY on the current GPU (what ``pm.sample(chains=4)``
    does with worker processes upstream).

    One chain keeps the GPU busy with a strictly sequential kernel chain (control kernel -> row
    pass -> control kernel ...), so a single chain is latency-bound at cfg2 sizes.  Chains are
    independent, hence each gets its own HIP stream and a host thread that feeds its state machine
    (the ctypes calls release the GIL); the row pass of one chain overlaps the control kernel of
    another.  Measured on MI355X at cfg2 (resident path, ``BENCH_r05.json``): 1 chain 2.2 M, 4 chains 5.8 M
    particle-steps/s aggregate; the current figures are the ``resident_path`` / ``concurrent_chains`` entries of the
    committed bench line (``tools/show_bench.py``).  The draws of every chain are bit-identical to the ones it
    produces when run alone (``tests/test_parity_gpu.py``)."""
    import threading

    out: list = [None] * chains
    errs: list = []

    def work(c: int) -> None:
        try:
            # (each chain's sampler takes a stream of its own: TorchHipMemory.sampler_stream)
            out[c] = sample_chain(op, tune, draws, chain=c, **kw)
        except BaseException as e:  # noqa: BLE001 - re-raised on the caller's thread
            errs.append(e)

    threads = [threading.Thread(target=work, args=(c,), name=f"pgbart-chain-{c}") for c in range(chains)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errs:
        raise errs[0]
    return out


def gather_chains(result: dict, dist=None, dst: int = 0, force_collective: bool = False, timings: dict | None = None):
    """The single end-of-run collective: gather every rank's draws and tree history on ``dst``.

    Dense draws travel as one tensor per rank with ``gather`` to ``dst`` (over RCCL/xGMI on GPUs every rank's
    shard reaches ``dst`` over its own direct link; only ``dst`` allocates the ``world`` receive buffers -- an
    ``all_gather`` would land 7/8 of the traffic, 7 x 80 MB per rank at cfg2, on ranks that drop it); the small
    ragged pieces (VI strings, tree history) travel pickled with ``gather_object``.  Returns the list of
    per-chain results on ``dst`` and ``None`` elsewhere.  Without a process group it returns ``[result]``; a group of ONE rank
    skips the collectives too unless ``force_collective`` is set (the GPU suite sets it to run the RCCL
    calls of this function at world size 1 before an 8-GPU job meets them for the first time).
    ``timings`` (a dict, filled in place): milliseconds of the stages -- ``h2d_ms`` staging the dense block on the
    device, ``collective_ms`` the ``gather`` itself (device to device), ``object_ms`` the pickled pieces, ``d2h_ms``
    the receive buffers back on the host (``dst`` only) -- and ``dense_bytes`` per rank.
    """
    import time

    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collective):
        return [{k: v for k, v in result.items() if k != "step"}]
    import torch

    world, rank = dist.get_world_size(), dist.get_rank()
    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    # keep_draws=False: only sigma travels densely
    mu = result["mu"] if result["mu"] is not None else np.empty((result["sigma"].shape[0], 0))

    def tick():
        if on_gpu:
            torch.cuda.synchronize()
        return time.perf_counter()

    t0 = tick()
    dense = torch.from_numpy(np.concatenate([mu, result["sigma"][:, None]], axis=1)).to(dev)
    parts = [torch.empty_like(dense) for _ in range(world)] if rank == dst else None
    t1 = tick()
    dist.gather(dense, parts, dst=dst)
    t2 = tick()
    small = {k: result[k] for k in ("chain", "variable_inclusion", "vi_counts", "history", "counters")}
    if "seed" in result:
        small["seed"] = result["seed"]
    gathered = [None] * world if rank == dst else None
    dist.gather_object(small, gathered, dst=dst)
    t3 = tick()
    if timings is not None:
        timings.update(h2d_ms=(t1 - t0) * 1e3, collective_ms=(t2 - t1) * 1e3, object_ms=(t3 - t2) * 1e3, d2h_ms=0.0,
                       dense_bytes=int(dense.numel() * dense.element_size()))
    if rank != dst:
        return None
    out = []
    hosts = [parts[r].cpu().numpy() for r in range(world)]
    if timings is not None:
        timings["d2h_ms"] = (tick() - t3) * 1e3
    for r in range(world):
        d = hosts[r]
        item = dict(gathered[r])
        item["mu"] = d[:, :-1] if result["mu"] is not None else None
        item["sigma"] = d[:, -1]
        out.append(item)
    return out


def attach_history(op: BARTOp, chains: list[dict]) -> None:
    """Put the gathered per-chain histories where the reference's predictors look for them
    (``op.all_trees``, ``utils.py:124-127``)."""
    del op.all_trees[:]
    for c in chains:
        op.all_trees.append(c["history"])
