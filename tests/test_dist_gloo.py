"""One chain per rank with a single end-of-run gather, exercised with
world_size=2 on the gloo backend.  On the MI355X node the same code runs over RCCL ("nccl")."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _data():
    rng = np.random.default_rng(11)
    X = rng.normal(size=(200, 3))
    Y = X[:, 0] - 2 * X[:, 2] + rng.normal(0, 0.3, 200)
    return X, Y


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    from _oracle import oracle_backend
    from pymc_bart_amd.chains import gather_chains, sample_chain
    from pymc_bart_amd.pgbart import BARTOp

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, Y = _data()
    res = sample_chain(BARTOp(X, Y, m=6), tune=15, draws=10, random_seed=3415, chain=rank,
                       backend=oracle_backend())
    chains = gather_chains(res, dist, dst=0)
    if rank == 0:
        assert len(chains) == world
        np.savez(out_path, mu=np.stack([c["mu"] for c in chains]), sigma=np.stack([c["sigma"] for c in chains]),
                 vi=np.stack([c["vi_counts"] for c in chains]),
                 n_batches=np.array([len(c["history"][1]) for c in chains]),
                 chain_ids=np.array([c["chain"] for c in chains]))
    else:
        assert chains is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_chains_gather_matches_single_process(tmp_path, oracle):
    from pymc_bart_amd.chains import attach_history, gather_chains, sample_chain
    from pymc_bart_amd.pgbart import BARTOp
    from pymc_bart_amd.utils import _get_posterior_sampler, _sample_posterior

    out = str(tmp_path / "gathered.npz")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    X, Y = _data()
    ops = []
    for chain in range(2):
        op = BARTOp(X, Y, m=6)
        ref = sample_chain(op, tune=15, draws=10, random_seed=3415, chain=chain, backend=oracle)
        assert np.array_equal(got["mu"][chain], ref["mu"])
        assert np.array_equal(got["sigma"][chain], ref["sigma"])
        assert np.array_equal(got["vi"][chain], ref["vi_counts"])
        ops.append((op, ref))
    assert list(got["chain_ids"]) == [0, 1] and list(got["n_batches"]) == [10, 10]
    assert not np.array_equal(got["mu"][0], got["mu"][1])  # independent chains
    # the gathered histories feed the multi-chain predictor exactly like the reference's op.all_trees
    op = BARTOp(X, Y, m=6)
    op.n_outputs = 1
    single = [gather_chains(r)[0] for _, r in ops]
    attach_history(op, single)
    sampler = _get_posterior_sampler(op, backend=oracle)
    assert sampler.n_draws == 20
    pred = _sample_posterior(sampler, X[:7], np.random.default_rng(0), size=5)
    assert pred.shape == (5, 7, 1)


def _worker8(rank, world, port, out_path):
    """One rank of an 8-rank job: a chain with a history of realistic size (100 draws x 20 trees per draw)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pickle
    import time

    import torch.distributed as dist

    from _oracle import oracle_backend
    from pymc_bart_amd.chains import gather_chains, sample_chain
    from pymc_bart_amd.pgbart import BARTOp

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, Y = _data()
    res = sample_chain(BARTOp(X, Y, m=200), tune=5, draws=100, random_seed=3415, chain=rank,
                       backend=oracle_backend(), keep_draws=True)
    dist.barrier()
    t0 = time.perf_counter()
    chains = gather_chains(res, dist, dst=0)
    dt = time.perf_counter() - t0
    if rank == 0:
        assert len(chains) == world
        hist_bytes = len(pickle.dumps(chains[0]["history"]))
        np.savez(out_path, sigma=np.stack([c["sigma"] for c in chains]),
                 mu0=np.stack([c["mu"][0] for c in chains]),
                 n_batches=np.array([len(c["history"][1]) for c in chains]),
                 trees_per_batch=np.array([int(c["history"][1][0].n_trees) for c in chains]),
                 chain_ids=np.array([c["chain"] for c in chains]), gather_s=dt, hist_bytes=hist_bytes)
    else:
        assert chains is None
    dist.barrier()
    dist.destroy_process_group()


def test_eight_chains_gather_a_history_of_realistic_size(tmp_path, oracle):
    """8 ranks (the node's 8 GPUs), 100 draws x 20 trees of history per chain through the one collective of
    the design (reference: bart.py:133-135 per-chain histories back to the parent, utils.py:122-127)."""
    from pymc_bart_amd.chains import sample_chain
    from pymc_bart_amd.pgbart import BARTOp

    out = str(tmp_path / "gathered8.npz")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker8, args=(8, port, out), nprocs=8, join=True)
    got = np.load(out)
    assert list(got["chain_ids"]) == list(range(8))
    assert list(got["n_batches"]) == [100] * 8 and list(got["trees_per_batch"]) == [20] * 8
    assert len({got["sigma"][c].tobytes() for c in range(8)}) == 8  # eight independent chains
    X, Y = _data()
    ref = sample_chain(BARTOp(X, Y, m=200), tune=5, draws=100, random_seed=3415, chain=5, backend=oracle)
    assert np.array_equal(got["sigma"][5], ref["sigma"]) and np.array_equal(got["mu0"][5], ref["mu"][0])
    # gather_object of 8 x ~0.3 MB of packed history: well under a second even over loopback TCP
    assert float(got["gather_s"]) < 20.0, float(got["gather_s"])
    print(f"gather of 8 chains: {float(got['gather_s']) * 1e3:.1f} ms, history {int(got['hist_bytes'])} B per chain")
