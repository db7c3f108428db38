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
