"""Child process of tests/test_rccl_gpu.py: one rank of a torch.distributed job on backend "nccl" (= RCCL on
ROCm).  Runs a short chain on cuda:LOCAL_RANK through the HIP library, pushes its results through the
collective branch of ``gather_chains`` and prints one JSON line.  Started by torch.distributed.run, never
exec'ed from a process that has touched the GPU."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world, local = (int(os.environ[k]) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    from pymc_bart_amd.chains import gather_chains, sample_chain
    from pymc_bart_amd.pgbart import BARTOp
    from pymc_bart_amd.sampler import default_backend

    be = default_backend(local)
    assert be.lib.backend_name == "hip-gfx950"
    rng = np.random.default_rng(11)
    X = rng.normal(size=(2000, 3))
    Y = X[:, 0] - 2 * X[:, 2] + rng.normal(0, 0.3, 2000)
    res = sample_chain(BARTOp(X, Y, m=6), tune=10, draws=8, random_seed=3415, chain=rank, backend=be)
    got = gather_chains(res, dist, dst=0, force_collective=True)
    # keep_draws=False: only sigma travels densely (the dense tensor has a zero-width mu block)
    res_nd = sample_chain(BARTOp(X, Y, m=6), tune=10, draws=8, random_seed=3415, chain=rank, backend=be,
                          keep_draws=False)
    got_nd = gather_chains(res_nd, dist, dst=0, force_collective=True)
    # a step method unpickled in this rank (how PyMC hands it to a worker) lands on THIS rank's GPU and resumes
    # the chain bit for bit
    import pickle

    twin = pickle.loads(pickle.dumps(res["step"]))
    a = res["step"].astep(None, {"sigma": 0.5})[0]
    b = twin.astep(None, {"sigma": 0.5})[0]
    ones = torch.ones(4, device="cuda")
    dist.all_reduce(ones)
    out = {"backend": dist.get_backend(), "world": world, "allreduce": float(ones[0].item()),
           "unpickled_device": twin._device_index, "local_rank": local, "current_device": torch.cuda.current_device(),
           "unpickled_resumes": bool(np.array_equal(a, b))}
    if rank == 0:
        out["chains"] = len(got)
        out["mu_equal"] = bool(np.array_equal(got[0]["mu"], res["mu"]))
        out["sigma_equal"] = bool(np.array_equal(got[0]["sigma"], res["sigma"]))
        out["vi_equal"] = bool(np.array_equal(got[0]["vi_counts"], res["vi_counts"]))
        out["n_batches"] = len(got[0]["history"][1])
        out["nodraws_mu_none"] = got_nd[0]["mu"] is None
        out["nodraws_sigma_equal"] = bool(np.array_equal(got_nd[0]["sigma"], res_nd["sigma"])
                                          and np.array_equal(res_nd["sigma"], res["sigma"]))
        out["nodraws_vi_equal"] = bool(np.array_equal(got_nd[0]["vi_counts"], res_nd["vi_counts"]))
        out["nodraws_n_batches"] = len(got_nd[0]["history"][1])
        print("RCCL_CHILD " + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
