"""Device-memory holder for the HIP backend.

PyTorch-ROCm is used for exactly three things: allocating HBM buffers, naming the
HIP stream the kernels are enqueued on, and ``torch.distributed`` (RCCL) for the
end-of-run gather.  All compute goes through ``libpgbart_hip.so``.
"""

from __future__ import annotations

import numpy as np


class TorchHipMemory:
    """HBM buffers held as torch tensors on one GPU.  Raises if no GPU is visible."""

    def __init__(self, device: int | None = None):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError(
                "pymc_bart_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                "there is no CPU fallback."
            )
        self.torch = torch
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", int(device))
        torch.cuda.set_device(self.device)

    @property
    def stream_ptr(self) -> int:
        return int(self.torch.cuda.current_stream(self.device).cuda_stream)

    _queue_pad: dict = {}
    _queue_lock = __import__("threading").Lock()

    @classmethod
    def _prime_queues(cls, torch, device):
        """HIP multiplexes the streams of a process onto 4 hardware queues: a stream that is used for the first
        time gets a queue of its own while fewer than 4 exist and one of the least-referenced queues afterwards --
        and two BUSY chains on one queue run at half speed (four concurrent chains on one GPU: 3.9 M instead of
        5.7 M particle-steps/s; `tools/multichain_probe.py`, `profiles/r03_experiments.md`).  With 1-3 queues
        taken when the chains arrive (the legacy default stream alone is enough) the fourth chain lands on another
        chain's queue.  The first sampler of a process therefore makes the count even before its own stream is
        used: it touches the default stream and three padding streams (kept alive, idle), after which every run
        of four new chain streams is spread over the four queues.  The first padding stream doubles as the
        output stream of every sampler (`pgb_set_output_stream`), so that one takes no queue of its own.
        Best effort: streams the application itself has used are not known here.  Limit: torch hands its streams
        out round-robin from a pool of 32 per device and priority, so in a process that creates samplers for a long
        time the ~30th LIVE sampler shares its hipStream_t with the padding / output stream or with another live
        sampler.  Results stay correct (every entry point of the library is synchronous at return, and a step's
        export is ordered behind its slots on whatever stream it is given); what stops holding is the balancing
        over the four hardware queues and the early export past the idle slots."""
        key = str(device)
        with cls._queue_lock:  # (chains.sample_chains builds its samplers from several threads at once)
            return cls._prime_locked(torch, device, key)

    @classmethod
    def _prime_locked(cls, torch, device, key):
        if key not in cls._queue_pad:
            pads = []
            torch.zeros(1, device=device)
            for _ in range(3):
                st = torch.cuda.Stream(device)
                with torch.cuda.stream(st):
                    torch.zeros(1, device=device)
                pads.append(st)
            torch.cuda.synchronize(device)
            cls._queue_pad[key] = pads
        return cls._queue_pad[key]

    def sampler_stream(self):
        """The HIP stream a new sampler enqueues its slots on: the caller's current stream -- unless that is the
        legacy default stream, which synchronises with every other stream of the device: then a stream of the
        sampler's own, after :meth:`_prime_queues`.  Every entry point of the library is synchronous at return,
        so the caller's own stream ordering is not involved.  Returns the torch stream (keep it alive)."""
        self._prime_queues(self.torch, self.device)
        cur = self.torch.cuda.current_stream(self.device)
        if int(cur.cuda_stream) == 0:
            return self.torch.cuda.Stream(self.device)
        return cur

    def output_stream(self):
        """The stream on which the results of `pgb_step_host` leave the device (shared by all samplers)."""
        return self._prime_queues(self.torch, self.device)[0]

    def from_host(self, arr: np.ndarray):
        t = self.torch.from_numpy(np.ascontiguousarray(arr))
        return t.to(self.device, non_blocking=False)

    def is_resident(self, obj) -> bool:
        """True for a float64 matrix that already lives in this GPU's HBM (a tensor from :meth:`from_host`)."""
        return isinstance(obj, self.torch.Tensor) and obj.is_cuda and obj.dtype == self.torch.float64

    def empty(self, shape, dtype=np.float64):
        tdt = {np.float64: self.torch.float64, np.int32: self.torch.int32}[np.dtype(dtype).type]
        return self.torch.empty(shape, dtype=tdt, device=self.device)

    def host_result(self, n: int) -> np.ndarray:
        """A float64 host array the GPU can DMA into directly: page-locked memory from torch's
        caching host allocator (recycled once the array is dropped, so steady-state cost is a free-list
        pop), handed out as a plain ndarray that owns a reference to its storage."""
        return self.torch.empty((int(n),), dtype=self.torch.float64, pin_memory=True).numpy()

    @staticmethod
    def ptr(buf) -> int:
        return int(buf.data_ptr())

    def to_host(self, buf) -> np.ndarray:
        return buf.cpu().numpy()

    def synchronize(self) -> None:
        self.torch.cuda.synchronize(self.device)
