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
