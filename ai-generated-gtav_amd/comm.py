"""RCCL collectives through the C-ABI (include/gtav_amd.h gtav_comm_*): what a non-Python host binds for the multi-GPU path.
The Python harness itself can use either this or torch.distributed (generate.all_gather_latents, train.all_reduce_gradients);
`Comm.from_torch_distributed()` builds the communicator for the ranks of an initialised process group by broadcasting the 128-byte
RCCL id through it (the out-of-band channel of gtav_comm_unique_id)."""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as _lib


class Comm:
    def __init__(self, nranks: int, rank: int, unique_id: bytes):
        assert len(unique_id) == 128
        self.nranks, self.rank = nranks, rank
        self._h = C.c_void_p(None)
        buf = C.create_string_buffer(unique_id, 128)
        _lib.check(_lib.load().gtav_comm_init(C.byref(self._h), nranks, rank, C.cast(buf, C.c_void_p)))

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().gtav_comm_unique_id(C.cast(buf, C.c_void_p)))
        return buf.raw

    @classmethod
    def from_torch_distributed(cls) -> "Comm":
        import torch.distributed as dist
        ids = [cls.unique_id() if dist.get_rank() == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        return cls(dist.get_world_size(), dist.get_rank(), ids[0])

    def all_reduce_(self, t: torch.Tensor, average: bool = False) -> torch.Tensor:
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        with torch.cuda.device(t.device):
            _lib.check(_lib.load().gtav_comm_allreduce_f32(self._h, t.data_ptr(), t.numel(), 1 if average else 0, _lib.current_stream()))
        return t

    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        assert t.is_cuda and t.is_contiguous()
        out = torch.empty((self.nranks,) + tuple(t.shape), device=t.device, dtype=t.dtype)
        with torch.cuda.device(t.device):
            _lib.check(_lib.load().gtav_comm_allgather(self._h, t.data_ptr(), out.data_ptr(), t.numel() * t.element_size(), _lib.current_stream()))
        return out

    def close(self):
        if self._h:
            _lib.load().gtav_comm_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
