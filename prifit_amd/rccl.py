"""Native RCCL communicator for the gradient exchange (the C-ABI path of SURVEY.md 8b: `prifit_allreduce_flat`).

    comm = NativeComm.from_process_group()     # unique id from rank 0 through the initialised torch.distributed group
    comm.allreduce_(flat_bucket)               # ONE sum all-reduce, in place, on PyTorch's current stream

`FlatGradBucket(model, native=True)` (or PRIFIT_NATIVE_RCCL=1) uses it instead of `torch.distributed.all_reduce`; the
process group is then only the side channel that carries the 128-byte unique id and the parameter broadcast.  The
default stays `torch.distributed` (backend "nccl" IS RCCL on ROCm): same collective, one code path fewer to set up.
"""
import atexit
import ctypes
import weakref

import torch
import torch.distributed as dist

from ._lib import call, cur_stream, dll, ptr


class NativeComm:
    def __init__(self, nranks, rank, unique_id: bytes):
        n = dll().prifit_comm_unique_id_bytes()
        if len(unique_id) != n:
            raise ValueError("unique id must be %d bytes" % n)
        self.nranks, self.rank = nranks, rank
        self._comm = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(unique_id, n)
        # torch.distributed's nccl backend has its RCCL loaded in this process; the library must bind THAT copy (a second
        # RCCL instance in one process is undefined): prifit_comm_in_process() says whether it did
        if dist.is_initialized() and dist.get_backend() == "nccl" and not dll().prifit_comm_in_process():
            raise RuntimeError("libprifit_hip.so did not find the RCCL already loaded by torch.distributed (nccl backend) and "
                               "would load a second copy; use FlatGradBucket(native=False)")
        call("prifit_comm_init", ctypes.byref(self._comm), nranks, rank, buf)
        _live.add(self)

    @staticmethod
    def new_unique_id() -> bytes:
        n = dll().prifit_comm_unique_id_bytes()
        buf = ctypes.create_string_buffer(n)
        call("prifit_comm_unique_id", buf)
        return buf.raw

    @classmethod
    def from_process_group(cls, group=None):
        """Rank 0 draws the id, the initialised torch.distributed group (any backend) carries it to the others."""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.new_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(world, rank, box[0])

    def allreduce_(self, flat):
        if not (flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()):
            raise ValueError("flat fp32 device buffer expected")
        call("prifit_allreduce_flat", ptr(flat), ctypes.c_longlong(flat.numel()), self._comm, cur_stream())
        return flat

    def destroy(self):
        if self._comm:
            comm, self._comm = self._comm, ctypes.c_void_p()
            call("prifit_comm_destroy", comm)

    def __del__(self):
        try:
            self.destroy()
        except Exception:   # interpreter shutdown / the library already unloaded
            pass


_live = weakref.WeakSet()


@atexit.register
def _destroy_all():
    for c in list(_live):
        try:
            c.destroy()
        except Exception:
            pass
