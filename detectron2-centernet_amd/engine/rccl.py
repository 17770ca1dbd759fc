"""RCCL communicator behind the C ABI (`ctdet_comm_*`, `ctdet_allreduce_bucket`): the gradient exchange without
torch.distributed's collectives.  The 128-byte unique id travels over whatever process group already exists (it is only
used as a bootstrap channel here: one broadcast of 128 bytes); a single process makes a world-size-1 communicator."""
import ctypes as C

import torch

from .. import _lib


class RcclComm:
    def __init__(self, rank=0, world=1, bootstrap_group=None):
        import torch.distributed as dist

        lib = _lib.lib()
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_char * 128)()
            _lib.check(lib.ctdet_comm_unique_id(C.cast(buf, C.c_void_p)), "ctdet_comm_unique_id")
            uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if world > 1:
            dev = "cuda" if dist.get_backend(bootstrap_group) == "nccl" else "cpu"
            t = uid.to(dev)
            dist.broadcast(t, src=0, group=bootstrap_group)
            uid = t.cpu()
        raw = (C.c_char * 128).from_buffer_copy(bytes(uid.tolist()))
        handle = C.c_void_p()
        _lib.check(lib.ctdet_comm_init(C.cast(raw, C.c_void_p), rank, world, C.byref(handle)), "ctdet_comm_init")
        self._h, self.rank, self.world = handle, rank, world

    def all_reduce_(self, flat_f32, stream=None):
        """in-place SUM all-reduce of a contiguous f32 device tensor on `stream` (default: the current stream)"""
        assert flat_f32.is_cuda and flat_f32.dtype == torch.float32 and flat_f32.is_contiguous()
        s = (stream or torch.cuda.current_stream()).cuda_stream
        _lib.check(_lib.lib().ctdet_allreduce_bucket(self._h, C.c_void_p(flat_f32.data_ptr()), flat_f32.numel(), C.c_void_p(s)),
                   "ctdet_allreduce_bucket")

    def broadcast_(self, flat_f32, root=0, stream=None):
        assert flat_f32.is_cuda and flat_f32.dtype == torch.float32 and flat_f32.is_contiguous()
        s = (stream or torch.cuda.current_stream()).cuda_stream
        _lib.check(_lib.lib().ctdet_bcast(self._h, C.c_void_p(flat_f32.data_ptr()), flat_f32.numel(), root, C.c_void_p(s)),
                   "ctdet_bcast")

    def close(self):
        if self._h:
            _lib.check(_lib.lib().ctdet_comm_destroy(self._h), "ctdet_comm_destroy")
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
