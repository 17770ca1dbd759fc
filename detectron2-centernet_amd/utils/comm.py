"""Process-group helpers with the names the reference's code uses (detectron2/utils/comm.py:1-263, the subset the
CenterNet path touches): world size / rank queries that work without an initialised group, a barrier, and the
node-local group that `launch` sets up."""
import torch
import torch.distributed as dist

_LOCAL_PROCESS_GROUP = None


def _on():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if _on() else 1


def get_rank():
    return dist.get_rank() if _on() else 0


def get_local_rank():
    if not _on():
        return 0
    assert _LOCAL_PROCESS_GROUP is not None
    return dist.get_rank(group=_LOCAL_PROCESS_GROUP)


def get_local_size():
    if not _on():
        return 1
    return dist.get_world_size(group=_LOCAL_PROCESS_GROUP)


def is_main_process():
    return get_rank() == 0


def synchronize():
    """barrier among all ranks (no-op for a single process)"""
    if get_world_size() == 1:
        return
    if dist.get_backend() == "nccl" and torch.cuda.is_available():
        dist.barrier(device_ids=[torch.cuda.current_device()])
    else:
        dist.barrier()


def gather(data, dst=0):
    """a picklable object from every rank -> list of them on `dst` ([] elsewhere); [data] for a single process
    (detectron2/utils/comm.py:166-205)"""
    if get_world_size() == 1:
        return [data]
    out = [None] * get_world_size() if get_rank() == dst else None
    dist.gather_object(data, out, dst=dst)
    return out if get_rank() == dst else []
