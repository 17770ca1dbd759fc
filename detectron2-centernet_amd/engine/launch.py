"""`launch` -- one process per GPU of this node (detectron2/engine/launch.py:24-94).

Same signature and behaviour as the reference: spawns `num_gpus_per_machine` workers, each joins the process group
(backend "nccl" = RCCL over xGMI on ROCm), binds to its GPU, builds the node-local group, then calls
`main_func(*args)`.  `dist_url="auto"` picks a free port on 127.0.0.1 (the container's hostname may not resolve).
Differences: the backend is a parameter (the CPU tests run the same code over gloo), and
HSA_ENABLE_IPC_MODE_LEGACY=0 is exported to the workers (the host driver only supports dmabuf IPC).
"""
import logging
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ..utils import comm

__all__ = ["launch"]


def _find_free_port():
    sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    sock.bind(("127.0.0.1", 0))      # the OS picks a free port
    port = sock.getsockname()[1]
    sock.close()
    return port


def launch(main_func, num_gpus_per_machine, num_machines=1, machine_rank=0, dist_url=None, args=(), backend="nccl"):
    world_size = num_machines * num_gpus_per_machine
    if world_size <= 1:
        main_func(*args)
        return
    if dist_url == "auto":
        assert num_machines == 1, "dist_url=auto not supported in multi-machine jobs."
        dist_url = f"tcp://127.0.0.1:{_find_free_port()}"
    if num_machines > 1 and dist_url.startswith("file://"):
        logging.getLogger(__name__).warning("file:// is not a reliable init_method in multi-machine jobs. Prefer tcp://")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mp.spawn(_distributed_worker, nprocs=num_gpus_per_machine,
             args=(main_func, world_size, num_gpus_per_machine, machine_rank, dist_url, args, backend), daemon=False)


def _distributed_worker(local_rank, main_func, world_size, num_gpus_per_machine, machine_rank, dist_url, args, backend):
    if backend == "nccl":
        assert torch.cuda.is_available(), "cuda is not available. Please check your installation."
        assert num_gpus_per_machine <= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)   # before the group exists: RCCL binds its communicator to the current device
    global_rank = machine_rank * num_gpus_per_machine + local_rank
    try:
        kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
        dist.init_process_group(backend=backend, init_method=dist_url, world_size=world_size, rank=global_rank, **kw)
    except Exception:
        logging.getLogger(__name__).error("Process group URL: {}".format(dist_url))
        raise
    comm.synchronize()   # prevents a possible timeout right after init_process_group (launch.py:76-78)
    # the local process group: ranks within the same machine
    assert comm._LOCAL_PROCESS_GROUP is None
    num_machines = world_size // num_gpus_per_machine
    for i in range(num_machines):
        ranks_on_i = list(range(i * num_gpus_per_machine, (i + 1) * num_gpus_per_machine))
        pg = dist.new_group(ranks_on_i)
        if i == machine_rank:
            comm._LOCAL_PROCESS_GROUP = pg
    try:
        main_func(*args)
    finally:
        dist.destroy_process_group()
