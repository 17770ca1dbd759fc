"""`launch` (detectron2/engine/launch.py:24-94) with two worker processes over gloo on the CPU: ranks, the local
process group, a collective, argument passing; and the single-process shortcut."""
import os

import torch

from detectron2_centernet_amd.engine import launch
from detectron2_centernet_amd.utils import comm


def _worker(outdir, scale):
    import torch.distributed as dist

    t = torch.tensor([float(comm.get_rank() + 1) * scale])
    dist.all_reduce(t)
    comm.synchronize()
    with open(os.path.join(outdir, f"rank{comm.get_rank()}.txt"), "w") as f:
        f.write(f"{comm.get_world_size()} {comm.get_rank()} {comm.get_local_rank()} {comm.get_local_size()} "
                f"{int(comm.is_main_process())} {t.item()}")


def test_launch_two_processes_gloo(tmp_path):
    launch(_worker, 2, num_machines=1, machine_rank=0, dist_url="auto", args=(str(tmp_path), 2.0), backend="gloo")
    got = [open(tmp_path / f"rank{r}.txt").read().split() for r in range(2)]
    assert got[0] == ["2", "0", "0", "2", "1", "6.0"]
    assert got[1] == ["2", "1", "1", "2", "0", "6.0"]


def test_launch_single_process_calls_directly(tmp_path):
    seen = []
    launch(lambda a, b: seen.append((a, b, comm.get_world_size(), comm.is_main_process())), 1, args=(3, 4))
    assert seen == [(3, 4, 1, True)]
