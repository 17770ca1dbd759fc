"""World-size-2 CPU test (gloo) of the data-parallel gradient path: flat gradient buffer in backward order,
bucket construction, all-reduce launched from autograd hooks, unused-parameter buckets, parameter broadcast."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from detectron2_centernet_amd.engine.reducer import BucketedReducer
        from detectron2_centernet_amd.solver.build import FlatSGD

        torch.manual_seed(100 + rank)  # different initial weights per rank: broadcast must equalise them
        net = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.ReLU(), torch.nn.Linear(300, 300), torch.nn.ReLU(),
                                  torch.nn.Linear(300, 7))
        unused = torch.nn.Parameter(torch.randn(50))       # never touched by forward (like DLA's outer `project`)
        groups = [(p, 1.0, 0.0) for p in list(net.parameters()) + [unused]]
        opt = FlatSGD(groups, base_lr=0.1, device=torch.device("cpu"))
        red = BucketedReducer(opt, bucket_bytes=64 * 1024)
        assert len(red.buckets) >= 3 and red.world == world
        assert sum(b[2] for b in red.buckets) == len(groups)
        red.broadcast_parameters()
        ref = [torch.zeros_like(opt.flat_param) for _ in range(world)]
        dist.all_gather(ref, opt.flat_param)
        assert torch.equal(ref[0], ref[1]), "parameters differ after broadcast"
        # the flat layout is reverse registration order and parameters are views into it
        assert opt.params[0] is unused and net[0].weight.data_ptr() >= opt.flat_param.data_ptr()
        torch.manual_seed(7 + rank)
        x = torch.randn(16, 40)
        opt.zero_grad()
        red.prepare()
        net(x).square().mean().backward()
        local = opt.flat_grad.clone() if False else None
        red.finish()
        # reference: sum of the per-rank gradients computed without the reducer
        net2 = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.ReLU(), torch.nn.Linear(300, 300), torch.nn.ReLU(),
                                   torch.nn.Linear(300, 7))
        net2.load_state_dict(net.state_dict())
        net2(x).square().mean().backward()
        mine = torch.cat([p.grad.reshape(-1) for p in reversed(list(net2.parameters()))])
        both = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        want = both[0] + both[1]
        got = opt.flat_grad[unused.numel():]
        assert torch.allclose(got, want, atol=1e-6), (got - want).abs().max()
        assert opt.flat_grad[:unused.numel()].abs().max() == 0
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_bucketed_reducer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


def test_lr_schedule_and_param_groups(tmp_path):
    from detectron2_centernet_amd.solver import param_groups, warmup_multistep_factor

    assert warmup_multistep_factor(0, (10, 20), 0.1, 0.001, 5) == pytest.approx(0.001)
    assert warmup_multistep_factor(5, (10, 20), 0.1, 0.001, 5) == 1.0
    assert warmup_multistep_factor(10, (10, 20), 0.1, 0.001, 5) == pytest.approx(0.1)
    assert warmup_multistep_factor(25, (10, 20), 0.1, 0.001, 5) == pytest.approx(0.01)

    from detectron2_centernet_amd.config import get_cfg
    cfg = get_cfg()
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4))
    g = param_groups(cfg, net)
    wd = {id(p): w for p, _, w in g}
    assert wd[id(net[0].weight)] == cfg.SOLVER.WEIGHT_DECAY and wd[id(net[0].bias)] == cfg.SOLVER.WEIGHT_DECAY_BIAS
    assert wd[id(net[1].weight)] == cfg.SOLVER.WEIGHT_DECAY_NORM == 0.0
