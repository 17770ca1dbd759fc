"""Probe (round 4): is a cross-stream dependency taken AFTER a HIP-graph replay honoured?

The data-parallel graph path replays forward + backward as one graph and then hands the flat gradient buffer to the
collective, which (gloo: its copy stream; RCCL: its communication stream) waits on an EVENT recorded on the launch stream
behind the graph.  Here the "collective" is a device-to-host copy on a side stream: after it has finished, the device is
synchronised and the host copy is compared with what the buffer finally holds.  Any difference = the event fired before
the graph's last kernels."""
import sys
import torch

sys.path.insert(0, ".")
import bench
from detectron2_centernet_amd.engine.bench_train import synthetic_batch
from detectron2_centernet_amd.engine.train_loop import SimpleTrainer


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    dev = torch.device("cuda:0")
    model, cfg = bench.build_model(prec, dev, calibrate=False)
    cfg.SOLVER.IMS_PER_BATCH = 16
    tr = SimpleTrainer(model, None, cfg)
    batch = synthetic_batch(16, 512, 0, dev)
    for _ in range(2):
        tr.run_step_tensors(*batch)
    key = tuple((tuple(t.shape), t.dtype) for t in batch)
    g = tr._graphs[key]
    tr._capture(g, *batch, with_step=False)
    assert g["graph"] is not None, g.get("failed")
    opt = tr.optimizer
    side = torch.cuda.Stream()
    host = torch.empty(opt.flat_grad.shape, dtype=torch.float32, pin_memory=True)
    bad = 0
    for it in range(steps):
        g["graph"].replay()
        ev = torch.cuda.Event()
        ev.record()                                   # on the launch stream, behind the graph
        side.wait_event(ev)
        with torch.cuda.stream(side):
            host.copy_(opt.flat_grad, non_blocking=True)
        side.synchronize()
        early = host.clone()
        torch.cuda.synchronize()
        final = opt.flat_grad.cpu()
        diff = (early - final).abs().max().item()
        nz = int((early != final).sum())
        if nz:
            bad += 1
            idx = (early != final).nonzero().flatten()
            print(f"step {it}: {nz} elements differ (max {diff:.3e}), first at {int(idx[0])}, last at {int(idx[-1])} of {final.numel()}", flush=True)
        opt.step()
        for p in opt.params:
            torch.autograd.graph.increment_version(p)
    print(f"{prec}: {bad} of {steps} replays were followed by an early event; losses {[float(v) for v in g['losses'].values()]}")


main()
