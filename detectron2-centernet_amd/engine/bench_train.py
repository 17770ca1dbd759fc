"""Training leg of bench.py: BASELINE.json configs[2]/[3] -- DLA-34 CenterNet, 512x512, batch 16 per GPU, synthetic
COCO-shaped targets; one step = device target generation + forward + losses + backward + (bucketed RCCL all-reduce
when WORLD_SIZE > 1) + SGD update + LR schedule."""
import time

import torch

from ..data.catalog import synthetic_sample
from .train_loop import SimpleTrainer


_MODE = {
    0: ("f16", "f16 activations / f32 master weights"),
    1: ("f32", "f32 activations, gradients and statistics, contractions as f32 FMA chains (the reference's own arithmetic)"),
    3: ("f16x3", "f32 activations, gradients and statistics; every contraction (forward, input and weight gradients, DCNv2 column "
                 "GEMMs) as hi*hi + lo*hi + hi*lo on the f16 matrix pipe with f32 accumulation (f32-grade results)"),
}


def synthetic_batch(B, size, rank, device, num_classes=80, max_boxes=32):
    imgs, boxes, classes, counts = [], torch.zeros(B, max_boxes, 4), torch.zeros(B, max_boxes, dtype=torch.int64), []
    for b in range(B):
        s = synthetic_sample(rank * 100000 + b, size=size, num_classes=num_classes, max_boxes=max_boxes)
        imgs.append(s["image"])
        n = s["boxes"].shape[0]
        boxes[b, :n], classes[b, :n] = s["boxes"], s["classes"]
        counts.append(n)
    return (torch.stack(imgs).to(device), boxes.to(device), classes.to(device),
            torch.tensor(counts, dtype=torch.int32, device=device))


def run_train_bench(model, cfg, args, B, rank, world, device, dist):
    cfg.SOLVER.IMS_PER_BATCH = B * world
    trainer = SimpleTrainer(model, None, cfg)
    images, boxes, classes, counts = synthetic_batch(B, args.size, rank, device)
    import contextlib
    import os
    # CTDET_BENCH_STREAM=1 (measurement only): the steps run on a stream of their own instead of the default (null) stream
    side = torch.cuda.Stream() if os.environ.get("CTDET_BENCH_STREAM") == "1" else None
    if side is not None:
        side.wait_stream(torch.cuda.current_stream())
    with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
        for _ in range(args.warmup):
            trainer.run_step_tensors(images, boxes, classes, counts)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
        for _ in range(args.steps):
            trainer.run_step_tensors(images, boxes, classes, counts)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if getattr(trainer, "_steplog", None) is not None:      # CTDET_TRAIN_CHECK=2
        import sys
        print(f"[rank {rank}] step log [hm, wh, off, max|feat|, max|logit|, max tgt, min tgt, grad ok, reduced ok, param ok]\n" +
              "\n".join(f"  {i}: " + " ".join(f"{v:.6g}" for v in r) for i, r in enumerate(trainer.step_log().tolist())),
              file=sys.stderr, flush=True)
    try:
        losses = trainer.metrics()
    except FloatingPointError:
        _diagnose(trainer, (images, boxes, classes, counts), rank)
        raise
    comm_rec = allreduce_bandwidth(trainer, dist, device) if (dist is not None and world > 1) else None
    return _record(model, args, B, world, elapsed, losses, trainer, comm_rec)


def _diagnose(trainer, batch, rank):
    """a non-finite loss ends the bench: say where the non-finite values sit (stderr) before the exception goes up"""
    import sys
    torch.cuda.synchronize()
    opt = trainer.optimizer
    msg = [f"[rank {rank}] non-finite loss at iteration {trainer.iter}: "
           f"params finite={bool(torch.isfinite(opt.flat_param).all())} grads finite={bool(torch.isfinite(opt.flat_grad).all())} "
           f"momentum finite={bool(torch.isfinite(opt.flat_mom).all())}"]
    for key, g in trainer._graphs.items():
        if g.get("inputs") is not None:
            same = [bool(torch.equal(a, b)) for a, b in zip(g["inputs"], batch)]
            msg.append(f"  captured step's static inputs equal the batch (images, boxes, classes, counts): {same}")
    bad = [n for n, b in trainer.model.named_buffers() if b.dtype.is_floating_point and not bool(torch.isfinite(b).all())]
    msg.append(f"  non-finite buffers: {bad[:8]}")
    names = {id(p): n for n, p in trainer.model.named_parameters()}
    badp = [names.get(id(p), "?") for p in opt.params if not bool(torch.isfinite(p).all())]
    msg.append(f"  non-finite parameters: {badp[:8]}")
    if getattr(trainer, "_steplog", None) is not None:
        msg.append("  step log [hm, wh, off, max|feat|, max|logit|, max tgt, min tgt, grad ok, reduced ok, param ok]:")
        for i, r in enumerate(trainer.step_log().tolist()):
            msg.append(f"    {i}: " + " ".join(f"{v:.6g}" for v in r))
    with torch.no_grad():
        t = trainer.model.train_batch_tensor(*batch)
    msg.append(f"  an eager forward on the batch now gives {dict((k, float(v)) for k, v in t.items())}")
    print("\n".join(msg), file=sys.stderr, flush=True)


def allreduce_bandwidth(trainer, dist, device, reps=10):
    """the exchange of one step by itself: the bucketed all-reduce of the flat gradient buffer (78.7 MB for DLA-34), timed with
    HIP events on the launch stream between two barriers, max over ranks.  bus bandwidth = 2 (N - 1) / N x bytes / time: what a
    ring moves per link -- to be read against 153 GB/s per xGMI link (7 per GPU on a fully connected node)."""
    red = trainer.reducer
    nbytes = trainer.optimizer.flat_grad.numel() * 4
    for _ in range(2):
        red.reduce_all()
    torch.cuda.synchronize()
    dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        red.reduce_all()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    t = torch.tensor([ms], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ms = float(t.item())
    n = red.world
    return {"bytes": nbytes, "buckets": len(red.buckets), "ms": ms, "algbw_GBs": nbytes / ms / 1e6,
            "busbw_GBs": 2.0 * (n - 1) / n * nbytes / ms / 1e6, "backend": dist.get_backend(),
            "note": "all buckets back to back, nothing overlapped; per-link peak 153 GB/s (xGMI)"}


def _record(model, args, B, world, elapsed, losses, trainer, comm_rec):
    return {
        "metric": "images/sec at 512x512 (train bs=16/GPU)" if (B == 16 and args.size == 512) else
                  f"images/sec at {args.size}x{args.size} (train bs={B}/GPU)",
        "value": world * B * args.steps / elapsed,
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": _MODE[model._ctx.compute][0],
        "data": "synthetic",
        "config": {"workload": f"{'DLA-34' if model.backbone_type == 'dla34' else 'ResNet'} CenterNet train step (targets+fwd+loss+bwd+allreduce+SGD), {B}x3x{args.size}x"
                               f"{args.size} per GPU, 80 classes, " + _MODE[model._ctx.compute][1],
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                   "final_losses": losses, "graph_state": trainer.graph_state,
                   "graph_nodes": next((g.get("nodes") for g in trainer._graphs.values() if g["graph"] is not None), None),
                   "allreduce": comm_rec},
    }
