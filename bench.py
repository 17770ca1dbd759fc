#!/usr/bin/env python
"""Benchmark of the CenterNet DLA-34 hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--task infer|train] [--batch B]

N = 1 workload (BASELINE.json configs[1]): DLA-34 CenterNet inference, batch 64, 512x512 synthetic uint8
images resident in HBM; one step = the whole eval forward of the meta-architecture (preprocess, backbone, heads,
sigmoid+clamp, peak-NMS, top-K decode, threshold + detector_postprocess, Instances).  For N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank runs the same per-GPU batch on its own shard of images
(weak scaling, no data-path collective for inference); the timed region is bracketed by barrier +
torch.cuda.synchronize and the maximum over ranks is reported.

Prints ONE JSON line (rank 0) with the throughput, the roofline of the dominant kernel (measured live with
HIP events around every launch of that kernel in an instrumented eager pass) and the CPU baseline (the
oracle's port of the reference arithmetic timed on this host's cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F16_PEAK_TFLOPS = 2500.0   # dense f16/bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
FP32_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0

BASE_YAML = """
MODEL:
  META_ARCHITECTURE: "CenterNet"
  BACKBONE:
    NAME: "build_dla34_backbone"
  PIXEL_MEAN: [0.408, 0.447, 0.470]
  PIXEL_STD: [0.289, 0.274, 0.278]
VERSION: 2
"""
DLA_YAML = """
_BASE_: "./Base-CenterNet.yaml"
MODEL:
  CENTERNET:
    FOCAL_LOSS_ALPHA: [1]
DATASETS:
  TRAIN: ("bulb_train",)
  TEST: ("bulb_val",)
INPUT:
  FORMAT: "RGB"
  MIN_SIZE_TRAIN: (640, 672, 704, 736, 768, 800)
SOLVER:
  IMS_PER_BATCH: 2
  BASE_LR: 2.5e-4
  STEPS: (159000, 212000)
  MAX_ITER: 265000
  CHECKPOINT_PERIOD: 10600
OUTPUT_DIR: "./output/centernet-bulb-aug"
VERSION: 2
"""


def build_model(precision, device, seed=0):
    import tempfile

    from detectron2_centernet_amd.config import get_cfg
    from detectron2_centernet_amd.data.catalog import register_synthetic
    from detectron2_centernet_amd.modeling import build_model as _build

    d = tempfile.mkdtemp(prefix="ctdet_cfg_")
    with open(os.path.join(d, "Base-CenterNet.yaml"), "w") as f:
        f.write(BASE_YAML)
    with open(os.path.join(d, "ctdet_dla_34_1x.yaml"), "w") as f:
        f.write(DLA_YAML)
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(d, "ctdet_dla_34_1x.yaml"))
    cfg.MODEL.CENTERNET.HIP_PRECISION = precision
    cfg.MODEL.DEVICE = str(device)
    register_synthetic("bulb_train", num_classes=80)   # COCO-shaped: 80 classes
    torch.manual_seed(seed)
    model = _build(cfg)
    # random-init weights of the reference architecture; DCN offset convs get small non-zero weights so the
    # deformable gathers are data dependent (offsets ~ N(0, ~1 px)), not the zero-init regular grid
    g = torch.Generator().manual_seed(seed + 1)
    for name, m in model.named_modules():
        if name.endswith("conv_offset_mask"):
            m.weight.data.copy_((torch.randn(m.weight.shape, generator=g) * (0.5 / (m.weight.shape[1] * 9) ** 0.5)).to(device))
            m.bias.data.copy_((torch.randn(m.bias.shape, generator=g) * 0.5).to(device))
    # the reference initialises the wh head to ~0 (centernet.py fill_fc_weights): every decoded box would have zero size and be
    # dropped as empty, and the step would end with no detections to post-process.  A constant wh bias gives 12-pixel boxes, so
    # each image yields its full 100 detections (threshold 0.05 < the ~0.1 scores of the -2.19 hm bias).
    model.wh[-1].bias.data.fill_(3.0)
    return model, cfg


def synthetic_images(B, size, rank, device):
    g = torch.Generator().manual_seed(1234 + rank)
    return torch.randint(0, 256, (B, 3, size, size), generator=g, dtype=torch.uint8).to(device)


def cpu_baseline(model, cfg, size, n_images=2, reps=3):
    """the oracle (CPU port of the reference arithmetic: torch-CPU convs + the DCNv2 restatement + decode) on a
    bounded sample: BASELINE.json configs[0] = 2 synthetic 512x512 images, forward + decode."""
    from oracle import model_ref as MR

    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(99)
    imgs = [torch.randint(0, 256, (3, size, size), generator=g, dtype=torch.uint8) for _ in range(n_images)]
    times = []
    with torch.no_grad():
        for _ in range(reps):
            t0 = time.perf_counter()
            MR.centernet_inference(sd, imgs, cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD)
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": n_images / med, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_images} synthetic {size}x{size} images, forward+decode, median of {reps} (first run included)"}


def roofline_pass(model, images, passes=2):
    """instrumented eager passes (no graph): every conv-shaped launch is issued ops.PROFILE_REP times back to
    back between one HIP event pair on the launch stream; returns per-kernel-instantiation aggregates."""
    from detectron2_centernet_amd import ops
    from detectron2_centernet_amd.modeling.meta_arch.centernet import _EvalEngine

    B, _, H, W = images.shape
    eng = _EvalEngine(model, B, H, W, H, W, images.dtype, use_graph=False)
    eng.images.copy_(images)
    agg = {}
    for p in range(passes):
        ops.PROFILE.clear()
        ops.PROFILE_ON = True
        with torch.no_grad():
            eng()
        ops.PROFILE_ON = False
        torch.cuda.synchronize()
        for name, flops, e0, e1, _bytes, _info in ops.PROFILE:
            a = agg.setdefault(name, {"ms": 0.0, "flops": 0.0, "launches": 0})
            a["ms"] += e0.elapsed_time(e1) / ops.PROFILE_REP
            a["flops"] += flops
            a["launches"] += 1
    for a in agg.values():
        a["ms"] /= passes
        a["flops"] /= passes
        a["launches"] //= passes
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--task", default="infer", choices=["infer", "train"])
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default 64 infer / 16 train)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--precision", default="f16", choices=["f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    # CTDET_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share devices, the
    # timing reduction goes over gloo); the driver's multi-GPU runs use the default: one rank per GPU, RCCL
    backend = os.environ.get("CTDET_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    B = args.batch or (64 if args.task == "infer" else 16)
    model, cfg = build_model(args.precision, device)

    if args.task == "train":
        from detectron2_centernet_amd.engine.bench_train import run_train_bench
        result = run_train_bench(model, cfg, args, B, rank, world, device, dist)
    else:
        model.eval()
        images = synthetic_images(B, args.size, rank, device)

        # serving loop with one step in flight: the next batch is enqueued before the host reads back the previous
        # batch's detection counts and builds its Instances; every step's full result is materialised inside the
        # timed region (the last one after the loop)
        with torch.no_grad():
            for _ in range(args.warmup):
                model.infer_batch_tensor(images)
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pending = None
            for _ in range(args.steps):
                h = model.infer_batch_tensor_async(images)
                if pending is not None:
                    out = pending.result()
                pending = h
            out = pending.result()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        assert len(out) == B
        result = {
            "metric": "images/sec at 512x512 (infer bs=64)" if (B == 64 and args.size == 512) else
                      f"images/sec at {args.size}x{args.size} (infer bs={B})",
            "value": world * B * args.steps / elapsed,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "config": {"workload": f"DLA-34 CenterNet eval forward+decode, {B}x3x{args.size}x{args.size} uint8 per GPU, "
                                   "80 classes, K=100, random-init weights, DCN offsets ~N(0,1px), "
                                   f"{sum(len(o['instances']) for o in out) / max(1, len(out)):.0f} detections per image post-processed",
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"replicas x{world}"},
        }
        if rank == 0 and not args.no_roofline:
            agg = roofline_pass(model, images)
            name, a = max(agg.items(), key=lambda kv: kv[1]["ms"])
            peak = MFMA_F16_PEAK_TFLOPS if args.precision == "f16" else FP32_PEAK_TFLOPS
            ach = a["flops"] / (a["ms"] * 1e-3) / 1e12
            total_ms = sum(v["ms"] for v in agg.values())
            total_fl = sum(v["flops"] for v in agg.values())
            # HBM bytes per launch of that kernel: not measurable from inside this process; taken from the committed
            # PMC summary (profiles/r01_hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over the same
            # forward, corrected as MI355X_MICROARCH.md prescribes) when it covers the kernel, else null
            traffic = None
            tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_hbm_traffic.json")
            if B == 64 and args.size == 512 and os.path.exists(tpath):
                with open(tpath) as f:
                    traffic = json.load(f)["bytes_per_launch"].get(name)
            result["roofline"] = {
                "bound": "mfma", "kernel": name, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "traffic": traffic, "launches_per_step": a["launches"], "kernel_ms_per_step": a["ms"],
                "all_conv_kernels": {"ms_per_step": total_ms, "achieved": total_fl / (total_ms * 1e-3) / 1e12,
                                     "frac": total_fl / (total_ms * 1e-3) / 1e12 / peak},
                "per_kernel": {k: {"ms": round(v["ms"], 4), "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                   "launches": v["launches"]} for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])},
            }
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(model, cfg, args.size)

    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
