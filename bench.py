#!/usr/bin/env python
"""Benchmark of the CenterNet DLA-34 hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--task infer|train] [--batch B] [--precision f16x3|f16|f32]

N = 1 workload (BASELINE.json configs[1]): DLA-34 CenterNet inference, batch 64, 512x512 synthetic uint8
images resident in HBM; one step = the whole eval forward of the meta-architecture (preprocess, backbone, heads,
sigmoid+clamp, peak-NMS, top-K decode, threshold + detector_postprocess, Instances).  For N > 1 every rank (one per GPU:
started by torch.distributed.run, or by this script itself when it is called with --gpus N and no WORLD_SIZE) runs the
same per-GPU batch on its own shard of images (weak scaling, no data-path collective for inference); the timed region
is bracketed by barrier + torch.cuda.synchronize and the maximum over ranks is reported.

Prints ONE JSON line (rank 0).  The headline (`value`, `dtype`) is the mode that meets north_star's parity bar at the
highest rate: f16x3 -- f32 tensors, every product as three f16 products on the f16 matrix pipe (heat map within 1e-5 of
the fp32 oracle).  With it: the roofline of the dominant kernel (measured live with HIP events around every launch in an
instrumented eager pass, plus the decode's HBM rate), the CPU baseline (the oracle's port of the reference arithmetic
timed on this host's cores, bounded sample), `accuracy` (heat-map error and top-K agreement against the fp32 oracle on
the CPU sample), the sub-records `f16` (f16 storage: the fastest mode, whose error exceeds the 1e-3 bar -- reported, not
the headline) and `f32` (the reference's own arithmetic on the f32 matrix pipe), each with its roofline and accuracy,
and `train` / `train_f32` (BASELINE.json's training half: bs 16 per GPU, whole step, data parallel over RCCL when N > 1,
with the dominant kernel's roofline and a CPU train-step baseline).
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


RES50_YAML = """
_BASE_: "./Base-CenterNet.yaml"
MODEL:
  BACKBONE:
    NAME: "build_resnet_backbone"
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
  STEPS: (225100, 337650)
  MAX_ITER: 450200
  CHECKPOINT_PERIOD: 9004
TEST:
  EVAL_PERIOD: 18008
OUTPUT_DIR: "./output"
VERSION: 2
"""


def build_model(precision, device, seed=0, calibrate=True, config="dla34"):
    import tempfile

    from detectron2_centernet_amd.config import get_cfg
    from detectron2_centernet_amd.data.catalog import register_synthetic
    from detectron2_centernet_amd.modeling import build_model as _build

    d = tempfile.mkdtemp(prefix="ctdet_cfg_")
    with open(os.path.join(d, "Base-CenterNet.yaml"), "w") as f:
        f.write(BASE_YAML)
    name = "ctdet_dla_34_1x.yaml" if config == "dla34" else "ctdet_res_50_1x.yaml"
    with open(os.path.join(d, name), "w") as f:
        f.write(DLA_YAML if config == "dla34" else RES50_YAML)
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(d, name))
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
    if config != "dla34":
        # the reference initialises the deconv / final head weights with std 0.001 (centernet.py:295-320) and fetches ImageNet
        # weights for the ResNet; offline: sane random scales so that activations neither vanish nor overflow
        for m in model.deconv_layers.modules():
            if isinstance(m, torch.nn.ConvTranspose2d):
                m.weight.data.normal_(0, (2.0 / (m.weight.shape[0] * 4)) ** 0.5, generator=None)
        return model, cfg
    if calibrate and torch.device(device).type == "cuda":
        if precision == "f16":
            calibrate_batchnorm(model, seed)
        else:   # the calibration pass runs on the f16 training kernels: take the statistics from an f16 twin (same seed)
            twin, _ = build_model("f16", device, seed, calibrate=True)
            model.load_state_dict(twin.state_dict())
            twin._engines = {}
            del twin
    return model, cfg


def calibrate_batchnorm(model, seed=0, n_images=4, size=512):
    """A random-init DLA-34 in eval mode (running_mean 0 / running_var 1) lets the activations collapse layer by layer: the
    head inputs end up ~1e-6 and the heat map is a constant, on which neither an error figure nor a top-K comparison means
    anything.  One training-mode forward with momentum 1 sets every BatchNorm's running statistics to the batch statistics
    of synthetic images, which is what a trained network's look like: O(1) activations at every layer, heat-map logits
    spread around the -2.19 bias.  Deterministic in `seed` up to f32 reduction order; every consumer (f16 / f32 engines,
    the CPU oracle) reads the resulting state dict."""
    from detectron2_centernet_amd import ops
    from detectron2_centernet_amd.engine import train_step as TS

    if model.backbone_type != "dla34" or model._ctx.compute != ops.F16:
        return
    g = torch.Generator().manual_seed(4321 + seed)
    imgs = torch.randint(0, 256, (n_images, 3, size, size), generator=g, dtype=torch.uint8).to(model.device)
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    saved = [m.momentum for m in bns]
    was_training = model.training
    model.train()
    for m in bns:
        m.momentum = 1.0
    try:
        with torch.no_grad():
            x = ops.preprocess(imgs, model._mean_host, model._std_host, size, size, out_dtype=model._ctx.dtype)
            TS.heads(model, TS.dla34(model.backbone, x)[-1])
    finally:
        for m, mom in zip(bns, saved):
            m.momentum = mom
            m.num_batches_tracked.zero_()
        model.train(was_training)
    torch.cuda.synchronize()


def synthetic_images(B, size, rank, device):
    g = torch.Generator().manual_seed(1234 + rank)
    return torch.randint(0, 256, (B, 3, size, size), generator=g, dtype=torch.uint8).to(device)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_sample_images(size, n_images=2):
    g = torch.Generator().manual_seed(99)
    return [torch.randint(0, 256, (3, size, size), generator=g, dtype=torch.uint8) for _ in range(n_images)]


def cpu_baseline(model, cfg, size, n_images=2, warm=2, reps=5, budget_s=60.0):
    """the oracle (CPU port of the reference arithmetic: torch-CPU convs + the DCNv2 restatement + decode) on a
    bounded sample: BASELINE.json configs[0] = 2 synthetic 512x512 images, forward + decode; `warm` untimed runs,
    then `reps` timed ones (BASELINE.md section 3 asks for >= 5; the run stops early only past `budget_s` of CPU time).
    Returns (record, oracle outputs of the sample) -- the outputs feed the `accuracy` records."""
    from oracle import model_ref as MR

    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    imgs = cpu_sample_images(size, n_images)
    times, out = [], None
    t_start = time.perf_counter()
    with torch.no_grad():
        for i in range(warm + reps):
            t0 = time.perf_counter()
            out = MR.centernet_inference(sd, imgs, cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD, thresh=0.0)
            if i >= warm:
                times.append(time.perf_counter() - t0)
            if len(times) >= reps or (len(times) >= 3 and time.perf_counter() - t_start > budget_s):
                break
    times.sort()
    med = times[len(times) // 2]
    rec = {"value": n_images / med, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
           "cpu": _cpu_model(), "nproc": os.cpu_count(),
           "sample": f"{n_images} synthetic {size}x{size} images, forward+decode, {warm} warm-up runs, median of {len(times)} timed"}
    return rec, out


def cpu_train_baseline(model, cfg, size, n_images=1, warm=1, reps=3, budget_s=60.0):
    """CPU train step of the oracle port (targets + forward in train mode + losses + autograd backward; no optimizer
    update) on a bounded sample of the bs-16 workload: `n_images` images per step."""
    from oracle import ctdet_oracle as O
    from oracle import model_ref as MR
    from detectron2_centernet_amd.data.catalog import synthetic_sample

    sd0 = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    smp = [synthetic_sample(i, size=size, num_classes=80, max_boxes=32) for i in range(n_images)]
    x, _ = O.preprocess([d["image"] for d in smp], cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD, 32)
    times = []
    t_start = time.perf_counter()
    for i in range(warm + reps):
        t0 = time.perf_counter()
        sd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
              for k, v in sd0.items()}
        targets = [O.gen_heatmap(d["boxes"], d["classes"], size // 4, size // 4, 80) for d in smp]
        z = MR.centernet_forward(sd, x, training=True)
        sum(MR.centernet_losses(z, targets, [1.0]).values()).backward()
        if i >= warm:
            times.append(time.perf_counter() - t0)
        if len(times) >= reps or (len(times) >= 2 and time.perf_counter() - t_start > budget_s):
            break
    times.sort()
    med = times[len(times) // 2]
    return {"value": n_images / med, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu": _cpu_model(), "nproc": os.cpu_count(),
            "sample": f"train step (targets+fwd+loss+bwd) on {n_images} synthetic {size}x{size} images (of the 16 per step), "
                      f"{warm} warm-up, median of {len(times)} timed"}


def accuracy_vs_oracle(model, imgs, oracle_out, K=100):
    """the HIP engine on the cpu_baseline sample against the fp32 oracle's outputs for the same images and weights:
    post-sigmoid heat-map error and agreement of the top-K (class, position) lists (north_star: within 1e-3 on fp32 heat
    maps, bit-exact peak indices / top-K)."""
    from oracle import ctdet_oracle as O

    _res, hm_ref, z = oracle_out
    thr, model.score_threshold = model.score_threshold, 0.0
    try:
        with torch.no_grad():
            model([{"image": im} for im in imgs])
    finally:
        model.score_threshold = thr
    B, _, H, W = len(imgs), 3, imgs[0].shape[1], imgs[0].shape[2]
    eng = list(model._engines.values())[-1]      # the engine of the call above (most recently used last)
    hm = eng.out[0].float().cpu().permute(0, 3, 1, 2)
    _b, sc, cl, ind = [t.cpu() for t in eng.dec]
    _rb, rs, rc, ri = O.ctdet_decode(hm_ref, z["wh"], z["reg"], down_ratio=4, K=K)
    same = (cl == rc) & (ind.long() == ri)
    got = [set(zip(cl[b].tolist(), ind[b].tolist())) for b in range(B)]
    ref = [set(zip(rc[b].tolist(), ri[b].tolist())) for b in range(B)]
    return {
        "sample": f"{B} images of the cpu_baseline sample, K={K}, vs the fp32 oracle",
        "hm_max_abs_err": float((hm - hm_ref).abs().max()),
        "score_max_abs_err": float((sc - rs).abs().max()),
        "topk_index_agreement": float(same.float().mean()),            # same (class, position) at the same rank
        "topk_class_agreement": float((cl == rc).float().mean()),
        "topk_set_agreement": sum(len(g & r) for g, r in zip(got, ref)) / float(B * K),   # as unordered sets
    }


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
        _aggregate(ops.PROFILE, agg)
    for a in agg.values():
        for k in ("ms", "flops", "bytes"):
            a[k] /= passes
        a["launches"] //= passes
    return agg


def _aggregate(profile, agg):
    for name, flops, e0, e1, nbytes, _info, reps in profile:
        a = agg.setdefault(name, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
        a["ms"] += e0.elapsed_time(e1) / reps
        a["flops"] += flops
        a["bytes"] += nbytes or 0.0
        a["launches"] += 1
    return agg


def _traffic_for(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over
    this same command, corrected as MI355X_MICROARCH.md prescribes).  The summary records the sha256 of the kernel sources
    it was measured on; a summary of other sources is stale and yields null."""
    import glob
    import hashlib

    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "detectron2-centernet_amd", "csrc", "*"))):
        if f.endswith((".hip", ".h")):
            with open(f, "rb") as fh:
                h.update(fh.read())
    cur = h.hexdigest()[:16]
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):
        with open(path) as f:
            t = json.load(f)
        if t.get("csrc_sha16") == cur:
            return t["bytes_per_launch"].get(kernel), os.path.basename(path)
    return None, None


def roofline_record(agg, peak_tflops, with_traffic):
    conv = {k: v for k, v in agg.items() if v["flops"] > 0}
    name, a = max(conv.items(), key=lambda kv: kv[1]["ms"])
    ach = a["flops"] / (a["ms"] * 1e-3) / 1e12
    total_ms = sum(v["ms"] for v in conv.values())
    total_fl = sum(v["flops"] for v in conv.values())
    traffic, src = _traffic_for(name) if with_traffic else (None, None)
    rec = {
        "bound": "mfma", "kernel": name, "achieved": ach, "peak": peak_tflops, "unit": "TFLOP/s", "frac": ach / peak_tflops,
        "traffic": traffic, "traffic_source": src, "launches_per_step": a["launches"], "kernel_ms_per_step": a["ms"],
        "all_conv_kernels": {"ms_per_step": total_ms, "achieved": total_fl / (total_ms * 1e-3) / 1e12,
                             "frac": total_fl / (total_ms * 1e-3) / 1e12 / peak_tflops},
        "per_kernel": {k: {"ms": round(v["ms"], 4), "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                           "launches": v["launches"]} for k, v in sorted(conv.items(), key=lambda kv: -kv[1]["ms"])},
    }
    d = agg.get("decode")
    if d is not None:   # peak-NMS + top-K decode: HBM-bound, algorithmic bytes = one read of the heat map
        gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
        rec["decode"] = {"bound": "hbm", "ms_per_step": d["ms"], "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes": d["bytes"]}
    return rec


def decode_trained_like(B, H, W, C, K, device, reps=100):
    """the decode on the kind of map a TRAINED network produces (the bench's random-init maps never touch the clamp):
    background exactly on `_sigmoid`'s 1e-4 floor, ~0.3 % of the cells sparse peaks above it; same call as the model's
    (heat_floor promised), timed with HIP events on the launch stream.  The map is generated on the device and 100 launches
    warm the clocks first: ten launches of 0.15 ms after a second of host-side tensor generation measured the clock ramp
    (0.24 ms in one run, 0.15 in the others)."""
    from detectron2_centernet_amd import ops
    g = torch.Generator(device=device).manual_seed(0)
    t = torch.randn(B, H, W, C, generator=g, device=device) - 12.0
    t[:, ::17, ::13, ::7] += 11.0
    hm = torch.clamp(torch.sigmoid(t), 1e-4, 1 - 1e-4)
    whreg = torch.rand(B, H, W, 4, generator=g, device=device)
    ws = ops.DecodeWorkspace(B, H, W, C, K, device)
    call = lambda: ops.decode(hm, whreg[..., :2], whreg[..., 2:], K, 4.0, workspace=ws, heat_floor=ops.SIGMOID_CLAMP_FLOOR)
    for _ in range(100):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gbs = hm.numel() * 4 / (ms * 1e-3) / 1e9
    return {"map": f"{B}x{H}x{W}x{C}: background on the 1e-4 clamp, {float((hm > 1e-4).float().mean()) * 100:.2f} % of the cells above it",
            "ms_per_step": ms, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "algorithmic_bytes": float(hm.numel() * 4)}


def timed_infer(model, images, steps, warmup, dist, backend, device):
    """serving loop with one step in flight: the next batch is enqueued before the host reads back the previous batch's
    detection counts and builds its Instances; every step's full result is materialised inside the timed region"""
    with torch.no_grad():
        for _ in range(warmup):
            model.infer_batch_tensor(images)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pending = None
        for _ in range(steps):
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
    return elapsed, out


def train_roofline(model, cfg, B, size, rank, device):
    """one instrumented eager training step (no graph): HIP events around the backward-side launches and every conv"""
    from detectron2_centernet_amd import ops
    from detectron2_centernet_amd.engine.bench_train import synthetic_batch

    images, boxes, classes, counts = synthetic_batch(B, size, rank, device)
    ops.PROFILE.clear()
    ops.PROFILE_ON = True
    # only rank 0 runs this step: a reducer of a data-parallel trainer that is still hooked to the parameters must not start
    # collectives from it (the other ranks are not there to answer)
    reducers = {id(r): r for r in (getattr(getattr(p, "_ctdet_grad_hook", None), "reducer", None) for p in model.parameters()) if r is not None}
    for r in reducers.values():
        r.enabled = False
    try:
        losses = model.train_batch_tensor(images, boxes, classes, counts)
        sum(losses.values()).backward()
    finally:
        ops.PROFILE_ON = False
        for r in reducers.values():
            r.enabled = True
    torch.cuda.synchronize()
    agg = _aggregate(ops.PROFILE, {})
    ops.PROFILE.clear()
    model.zero_grad(set_to_none=True)
    name, a = max(agg.items(), key=lambda kv: kv[1]["ms"])
    if a["flops"] > 0:
        mfma_peak = PEAKS[{ops.F16: "f16", ops.F32: "f32", ops.F16X3: "f16x3"}[model._ctx.compute]]
        ach, peak, unit, bound = a["flops"] / (a["ms"] * 1e-3) / 1e12, mfma_peak, "TFLOP/s", "mfma"
    else:
        ach, peak, unit, bound = a["bytes"] / (a["ms"] * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s", "hbm"
    return {"bound": bound, "kernel": name, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak, "traffic": None,
            "launches_per_step": a["launches"], "kernel_ms_per_step": a["ms"],
            "per_kernel": {k: {"ms": round(v["ms"], 4), "launches": v["launches"],
                               "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                               "GBs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)}
                           for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}}


PEAKS = {"f16": MFMA_F16_PEAK_TFLOPS, "f32": FP32_PEAK_TFLOPS, "f16x3": MFMA_F16_PEAK_TFLOPS / 3.0}
MODE_NOTES = {
    "f16x3": "f32 tensors; every product as a_hi*w_hi + a_lo*w_hi + a_hi*w_lo on the f16 matrix pipe, f32 accumulation "
             "(f32-grade results); roofline peak = 2.5 PFLOP/s / 3 products",
    "f16": "f16 activations and weights, f32 accumulation: the fastest mode; its heat-map error is above north_star's 1e-3, "
           "so it is reported here and is not the headline",
    "f32": "the reference's own arithmetic (fp32 in, fp32 accumulate) on the f32 matrix pipe; peak 157.3 TFLOP/s: "
           "64x3x512x512 needs >= 26.9 ms per step",
}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU, rendezvous on 127.0.0.1) and
    relay rank 0's line; non-zero exit if any rank fails.  Runs BEFORE this process touches the GPU.  With fewer visible GPUs
    than ranks the ranks share devices and the timing reduction goes over gloo (a rehearsal: the line says so)."""
    import socket
    import subprocess

    import tempfile
    import time

    n = args.gpus
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ndev = torch.cuda.device_count()       # does not initialise the GPU on this image
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if ndev < n:
            env.setdefault("CTDET_BENCH_BACKEND", "gloo")
        out = tempfile.TemporaryFile()
        logs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    # supervise every rank: the first one that fails takes the others down (a rank that died before the rendezvous would
    # otherwise leave the rest in init_process_group / a barrier until the collective timeout), and the whole job has a deadline
    deadline = time.time() + float(os.environ.get("CTDET_BENCH_TIMEOUT", "3000"))
    rcs = [None] * n
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        failed = [i for i, rc in enumerate(rcs) if rc not in (None, 0)]
        if failed or time.time() > deadline:
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    try:
                        rcs[i] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        rcs[i] = p.wait()
            why = f"rank(s) {failed} failed" if failed else "deadline passed"
            raise SystemExit(f"bench.py --gpus {n}: {why}; rank exit codes {rcs}")
        time.sleep(0.2)
    logs[0].seek(0)
    out = logs[0].read().decode()
    sys.stdout.write(out)
    sys.stdout.flush()
    line = [ln for ln in out.splitlines() if ln.startswith("{")]
    if not line or json.loads(line[-1]).get("n_gpus") != n:
        raise SystemExit(f"bench.py --gpus {n}: fewer than {n} ranks joined")


def infer_record(precision, state, images, steps, warmup, dist, backend, device, world, config, with_roofline, headline,
                 oracle_out, size):
    """one precision of the inference workload: timed serving loop, roofline of its kernels, accuracy against the oracle"""
    model, cfg = build_model(precision, device, calibrate=state is None, config=config)
    if state is not None:
        model.load_state_dict(state)      # the same weights and calibrated BatchNorm statistics in every mode
    model.eval()
    B = images.shape[0]
    elapsed, out = timed_infer(model, images, steps, warmup, dist, backend, device)
    assert len(out) == B
    rec = {"value": world * B * steps / elapsed, "unit": "images/s", "ms_per_step": 1000.0 * elapsed / steps, "steps": steps,
           "warmup": warmup, "dtype": precision, "note": MODE_NOTES[precision],
           "detections_per_image": sum(len(o["instances"]) for o in out) / max(1, len(out)),
           # node types of the captured eval step (kernels only: no memset / memcpy node; engine/graph_nodes.py)
           "graph_nodes": next((e.graph_nodes for e in model._engines.values() if e.B == B and e.graph is not None), None)}
    if with_roofline:
        rec["roofline"] = roofline_record(roofline_pass(model, images, passes=2 if headline else 1), PEAKS[precision],
                                          with_traffic=headline)
        if headline and "decode" in rec["roofline"]:
            r = model.backbone.down_ratio
            rec["roofline"]["decode"]["trained_like"] = decode_trained_like(
                B, images.shape[-2] // r, images.shape[-1] // r, model.num_classes, model.topk_candidates, device)
    return rec, model, cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--task", default="infer", choices=["infer", "train"])
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default 64 infer / 16 train)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "f16", "f32"],
                    help="headline mode (default f16x3: the fastest mode inside north_star's 1e-3 parity bar)")
    ap.add_argument("--config", default="dla34", choices=["dla34", "r50"],
                    help="dla34: BASELINE.json configs[1] (+[2]); r50: configs[4], ResNet-50 CenterNet 800x800 bs 8 per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-f32", action="store_true", help="skip the f32 sub-records (inference and training)")
    ap.add_argument("--no-f16", action="store_true", help="skip the f16 inference sub-record")
    ap.add_argument("--no-train", action="store_true", help="skip the training sub-records")
    ap.add_argument("--require-graph", action="store_true", help="fail if the training step did not replay as a HIP graph")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
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
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but {dist.get_world_size()} ranks joined the group")
        # one rank per GPU on the RCCL path: every rank reports the device it sits on (PCI bus id / uuid where torch exposes
        # them, ordinal otherwise) and the set must have N members
        props = torch.cuda.get_device_properties(device)
        ident = f"{os.uname().nodename}:{getattr(props, 'uuid', None) or getattr(props, 'pci_bus_id', None) or dev_index}:{dev_index}"
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        if backend == "nccl" and len(set(idents)) != world:
            raise SystemExit(f"--gpus {args.gpus} over RCCL needs {world} distinct devices, the ranks sit on {idents}")

    if args.config == "r50":
        args.size = 800 if args.size == 512 else args.size
        args.batch = args.batch or 8
        args.no_f32 = args.no_f16 = args.no_cpu_baseline = True
        if args.precision == "f16x3":
            args.precision = "f16"          # the ResNet config is the MFMA / xGMI stress case: its throughput mode

    if args.task == "train":
        from detectron2_centernet_amd.engine.bench_train import run_train_bench
        model, cfg = build_model(args.precision, device, config=args.config)
        result = run_train_bench(model, cfg, args, args.batch or 16, rank, world, device, dist)
        if rank == 0 and not args.no_roofline:
            result["roofline"] = train_roofline(model, cfg, args.batch or 16, args.size, rank, device)
        if args.require_graph and world == 1 and result["config"]["graph_state"] != "captured":
            raise SystemExit(f"training step did not replay as a HIP graph: {result['config']['graph_state']}")
    else:
        B = args.batch or 64
        headline = B == 64 and args.size == 512 and args.config == "dla34"
        images = synthetic_images(B, args.size, rank, device)
        solo = rank == 0 and world == 1
        rec, model, cfg = infer_record(args.precision, None, images, args.steps, args.warmup, dist, backend, device, world,
                                       args.config, rank == 0 and not args.no_roofline, headline, None, args.size)
        result = {
            "metric": "images/sec at 512x512 (infer bs=64)" if headline else
                      f"images/sec at {args.size}x{args.size} (infer bs={B})",
            "value": rec["value"],
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": rec["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "config": {"workload": f"{'DLA-34' if args.config == 'dla34' else 'ResNet-50'} CenterNet eval forward+decode, {B}x3x{args.size}x{args.size} uint8 per GPU, "
                                   "80 classes, K=100, random-init weights (BatchNorm statistics calibrated), DCN offsets ~N(0,1px), "
                                   f"{rec['detections_per_image']:.0f} detections per image post-processed",
                       "precision": MODE_NOTES[args.precision],
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"replicas x{world}",
                       "ranks_share_devices": backend != "nccl" and world > 1,
                       "graph_nodes": rec.get("graph_nodes")},
        }
        if "roofline" in rec:
            result["roofline"] = rec["roofline"]
        oracle_out = None
        if solo and not args.no_cpu_baseline:
            result["cpu_baseline"], oracle_out = cpu_baseline(model, cfg, args.size)
            # the same two images through the HIP engine: error of this mode against the reference's fp32 arithmetic
            result["accuracy"] = accuracy_vs_oracle(model, cpu_sample_images(args.size), oracle_out)
        state = {k: v.clone() for k, v in model.state_dict().items()}
        model._engines = {}
        del model
        torch.cuda.empty_cache()

        # ---- the other modes on the same workload and weights (single-GPU runs only)
        if solo and args.config == "dla34":
            for mode, skip, st, wu in (("f16", args.no_f16, max(5, args.steps // 2), 3),
                                       ("f32", args.no_f32, max(10, args.steps // 5), 2)):
                if skip or mode == args.precision:
                    continue
                sub, m, _ = infer_record(mode, state, images, st, wu, None, backend, device, 1, args.config,
                                         not args.no_roofline, False, oracle_out, args.size)
                if oracle_out is not None:
                    sub["accuracy"] = accuracy_vs_oracle(m, cpu_sample_images(args.size), oracle_out)
                result[mode] = sub
                m._engines = {}
                del m
                torch.cuda.empty_cache()

        # ---- the training half of BASELINE.json's metric (train bs=16/GPU): every rank, data parallel when world > 1
        if (headline or args.config == "r50") and not args.no_train:
            from detectron2_centernet_amd.engine.bench_train import run_train_bench

            tB = 16 if args.config == "dla34" else B
            # `train` = the parity-grade mode on the matrix pipe (f16x3: f32 tensors, split products), as the inference headline;
            # `train_f16` the throughput mode (outside the tolerance), `train_f32` the reference's own arithmetic
            main_prec = "f16x3" if args.config == "dla34" else "f16"
            for key, prec, skip in (("train", main_prec, False),
                                    ("train_f16", "f16", args.no_f16 or args.config != "dla34"),
                                    ("train_f32", "f32", args.no_f32 or args.config != "dla34")):
                if skip:
                    continue
                tm, tcfg = build_model(prec, device, config=args.config)
                few = prec == "f32"
                targs = argparse.Namespace(steps=max(5, args.steps // (5 if few else 2)), warmup=3 if few else max(4, args.warmup // 2),
                                           size=args.size)
                tr = run_train_bench(tm, tcfg, targs, tB, rank, world, device, dist)
                trec = {k: tr[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "scaling")}
                trec["graph_state"] = tr["config"]["graph_state"]
                trec["config"] = tr["config"]
                if args.require_graph and world == 1 and trec["graph_state"] != "captured":
                    raise SystemExit(f"training step ({prec}) did not replay as a HIP graph: {trec['graph_state']}")
                if rank == 0 and not args.no_roofline:
                    trec["roofline"] = train_roofline(tm, tcfg, tB, args.size, rank, device)
                if solo and not args.no_cpu_baseline and key == "train":
                    trec["cpu_baseline"] = cpu_train_baseline(tm, tcfg, args.size)
                result[key] = trec
                del tm
                torch.cuda.empty_cache()

    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
