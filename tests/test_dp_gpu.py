"""Data-parallel training step on the GPU: two ranks sharing the device, gloo collectives on the device tensors (the rehearsal
mode of `bench.py --gpus N`; the RCCL run needs as many GPUs as ranks).  Both ranks get the SAME batch, so the averaged
gradient equals the single-rank gradient and the parameters after the step must equal those of a single-process step --
through everything that differs in the data-parallel path: eager backward, gradients written straight into the flat buffer,
the per-bucket tap-major -> OIHW scatter launched from the reducer, the hooks that count a bucket's parameters (called by
autograd for some parameters and by the backward kernels' wrappers for the others), the bucketed all-reduce."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _one_step(outfile):
    import bench
    from detectron2_centernet_amd.engine.bench_train import synthetic_batch
    from detectron2_centernet_amd.engine.train_loop import SimpleTrainer

    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    model, cfg = bench.build_model("f16", dev, seed=5, calibrate=False)
    model.train()
    cfg.SOLVER.IMS_PER_BATCH = 2
    tr = SimpleTrainer(model, None, cfg)
    if tr.reducer.world > 1:
        assert len(tr.reducer.buckets) >= 2      # the exchange really is bucketed
    batch = synthetic_batch(2, 128, 0, dev)       # rank argument fixed: identical data on every rank
    tr.run_step_tensors(*batch)
    torch.cuda.synchronize()
    names = {id(p): n for n, p in model.named_parameters()}
    layout = [(names[id(p)], off, n) for p, (off, n) in zip(tr.optimizer.params, tr.optimizer.offsets)]
    torch.save({"param": tr.optimizer.flat_param.cpu(), "grad": tr.optimizer.flat_grad.cpu(),
                "world": tr.reducer.world, "layout": layout, "calls": dict(tr.reducer.calls), "buckets": [tuple(b) for b in tr.reducer.buckets]}, outfile)


def _worker(outdir):
    from detectron2_centernet_amd.utils import comm

    _one_step(os.path.join(outdir, f"rank{comm.get_rank()}.pt"))


def test_two_rank_step_equals_single_rank_step(tmp_path):
    from detectron2_centernet_amd.engine import launch

    os.environ["CTDET_TRAIN_GRAPH"] = "0"
    try:
        _one_step(str(tmp_path / "single.pt"))
        _one_step(str(tmp_path / "single2.pt"))     # the run-to-run noise of the step itself (order of the f32 atomics)
        launch(_worker, 2, num_machines=1, machine_rank=0, dist_url="auto", args=(str(tmp_path),), backend="gloo")
    finally:
        os.environ.pop("CTDET_TRAIN_GRAPH", None)
    ref = torch.load(tmp_path / "single.pt")
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert ref["world"] == 1 and r0["world"] == 2 and r1["world"] == 2
    # DLA-34 has parameters that never receive a gradient (the `project` of the multi-level Trees: dla.py computes it and the
    # inner Tree ignores it); their bucket goes out in finish()
    missing = [r0["layout"][i][0] for i in range(len(r0["layout"])) if i not in r0["calls"]]
    assert all(".project." in n for n in missing), missing
    if not torch.equal(r0["grad"], r1["grad"]):                # one all-reduce result on both ranks
        d = (r0["grad"] - r1["grad"]).abs()
        bad = [(n, d[o:o + k].max().item(), r0["grad"][o:o + k].abs().max().item()) for n, o, k in r0["layout"]
               if d[o:o + k].max().item() > 0]
        raise AssertionError(f"ranks disagree after the exchange in {len(bad)} parameters, buckets {r0['buckets']}: {bad[:12]}")
    gs = ref["grad"].abs().max().item()
    assert gs > 0
    ref2 = torch.load(tmp_path / "single2.pt")
    noise = (ref2["grad"] - ref["grad"]).abs().max().item()
    # each rank's gradient is pre-divided by the world size and the two are summed: the single-rank gradient up to the
    # run-to-run noise of the step (f16 activations of a random-init net amplify the atomics' rounding order)
    assert (r0["grad"] - ref["grad"]).abs().max().item() <= 4 * noise + 5e-3 * gs, (noise, gs)
    cos = torch.nn.functional.cosine_similarity(r0["grad"].double(), ref["grad"].double(), dim=0).item()
    cos_noise = torch.nn.functional.cosine_similarity(ref2["grad"].double(), ref["grad"].double(), dim=0).item()
    assert cos > 0.9999 and 1 - cos <= 4 * (1 - cos_noise) + 2e-5, (cos, cos_noise)
    pnoise = (ref2["param"] - ref["param"]).abs().max().item()
    assert (r0["param"] - ref["param"]).abs().max().item() <= 4 * pnoise + 1e-5
    assert torch.equal(r0["param"], r1["param"])


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE starts two ranks itself (before touching the GPU) and
    rank 0 prints ONE line with n_gpus == 2; with fewer GPUs than ranks the ranks share the device and the line says so"""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "CTDET_BENCH_BACKEND")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline", "--no-f32", "--no-f16", "--no-train", "--no-roofline"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dtype"] == "f16x3" and rec["value"] > 0
    assert rec["config"]["ranks_share_devices"] == (torch.cuda.device_count() < 2)
