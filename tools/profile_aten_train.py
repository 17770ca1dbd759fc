"""ATen kernels left in one eager training step: python tools/profile_aten_train.py
(torch.profiler totals per operator, then the python call sites counted by wrapping the tensor constructors / casts)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["CTDET_TRAIN_GRAPH"] = "0"
import bench  # noqa: E402
from detectron2_centernet_amd.engine.bench_train import synthetic_batch  # noqa: E402
from detectron2_centernet_amd.engine.train_loop import SimpleTrainer  # noqa: E402

dev = torch.device("cuda:0")
model, cfg = bench.build_model("f16", dev)
model.train()
cfg.SOLVER.IMS_PER_BATCH = 16
trainer = SimpleTrainer(model, None, cfg)
batch = synthetic_batch(16, 512, 0, dev)
for _ in range(3):
    trainer.run_step_tensors(*batch)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    trainer.run_step_tensors(*batch)
    torch.cuda.synchronize()
rows = [(getattr(ev, "self_device_time_total", 0) or 0, ev.count, ev.key) for ev in prof.key_averages()]
rows = sorted(r for r in rows if r[0] > 0 and r[2].startswith("aten::"))[::-1]
print(f"ATen device time in one eager step: {sum(r[0] for r in rows) / 1000:.2f} ms over {sum(r[1] for r in rows)} operator calls")
for t, n, k in rows[:12]:
    print(f"{t / 1000:7.3f} ms {n:4d}x  {k}")
print("python call sites that launch them (forward and custom backward functions; autograd's own accumulation is not listed):")

import traceback  # noqa: E402

sites = collections.Counter()


def _site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "centernet_amd" in fr.filename and "tools" not in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:90]}"
    return "?"


def _wrap(obj, name, label, cond=None):
    orig = getattr(obj, name)

    def f(*a, **k):
        if cond is None or cond(*a, **k):
            sites[(label, _site())] += 1
        return orig(*a, **k)
    setattr(obj, name, f)


_wrap(torch, "zeros", "zeros")
_wrap(torch, "zeros_like", "zeros_like")
_wrap(torch, "cat", "cat")
_wrap(torch.nn.functional, "pad", "pad")
_wrap(torch.Tensor, "contiguous", "contiguous(copy)", lambda t, *a, **k: not t.is_contiguous())
_wrap(torch.Tensor, "to", "to(dtype)", lambda t, *a, **k: any(isinstance(x, torch.dtype) and x != t.dtype for x in list(a) + list(k.values())))
_wrap(torch.Tensor, "half", "half", lambda t: t.dtype != torch.float16)
_wrap(torch.Tensor, "float", "float", lambda t: t.dtype != torch.float32)
_wrap(torch.Tensor, "clone", "clone")
_wrap(torch.Tensor, "copy_", "copy_")
_wrap(torch.Tensor, "zero_", "zero_")
_wrap(torch.Tensor, "fill_", "fill_")
_wrap(torch.Tensor, "__add__", "add")
_wrap(torch.Tensor, "__mul__", "mul")
trainer.run_step_tensors(*batch)
torch.cuda.synchronize()
for (label, site), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d}x {label:18s} {site}")
