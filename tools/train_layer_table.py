"""Dev tool: one line per profiled launch region of ONE eager training step at BASELINE's per-GPU shape (16 x 3 x 512^2): time,
TFLOP/s, algorithmic GB/s, what it was.  usage: python tools/train_layer_table.py [f16x3|f16|f32] [name filter]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from detectron2_centernet_amd import ops  # noqa: E402
from detectron2_centernet_amd.engine.bench_train import synthetic_batch  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda:0")
model, cfg = bench.build_model(prec, dev, calibrate=False)
model.train()
batch = synthetic_batch(16, 512, 0, dev)
for it in range(2):
    ops.PROFILE.clear()
    ops.PROFILE_ON = it == 1
    losses = model.train_batch_tensor(*batch)
    sum(losses.values()).backward()
    ops.PROFILE_ON = False
    model.zero_grad(set_to_none=True)
torch.cuda.synchronize()
tot = 0.0
for name, flops, e0, e1, nbytes, info, reps in ops.PROFILE:
    ms = e0.elapsed_time(e1) / reps
    tot += ms
    if flt in name:
        print(f"{ms * 1000:9.1f} us {flops / ms / 1e9:8.1f} TF/s {nbytes / ms / 1e6:8.0f} GB/s  {name}  [{info}]")
print(f"profiled total {tot:.3f} ms")
