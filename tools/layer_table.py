"""Dev tool: one line per conv-shaped launch of the bench workload's eval forward (instrumented eager pass, every
launch repeated ops.PROFILE_REP times between one HIP event pair): time, TFLOP/s, algorithmic GB/s and the fraction of
max(MFMA time, HBM time) at 2.5 PFLOP/s (f16; f16x3: a third of it; f32: 157.3 TFLOP/s) / 8 TB/s.
usage: python3 tools/layer_table.py [batch] [f16|f32|f16x3] [ctdet tuning flags]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from detectron2_centernet_amd import ops  # noqa: E402
from detectron2_centernet_amd.modeling.meta_arch.centernet import _EvalEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
PREC = sys.argv[2] if len(sys.argv) > 2 else "f16"
if len(sys.argv) > 3:
    from detectron2_centernet_amd import _lib  # noqa: E402
    _lib.lib().ctdet_set_tuning_flags(int(sys.argv[3]))
PEAK = {"f16": 2.5e15, "f32": 157.3e12, "f16x3": 2.5e15 / 3}[PREC]
dev = torch.device("cuda:0")
model, cfg = bench.build_model(PREC, dev)
model.eval()
images = bench.synthetic_images(B, 512, 0, dev)
eng = _EvalEngine(model, B, 512, 512, 512, 512, images.dtype, use_graph=False)
eng.images.copy_(images)
with torch.no_grad():
    eng()
    ops.PROFILE.clear()
    ops.PROFILE_ON = True
    eng()
    ops.PROFILE_ON = False
torch.cuda.synchronize()
tot = 0.0
ideal = 0.0
print(f"{'us':>8} {'TF/s':>7} {'GB/s':>7} {'frac':>5}  kernel / layer")
for name, flops, e0, e1, nbytes, info, reps in ops.PROFILE:
    ms = e0.elapsed_time(e1) / reps
    t_roof = max(flops / PEAK, nbytes / 8e12) * 1e3
    tot += ms
    ideal += t_roof
    print(f"{ms*1e3:8.1f} {flops/ms/1e9:7.1f} {nbytes/ms/1e6:7.0f} {t_roof/ms:5.2f}  {name}  [{info}]")
print(f"total {tot:.3f} ms, roofline sum {ideal:.3f} ms")
