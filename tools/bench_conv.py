#!/usr/bin/env python
"""Dev tool: time one conv-shaped launch (HIP events, repeated launches) -- used to tune the kernels and as the
target of rocprofv3 --pmc runs.  Not part of the product or of the tests."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectron2_centernet_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=64)
ap.add_argument("--H", type=int, default=32)
ap.add_argument("--W", type=int, default=32)
ap.add_argument("--cin", type=int, default=256)
ap.add_argument("--cout", type=int, default=256)
ap.add_argument("--k", type=int, default=3)
ap.add_argument("--stride", type=int, default=1)
ap.add_argument("--dcn", action="store_true")
ap.add_argument("--window", action="store_true")
ap.add_argument("--off-std", type=float, default=1.0)
ap.add_argument("--f32out", action="store_true")
ap.add_argument("--reps", type=int, default=1000, help="timed launches; short runs (tens of launches) measure the clock ramp, not the kernel")
ap.add_argument("--tap-major", action="store_true")
ap.add_argument("--f32", action="store_true", help="the f32 (reference precision) kernels")
ap.add_argument("--f16x3", action="store_true", help="f32 tensors, split f16 products (the f16x3 mode)")
ap.add_argument("--tune", type=int, default=0, help="ctdet_set_tuning_flags bits (see _lib.TUNE_*)")
a = ap.parse_args()
dev = torch.device("cuda:0")
if a.tune:
    from detectron2_centernet_amd import _lib
    _lib.lib().ctdet_set_tuning_flags(a.tune)
g = torch.Generator().manual_seed(0)
x = torch.randn(a.B, a.H, a.W, a.cin, generator=g).half().to(dev)
if a.f32 or a.f16x3:
    a.f32 = True
    x = x.float()
w = (torch.randn(a.cout, a.cin, a.k, a.k, generator=g) / (a.cin * a.k * a.k) ** 0.5).to(dev)
p = ops.PackedConv(w, None, None, stride=a.stride, pad=a.k // 2, compute=ops.F16X3 if a.f16x3 else (ops.F32 if a.f32 else ops.F16),
                   tap_major=a.tap_major and not a.dcn)
od = torch.float32 if (a.f32out or a.f32) else torch.float16
if a.dcn:
    om = torch.randn(a.B, a.H, a.W, 28, generator=g)
    om[..., :18] *= a.off_std
    om = om.to(dev)
    f = lambda: ops.dcnv2(x, om, p, act=ops.ACT_RELU, out_dtype=od)
else:
    f = lambda: ops.conv2d(x, p, act=ops.ACT_RELU, out_dtype=od)
for _ in range(200):
    y = f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    y = f()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
Ho, Wo = y.shape[1], y.shape[2]
fl = 2.0 * a.B * Ho * Wo * a.cout * a.cin * a.k * a.k
print(f"{'dcn' if a.dcn else 'conv'} B{a.B} {a.H}x{a.W} {a.cin}->{a.cout} k{a.k} s{a.stride}: {ms*1000:.1f} us  {fl/ms/1e9:.1f} TFLOP/s")
