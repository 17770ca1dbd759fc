#!/usr/bin/env python
"""Dev tool: time the fused DLA base kernel (normalisation + stem + level0 + level1) at the bench shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectron2_centernet_amd import _lib, ops  # noqa: E402

if os.environ.get("CTDET_BENCH_LIB"):
    _lib.LIB_PATH = os.environ["CTDET_BENCH_LIB"]

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
X3 = len(sys.argv) > 3 and sys.argv[3] == "f16x3"
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
img = torch.randint(0, 256, (B, 3, S, S), generator=g, dtype=torch.uint8).to(dev)
one = lambda c: (torch.ones(c, device=dev), torch.zeros(c, device=dev))
pb = (ops.PackedDlaBaseX3 if X3 else ops.PackedDlaBase)(torch.randn(16, 3, 7, 7, generator=g).to(dev) / 12, one(16), torch.randn(16, 16, 3, 3, generator=g).to(dev) / 12,
                       one(16), torch.randn(32, 16, 3, 3, generator=g).to(dev) / 12, one(32))
out = torch.empty(B, S // 2, S // 2, 32, dtype=torch.float32 if X3 else torch.float16, device=dev)
pool = torch.empty(B, S // 4, S // 4, 32, dtype=out.dtype, device=dev)
f = lambda: ops.dla_base_fused(img, [0.4, 0.45, 0.48], [0.22, 0.22, 0.23], S, S, pb, out=out, pooled=pool)
for _ in range(3):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    f()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
px = B * S * S
flops = 2.0 * (px * 16 * 147 + px * 16 * 144 + px / 4 * 32 * 144)
print(f"dla_base {'f16x3' if X3 else 'f16'} B{B} {S}x{S}: {ms*1000:.1f} us  {flops/ms/1e9:.1f} TFLOP/s (useful)  {(img.numel() + out.numel() * out.element_size() * 1.25)/ms/1e6:.0f} GB/s")
