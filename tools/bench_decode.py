#!/usr/bin/env python
"""Dev tool: time the batched decode on a synthetic bs-64 heatmap."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectron2_centernet_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B = 64
hm = torch.clamp(torch.sigmoid(torch.randn(B, 128, 128, 80, generator=g) * 0.2 - 2.19), 1e-4, 1 - 1e-4).to(dev)
whreg = torch.rand(B, 128, 128, 4, generator=g).to(dev)
ws = ops.DecodeWorkspace(B, dev)
for _ in range(2):
    ops.decode(hm, whreg[..., :2], whreg[..., 2:], 100, 4.0, workspace=ws)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.decode(hm, whreg[..., :2], whreg[..., 2:], 100, 4.0, workspace=ws)
e1.record(); torch.cuda.synchronize()
print(f"decode bs{B}: {e0.elapsed_time(e1)/10*1000:.0f} us  (CTDET_DEC_DEBUG={os.environ.get('CTDET_DEC_DEBUG','0')})")
