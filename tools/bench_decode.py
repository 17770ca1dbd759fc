#!/usr/bin/env python
"""Dev tool: time the batched decode on synthetic bs-64 heat maps of different character."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectron2_centernet_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B = 64
whreg = torch.rand(B, 128, 128, 4, generator=g).to(dev)
ws = ops.DecodeWorkspace(B, 128, 128, 80, 100, dev)


def run(name, hm):
    for floor in (ops.SIGMOID_CLAMP_FLOOR, 0.0):     # the model's call (clamp floor promised) / the plain call
        for _ in range(2):
            ops.decode(hm, whreg[..., :2], whreg[..., 2:], 100, 4.0, workspace=ws, heat_floor=floor)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.decode(hm, whreg[..., :2], whreg[..., 2:], 100, 4.0, workspace=ws, heat_floor=floor)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1000
        print(f"decode bs{B} {name:28s} heat_floor={floor:<6g}: {us:6.0f} us  = {hm.numel() * 4 / us / 1e3:6.0f} GB/s of heat-map reads", flush=True)


clamp = lambda t: torch.clamp(torch.sigmoid(t), 1e-4, 1 - 1e-4)
run("spread (sigma 0.5 logits)", clamp(torch.randn(B, 128, 128, 80, generator=g) * 0.5 - 2.19).to(dev))
run("narrow band (sigma 1e-3)", clamp(torch.randn(B, 128, 128, 80, generator=g) * 1e-3 - 2.19).to(dev))
run("constant", torch.full((B, 128, 128, 80), 0.1, device=dev))
t = torch.randn(B, 128, 128, 80, generator=g) * 1.0 - 12.0      # trained-like: background clamps to 1e-4, few peaks
t[:, ::17, ::13, ::7] += 11.0
run("trained-like (sparse peaks)", clamp(t).to(dev))
