"""microbench of the DCNv2 backward scatter (d(input), d(offset), d(mask)): python tools/bench_col2im.py B H W Cin [off_std]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import detectron2_centernet_amd  # noqa: F401,E402
from detectron2_centernet_amd import _lib, ops, ops_train as ot  # noqa: E402

if os.environ.get("CTDET_BENCH_LIB"):
    _lib.LIB_PATH = os.environ["CTDET_BENCH_LIB"]
X3 = os.environ.get("PREC", "f16") == "f16x3"      # f32 tensors, the f32 LDS-window scatter of the f16x3 training mode

B, H, W, Cin = [int(v) for v in sys.argv[1:5]]
std = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
dev = torch.device("cuda:0")
if os.environ.get("TUNE"):       # ctdet_set_tuning_flags bits, e.g. TUNE=128: the LDS fixed-point scatter kernel
    _lib.lib().ctdet_set_tuning_flags(int(os.environ["TUNE"]))
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, W, Cin, generator=g).half().to(dev)
dcol = torch.randn(B, H, W, 9 * Cin, generator=g).half().to(dev)
kw = {}
if X3:
    x, dcol, kw = x.float(), dcol.float(), {"comp": ops.F16X3}
om = torch.randn(B, H, W, 28, generator=g)
om[..., :18] *= std
om = om.to(dev)
for _ in range(2):
    ot.dcn_col2im_coord(dcol, x, om, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ot.dcn_col2im_coord(dcol, x, om, **kw)
e1.record()
torch.cuda.synchronize()
print(f"col2im B{B} {H}x{W} Cin{Cin} off_std {std} TUNE={os.environ.get('TUNE', '0')}: {e0.elapsed_time(e1) / 5 * 1000:.0f} us")
