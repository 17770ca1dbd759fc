import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectron2_centernet_amd import ops, _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
worst = 0.0
for (B, H, W, Cin, Cout) in ((1, 8, 32, 64, 64), (2, 16, 64, 128, 27), (3, 24, 32, 64, 80), (2, 8, 64, 256, 256), (1, 16, 32, 192, 128), (2, 8, 32, 64, 16)):
    x = torch.randn(B, Cin, H, W, generator=g).half().float()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).half().float()
    bias = torch.randn(Cout, generator=g)
    ref = (F.conv2d(x, w, None, 1, 1) + bias.view(1, -1, 1, 1)).relu()
    pc = ops.PackedConv(w.to(dev), None, bias.to(dev), stride=1, pad=1, compute=ops.F16)
    xd = x.permute(0, 2, 3, 1).contiguous().half().to(dev)
    outs = []
    for tune in (0, _lib.TUNE_NO_HALO_TAP2):
        _lib.lib().ctdet_set_tuning_flags(tune)
        for od in (torch.float32, torch.float16):
            y = ops.conv2d(xd, pc, act=ops.ACT_RELU, out_dtype=od)
            outs.append(y[..., :Cout].float().cpu().permute(0, 3, 1, 2))
    _lib.lib().ctdet_set_tuning_flags(0)
    e_old, e_new = (outs[0] - ref).abs().max().item(), (outs[2] - ref).abs().max().item()
    d32, d16 = (outs[0] - outs[2]).abs().max().item(), (outs[1] - outs[3]).abs().max().item()
    print(f"B{B} {H}x{W} {Cin}->{Cout}: err old {e_old:.2e} new {e_new:.2e}  old-vs-new f32 {d32:.2e} f16 {d16:.2e}")
    worst = max(worst, e_new / max(1.0, ref.abs().max().item()))
print("worst rel err", worst)
assert worst < 1e-5
