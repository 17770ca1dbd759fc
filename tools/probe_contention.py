"""Probe (round 4): which backward kernel gives run-to-run different results when a SECOND process shares the GPU?
Two data-parallel ranks on one device showed gradients that deviate from the single-process ones from the DCN nodes on
(tests/test_dp_gpu.py); the single-process step is reproducible to 1e-7.  Rank A repeats the pieces of one DeformConvV2
backward (and a few plain convs) on fixed inputs and compares every output with its first result; rank B (a child process)
runs training steps next to it."""
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def hammer():
    import bench
    from detectron2_centernet_amd.engine.bench_train import synthetic_batch
    from detectron2_centernet_amd.engine.train_loop import SimpleTrainer
    dev = torch.device("cuda:0")
    model, cfg = bench.build_model("f16", dev, calibrate=False)
    cfg.SOLVER.IMS_PER_BATCH = 16
    tr = SimpleTrainer(model, None, cfg)
    batch = synthetic_batch(16, 512, 0, dev)
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]):
        tr.run_step_tensors(*batch)
        torch.cuda.synchronize()


def main():
    if sys.argv[1] == "hammer":
        return hammer()
    contend = sys.argv[1] == "contend"
    from detectron2_centernet_amd import ops, ops_train as ot
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    B, H, W, Cin, Cout = 16, 128, 128, 64, 64
    x = (torch.randn(B, H, W, Cin, generator=g)).half().to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 24).to(dev)
    w_off = (torch.randn(27, Cin, 3, 3, generator=g) * 0.02).to(dev)
    b_off = (torch.randn(27, generator=g) * 0.3).to(dev)
    dy = torch.randn(B, H, W, Cout, generator=g).half().to(dev)
    p_off = ops.PackedConv(w_off, None, b_off, stride=1, pad=1, compute=ops.F16)
    om = ops.conv2d(x, p_off, out_dtype=torch.float32)
    child = None
    if contend:
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "hammer", "40"])
        time.sleep(15)          # model build + first steps of the other process

    def pieces():
        out = {}
        out["cols"] = ot.dcn_cols(x, om)
        out["dcol"] = ot.dcn_dcol(dy, w, True)
        dx, dom = ot.dcn_col2im_coord(out["dcol"], x, om, dom_channels=32, dcol_chunked=True)
        out["dx"], out["dom"] = dx, dom
        out["dw_dcn"] = ot.conv_wgrad(out["cols"], dy, Cout, 1, 1, 1, 0, scale=1.0)
        out["dw_off"] = ot.conv_wgrad(x, dom, 32, 3, 3, 1, 1, scale=1.0)
        pt = ops.PackedConv(w_off, None, None, stride=1, pad=1, compute=ops.F16, transposed=True, cin_pad=32)
        out["dx_off"] = ops.conv2d(dom, pt, out_dtype=torch.float32)
        p = ops.PackedConv(w, None, None, stride=1, pad=1, compute=ops.F16, cout_align=64)
        out["fwd_dcn"] = ops.dcnv2(x, om, p)
        out["fwd_fused"] = ops.dcnv2_offset(x, p_off, p) if ops.dcnv2_offset_supported(x, p_off, p) else out["fwd_dcn"]
        pc = ops.PackedConv(w, None, None, stride=1, pad=1, compute=ops.F16)
        out["conv3x3"] = ops.conv2d(x, pc)
        out["dgrad3x3"] = ot.conv_dgrad(dy, w, 1, 1, (H, W))
        out["wgrad3x3"] = ot.conv_wgrad(x, dy, Cout, 3, 3, 1, 1, scale=1.0)
        z, mean, invstd, scale = ot.bn_train_fwd(out["conv3x3"], torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev),
                                                 torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev), 1e-5, 0.1)
        out["bn_z"] = z
        out["bn_dy"] = ot.bn_train_bwd(dy, z, out["conv3x3"], mean, invstd, scale, grad_mult=1.0)[0]
        return {k: v.float().clone() for k, v in out.items()}

    ref = pieces()
    torch.cuda.synchronize()
    worst = {k: 0.0 for k in ref}
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    for it in range(n):
        cur = pieces()
        torch.cuda.synchronize()
        for k in ref:
            d = (cur[k] - ref[k]).abs().max().item() / max(1e-30, ref[k].abs().max().item())
            worst[k] = max(worst[k], d)
    print(("WITH a second process on the GPU" if contend else "alone") + f", {n} repetitions: worst relative deviation from the first run")
    for k, v in worst.items():
        print(f"  {k:12s} {v:.3e}")
    if child is not None:
        child.wait()


main()
