"""Probe (round 4): where do the d(offset / mask) outputs of dcn_col2im_window_kernel deviate when a second process shares the
GPU?  (tools/probe_contention.py: dom is the only output that changes; dx, computed by the same kernel, does not.)"""
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from detectron2_centernet_amd import _lib
    if os.environ.get("CTDET_PROBE_LIB"):
        _lib.LIB_PATH = os.environ["CTDET_PROBE_LIB"]
    from detectron2_centernet_amd import ops, ops_train as ot
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    B, H, W, Cin = [int(v) for v in sys.argv[1:5]] if len(sys.argv) > 4 else (16, 128, 128, 64)
    nch = Cin // 32
    f32 = "f32" in sys.argv
    cast = (lambda t: t.float()) if f32 else (lambda t: t.half())
    kw = {"comp": ops.F16X3} if f32 else {}
    x = cast(torch.randn(B, H, W, Cin, generator=g)).to(dev)
    om = torch.randn(B, H, W, 28, generator=g)
    om[..., :18] *= 0.7
    om = om.to(dev)
    dcol = cast(torch.randn(B, H, W, 9 * Cin, generator=g)).to(dev)
    ref_dx, ref_dom = ot.dcn_col2im_coord(dcol, x, om, dom_channels=32, dcol_chunked=True, **kw)
    ref32 = ot.dcn_col2im_coord(dcol, x, om, dcol_chunked=True, **kw)[1]
    with _lib.tuning(_lib.TUNE_NO_COL2IM_WINDOW):
        gen_dom = ot.dcn_col2im_coord(dcol, x, om, dcol_chunked=True, **kw)[1]
    torch.cuda.synchronize()
    print("alone: window f16 dom vs window f32 dom", (ref_dom[..., :27].float() - ref32[..., :27]).abs().max().item(),
          " window vs generic kernel", (ref32 - gen_dom).abs().max().item() / gen_dom.abs().max().item())
    here = os.path.dirname(os.path.abspath(__file__))
    child = None
    if "nochild" not in sys.argv:
        child = subprocess.Popen([sys.executable, os.path.join(here, "probe_contention.py"), "hammer", "45"])
        time.sleep(15)
    stats = {"f16dom": 0, "f32dom": 0, "generic": 0}
    first = None
    for it in range(int(os.environ.get('CTDET_PROBE_ITERS', '60'))):
        dx, dom = ot.dcn_col2im_coord(dcol, x, om, dom_channels=32, dcol_chunked=True, **kw)
        dom32 = ot.dcn_col2im_coord(dcol, x, om, dcol_chunked=True, **kw)[1]
        with _lib.tuning(_lib.TUNE_NO_COL2IM_WINDOW):
            dg = ot.dcn_col2im_coord(dcol, x, om, dcol_chunked=True, **kw)[1]
        torch.cuda.synchronize()
        bad = (dom != ref_dom)
        stats["f16dom"] += int(bad.any())
        stats["f32dom"] += int((dom32 != ref32).any())
        stats["generic"] += int(((dg - gen_dom).abs() > 1e-5 * gen_dom.abs().max()).any())
        assert (dx - ref_dx).abs().max().item() <= 1e-6 * ref_dx.abs().max().item()
        if bad.any() and first is None:
            first = True
            idx = bad.nonzero()
            print(f"iteration {it}: {idx.shape[0]} dom elements differ")
            print("  by channel:", torch.bincount(idx[:, 3], minlength=32).tolist())
            print("  by image:", torch.bincount(idx[:, 0], minlength=B).tolist())
            print("  by row % 8:", torch.bincount(idx[:, 1] % 8, minlength=8).tolist(), " by col % 16:", torch.bincount(idx[:, 2] % 16, minlength=16).tolist())
            tiles = (idx[:, 0] * 100000 + (idx[:, 1] // 8) * 100 + idx[:, 2] // 16)
            print("  distinct tiles:", tiles.unique().numel(), "of", B * (H // 8) * (W // 16))
            b0, y0, x0, c0 = idx[0].tolist()
            print("  first:", idx[0].tolist(), "got", dom[b0, y0, x0].float().tolist(), "\n   want", ref_dom[b0, y0, x0].float().tolist())
            # is the wrong value a partial sum (a chunk missing)?  recompute per chunk
            print("   f32 dom got ", dom32[b0, y0, x0].tolist()[c0 if c0 < 27 else 0], " want ", ref32[b0, y0, x0].tolist()[c0 if c0 < 27 else 0])
    print("iterations with deviations, of 60:", stats)
    if child is not None:
        child.wait()


main()
