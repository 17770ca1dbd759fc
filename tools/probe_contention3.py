"""Probe (round 4): per-(pixel, tap, chunk, lane group) corner dots of dcn_col2im_window_kernel under GPU sharing, from a debug
build of the library (tools/_libctdet_dbg.so: -DCTDET_DEBUG_COL2IM)."""
import ctypes as C
import os
import subprocess
import sys
import time

import torch

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
from detectron2_centernet_amd import _lib
_lib.LIB_PATH = os.path.join(here, "_libctdet_dbg.so")
from detectron2_centernet_amd import ops, ops_train as ot


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    B, H, W, Cin = 16, 128, 128, 64
    x = (torch.randn(B, H, W, Cin, generator=g)).half().to(dev)
    om = torch.randn(B, H, W, 28, generator=g)
    om[..., :18] *= 0.7
    om = om.to(dev)
    dcol = torch.randn(B, H, W, 9 * Cin, generator=g).half().to(dev)
    lib = _lib.lib()
    dbg = torch.zeros(B * H * W, 9, 2, 4, 8, device=dev)
    lib.ctdet_debug_col2im_buffer.argtypes = [C.c_void_p]
    assert lib.ctdet_debug_col2im_buffer(C.c_void_p(dbg.data_ptr())) == 0
    _, ref_dom = ot.dcn_col2im_coord(dcol, x, om, dcol_chunked=True)
    torch.cuda.synchronize()
    ref_dbg = dbg.clone()
    child = subprocess.Popen([sys.executable, os.path.join(here, "probe_contention.py"), "hammer", "45"])
    time.sleep(15)
    shown = 0
    for it in range(60):
        dbg.zero_()
        _, dom = ot.dcn_col2im_coord(dcol, x, om, dcol_chunked=True)
        torch.cuda.synchronize()
        bad = (dom != ref_dom)
        dbad = (dbg != ref_dbg)
        if bad.any() or dbad.any():
            print(f"iteration {it}: {int(bad.sum())} dom elements differ, {int(dbad.sum())} debug words differ")
            if dbad.any() and shown < 3:
                shown += 1
                idx = dbad.nonzero()
                print("  debug words by field (sq0..3, hw, lw, off, inwin):", torch.bincount(idx[:, 4], minlength=8).tolist())
                print("  by tap:", torch.bincount(idx[:, 1], minlength=9).tolist(), " by chunk:", torch.bincount(idx[:, 2], minlength=2).tolist(),
                      " by lane group q:", torch.bincount(idx[:, 3], minlength=4).tolist())
                for r in idx[:6].tolist():
                    m, t, ch, q, f = r
                    print(f"   pixel {m} (img {m // (H * W)}, y {(m // W) % H}, x {m % W}) tap {t} chunk {ch} q {q}: got {dbg[m, t, ch, q].tolist()}\n      want {ref_dbg[m, t, ch, q].tolist()}")
            if bad.any() and not dbad.any():
                print("  (the dots agree: the difference arises behind them)")
    child.wait()


main()
