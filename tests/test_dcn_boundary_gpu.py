"""The reference-facing DCNv2 boundary (detectron2/layers/deform_conv.py:180-302, 406-519) on the HIP kernels: the functional
`modulated_deform_conv` (forward AND backward, reference signature: NCHW tensors, sigmoid-ed mask), the
`ModulatedDeformConv` module and `DCN.forward`, against
  * the hand-derived known answers G11 (tests/golden/make_dcn_known_answers.py), and
  * the oracle's kernel-by-kernel restatement of the reference backward (oracle.dcnv2_backward,
    deform_conv_cuda_kernel.cu:871-1066 + deform_conv_cuda.cu:929-1129), including the sentinel / truncation branches.
f32 inputs run the reference's own arithmetic (tolerance 1e-5 of scale); f16 inputs the MFMA throughput mode."""
import os

import numpy as np
import pytest
import torch

from oracle import ctdet_oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
G11_CASES = ["zero", "negzero", "int_1_0", "int_0_m2", "int_m1_3", "int_2_2", "half_h", "half_w_masked", "border"]


def _close(got, ref, tol, what):
    err = (got.double().cpu() - ref.double()).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    assert err <= tol * scale, f"{what}: max err {err} (scale {scale})"


@pytest.mark.parametrize("case", G11_CASES)
def test_modulated_deform_conv_known_answers(dev, case):
    from detectron2_centernet_amd.layers.deform_conv import modulated_deform_conv

    g = np.load(os.path.join(GOLD, "g11_dcn_known_answers.npz"))
    x, w, bias, go = (torch.from_numpy(g[k]).float() for k in ("x", "w", "bias", "grad_out"))
    off, mask = torch.from_numpy(g[f"{case}_offset"]).float(), torch.from_numpy(g[f"{case}_mask"]).float()
    leaves = [t.to(dev).requires_grad_(True) for t in (x, off, mask, w, bias)]
    out = modulated_deform_conv(*leaves, stride=1, padding=1, dilation=1)
    assert out.shape == (2, 8, 6, 7) and out.dtype == torch.float32
    _close(out.detach(), torch.from_numpy(g[f"{case}_out"]), 1e-5, f"{case} out")
    out.backward(go.to(dev))
    for key, t in (("grad_input", leaves[0]), ("grad_weight", leaves[3]), ("grad_bias", leaves[4])):
        if f"{case}_{key}" in g.files:
            _close(t.grad, torch.from_numpy(g[f"{case}_{key}"]), 2e-5, f"{case} {key}")
    if f"{case}_grad_offset_h" in g.files:
        _close(leaves[1].grad[:, 0::2], torch.from_numpy(g[f"{case}_grad_offset_h"]), 2e-5, f"{case} grad_offset_h")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float16, 8e-3)])
def test_modulated_deform_conv_backward_matches_restated_kernels(dev, dtype, tol):
    """random geometry incl. samples far outside the image; every gradient the reference's backward produces"""
    from detectron2_centernet_amd.layers.deform_conv import modulated_deform_conv

    gen = torch.Generator().manual_seed(3)
    for B, C, Co, H, W, std in ((2, 32, 24, 9, 11, 1.5), (1, 64, 64, 8, 16, 4.0), (1, 32, 8, 5, 7, 12.0)):
        q = (lambda t: t.half().float()) if dtype == torch.float16 else (lambda t: t)
        x = q(torch.randn(B, C, H, W, generator=gen))
        off = torch.randn(B, 18, H, W, generator=gen) * std
        mask = torch.sigmoid(torch.randn(B, 9, H, W, generator=gen))
        w = q(torch.randn(Co, C, 3, 3, generator=gen) / (C * 9) ** 0.5)
        bias = torch.randn(Co, generator=gen)
        go = q(torch.randn(B, Co, H, W, generator=gen) * 0.1)
        ref_out = O.dcnv2_forward(x.double(), off.double(), mask.double(), w.double(), bias.double(), 1, 1, 1)
        ref = O.dcnv2_backward(x.double(), off.double(), mask.double(), w.double(), go.double())
        leaves = [x.to(dev).to(dtype).requires_grad_(True), off.to(dev).requires_grad_(True),
                  mask.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True), bias.to(dev).requires_grad_(True)]
        out = modulated_deform_conv(*leaves, stride=1, padding=1)
        assert out.dtype == dtype and tuple(out.shape) == (B, Co, H, W)
        _close(out.detach().float(), ref_out, tol, "forward")
        out.backward(go.to(dev).to(dtype))
        for name, t, r in zip(("input", "offset", "mask", "weight", "bias"), leaves, ref):
            _close(t.grad.float(), r, tol * (4 if name in ("offset", "mask") else 1), f"grad_{name} ({B}x{C}->{Co} std {std})")


def test_dcn_backward_sentinel_and_truncation_on_device(dev):
    """the branches of kernel.cu:927-934 / :1027-1029 (see tests/test_oracle_golden.py for the hand values): outside samples
    give exactly zero offset / mask / input gradients, h_im in (-1, 0) scatters (1 + h_im) to row 0 only"""
    from detectron2_centernet_amd.layers.deform_conv import modulated_deform_conv

    C = 32
    x = torch.arange(1, 13, dtype=torch.float32).reshape(1, 1, 3, 4).repeat(1, C, 1, 1)
    w = torch.zeros(4, C, 3, 3)
    w[0, 0, 1, 1] = 2.0
    off = torch.zeros(1, 18, 3, 4)
    mask = torch.ones(1, 9, 3, 4)
    off[0, 8, 0, 0] = -0.25
    off[0, 8, 2, 3] = 5.0
    off[0, 9, 1, 1] = -9.0
    go = torch.zeros(1, 4, 3, 4)
    go[0, 0] = 1.0
    leaves = [t.to(dev).requires_grad_(True) for t in (x, off, mask, w)]
    out = modulated_deform_conv(*leaves, stride=1, padding=1)
    out.backward(go.to(dev))
    o = out.detach().cpu()
    assert o[0, 0, 0, 0].item() == pytest.approx(1.5) and o[0, 0, 2, 3].item() == 0.0 and o[0, 0, 1, 1].item() == 0.0
    gx, goff, gm = leaves[0].grad.cpu(), leaves[1].grad.cpu(), leaves[2].grad.cpu()
    ref = O.dcnv2_backward(x.double(), off.double(), mask.double(), w.double(), go.double(), with_bias=False)
    _close(gx, ref[0], 1e-6, "grad_input")
    _close(goff, ref[1], 1e-6, "grad_offset")
    _close(gm, ref[2], 1e-6, "grad_mask")
    assert goff[0, 8, 2, 3].item() == 0.0 and goff[0, 9, 2, 3].item() == 0.0 and gm[0, 4, 2, 3].item() == 0.0
    assert goff[0, 8, 1, 1].item() == 0.0 and goff[0, 9, 1, 1].item() == 0.0 and gm[0, 4, 1, 1].item() == 0.0
    assert gx[0, 0, 0, 0].item() == pytest.approx(1.5) and gx[0, 0, 2, 3].item() == 0.0 and gx[0, 0, 1, 1].item() == 0.0
    assert goff[0, 8, 0, 0].item() == pytest.approx(2.0)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float16, 8e-3)])
def test_dcn_modules_nchw_forward_backward(dev, dtype, tol):
    """`ModulatedDeformConv` (deform_conv.py:406-495) and the third-party `DCN` wrapper's forward (27-channel
    conv_offset_mask -> offset / sigmoid(mask) -> modulated deformable conv), NCHW in and out, with gradients for the
    parameters and the input"""
    from detectron2_centernet_amd.layers.deform_conv import DCN, ModulatedDeformConv

    gen = torch.Generator().manual_seed(11)
    q = (lambda t: t.half().float()) if dtype == torch.float16 else (lambda t: t)
    B, C, Co, H, W = 2, 32, 16, 8, 10
    x = q(torch.randn(B, C, H, W, generator=gen))
    go = q(torch.randn(B, Co, H, W, generator=gen) * 0.1)
    # --- ModulatedDeformConv: offsets / masks are inputs
    m = ModulatedDeformConv(C, Co, 3, stride=1, padding=1, bias=True).to(dev)
    m.weight.data.copy_(q(m.weight.data.cpu()).to(dev))
    m.bias.data.copy_(torch.randn(Co, generator=gen).to(dev))
    off = torch.randn(B, 18, H, W, generator=gen) * 2
    mask = torch.sigmoid(torch.randn(B, 9, H, W, generator=gen))
    xd = x.to(dev).to(dtype).requires_grad_(True)
    out = m(xd, off.to(dev), mask.to(dev))
    out.backward(go.to(dev).to(dtype))
    wd, bd = m.weight.detach().double().cpu(), m.bias.detach().double().cpu()
    _close(out.detach().float(), O.dcnv2_forward(x.double(), off.double(), mask.double(), wd, bd, 1, 1, 1), tol, "MDC forward")
    ref = O.dcnv2_backward(x.double(), off.double(), mask.double(), wd, go.double())
    _close(xd.grad.float(), ref[0], tol, "MDC grad_input")
    _close(m.weight.grad, ref[3], tol, "MDC grad_weight")
    _close(m.bias.grad, ref[4], tol, "MDC grad_bias")
    # --- DCN: offsets / masks come from its own conv_offset_mask
    d = DCN(C, Co, (3, 3), 1, 1).to(dev)
    d.weight.data.copy_(q(d.weight.data.cpu()).to(dev))
    d.conv_offset_mask.weight.data.copy_(q(torch.randn(27, C, 3, 3, generator=gen) * 0.05).to(dev))
    d.conv_offset_mask.bias.data.copy_((torch.randn(27, generator=gen) * 0.5).to(dev))
    sd = {k: v.detach().double().cpu().requires_grad_(True) for k, v in d.state_dict().items()}
    xr = x.double().requires_grad_(True)
    ref_out = O.dcn_module_forward(xr, sd["conv_offset_mask.weight"], sd["conv_offset_mask.bias"], sd["weight"], sd["bias"])
    ref_out.backward(go.double())
    xd = x.to(dev).to(dtype).requires_grad_(True)
    out = d(xd)
    assert tuple(out.shape) == (B, Co, H, W)
    out.backward(go.to(dev).to(dtype))
    _close(out.detach().float(), ref_out.detach(), tol, "DCN forward")
    _close(xd.grad.float(), xr.grad, 2 * tol, "DCN grad_input")
    _close(d.weight.grad, sd["weight"].grad, tol, "DCN grad_weight")
    _close(d.conv_offset_mask.weight.grad, sd["conv_offset_mask.weight"].grad, 4 * tol, "DCN grad conv_offset_mask.weight")
    _close(d.conv_offset_mask.bias.grad, sd["conv_offset_mask.bias"].grad, 4 * tol, "DCN grad conv_offset_mask.bias")
    # without a gradient request the module runs the fused inference kernel and gives the same values
    with torch.no_grad():
        out2 = d(x.to(dev).to(dtype))
    _close(out2.float(), ref_out.detach(), tol, "DCN forward (no grad)")
