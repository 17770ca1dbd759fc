"""The two f32-tensor modes through the C ABI against torch's fp32 convolution and the DCNv2 oracle: f32 (the reference's
own arithmetic on the f32 matrix pipe: conv_f32_mfma_kernel / dcn_f32_mfma_kernel / the scalar fallback) and f16x3 (the
same kernels' split instantiations: every product as a_hi*w_hi + a_lo*w_hi + a_hi*w_lo on the f16 matrix pipe with f32
accumulation, include/ctdet_hip.h ctdet_conv_desc).  Every test runs in both modes with the SAME tolerance.

Tolerance: 1e-5 of the output scale (f32 sums in a different order; north_star allows 1e-3 on fp32 values).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ctdet_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def ops():
    import detectron2_centernet_amd.ops as ops

    return ops


@pytest.fixture(params=["f32", "f16x3"])
def comp(request, ops):
    return {"f32": ops.F32, "f16x3": ops.F16X3}[request.param]


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def test_conv_f32_random_shapes(ops, dev, comp):
    """every tile of the f32 MFMA kernel (16..128 couts, 128/256-pixel tiles), Cin below / not dividing the 16-k step,
    strides, dilation, ragged maps, residual; Cin % 4 != 0 takes the scalar kernel"""
    rng = np.random.RandomState(2026)
    for it in range(40):
        k = int(rng.choice([1, 3, 3, 7]))
        stride = int(rng.choice([1, 1, 2]))
        dil = int(rng.choice([1, 1, 2])) if k == 3 else 1
        Cin = int(rng.choice([3, 4, 6, 8, 12, 16, 20, 32, 64, 96, 128]))
        Cout = int(rng.choice([2, 4, 16, 27, 32, 40, 64, 80, 128, 132, 256]))
        B = int(rng.randint(1, 4))
        H, W = int(rng.randint(5, 45)), int(rng.randint(5, 45))
        if it % 5 == 0:
            B, H, W = 5, 56, 64     # enough pixel tiles for the 256-pixel variants
        use_res, relu = bool(rng.rand() < 0.4), bool(rng.rand() < 0.6)
        pad = dil * (k // 2)
        g = torch.Generator().manual_seed(3000 + it)
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
        scale = torch.rand(Cout, generator=g) + 0.5
        bias = torch.randn(Cout, generator=g)
        ref = F.conv2d(x, w, None, stride, pad, dil) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
        pc = ops.PackedConv(w.to(dev), scale.to(dev), bias.to(dev), stride=stride, pad=pad, dil=dil, compute=comp)
        res_d = None
        if use_res:
            res = torch.randn(ref.shape, generator=g)
            ref = ref + res
            res_d = torch.zeros(ref.shape[0], ref.shape[2], ref.shape[3], pc.Cout_eff)
            res_d[..., :Cout] = nhwc(res)
            res_d = res_d.to(dev)
        if relu:
            ref = ref.relu()
        y = ops.conv2d(nhwc(x).to(dev), pc, act=ops.ACT_RELU if relu else ops.ACT_NONE, residual=res_d)
        got = nchw(y[..., :Cout].cpu())
        assert got.shape == ref.shape
        err = (got - ref).abs().max().item()
        assert err < TOL * max(1.0, ref.abs().max().item()), \
            f"case {it}: B{B} {H}x{W} {Cin}->{Cout} k{k} s{stride} d{dil} res={use_res}: max err {err}"


@pytest.mark.parametrize("Cin,Cout", [(16, 64), (32, 27), (48, 80), (64, 64), (64, 256), (96, 32), (128, 128), (256, 16)])
def test_f16x3_halo_pair_kernels(ops, dev, Cin, Cout):
    """3x3 / s1 / p1 on tile-divisible maps in the f16x3 mode: the tap-pair kernels.  An even number of 16-channel chunks takes the
    cross-chunk packing (korder 3: tap 8 of a chunk pairs with tap 8 of the next, nine steps per two chunks), an odd one the
    per-chunk packing with a zero tenth tap (korder 2); both 32- and 64-cout tiles, one to sixteen chunks, image borders
    inside and across workgroup tiles, residual + ReLU epilogue"""
    g = torch.Generator().manual_seed(Cin * 1000 + Cout)
    for (B, H, W) in ((1, 8, 32), (2, 16, 64), (3, 24, 32), (3, 16, 16), (2, 32, 48)):      # the last two: 16x16-pixel tiles
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
        scale, bias = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
        res = torch.randn(B, Cout, H, W, generator=g)
        ref = (F.conv2d(x, w, None, 1, 1) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1) + res).relu()
        pc = ops.PackedConv(w.to(dev), scale.to(dev), bias.to(dev), stride=1, pad=1, compute=ops.F16X3)
        res_d = torch.zeros(B, H, W, pc.Cout_eff)
        res_d[..., :Cout] = nhwc(res)
        y = ops.conv2d(nhwc(x).to(dev), pc, act=ops.ACT_RELU, residual=res_d.to(dev))
        if Cin == 16 and W % 64 == 0:       # a contiguous 16-channel input on a 64-divisible map is the LDS-window kernel's
            assert pc.w_pair is None
        elif W % 32 != 0 and (Cin // 16) % 2 != 0:      # 16x16 tiles exist for the cross-chunk packing only: uniform-K kernel
            assert pc.w_pair is None
        else:
            assert pc.w_pair is not None and pc.pair_korder == (3 if (Cin // 16) % 2 == 0 else 2)
        err = (nchw(y[..., :Cout].cpu()) - ref).abs().max().item()
        assert err < TOL * max(1.0, ref.abs().max().item()), f"B{B} {H}x{W} {Cin}->{Cout}: max err {err}"


def test_conv_f32_slices_and_cat(ops, dev, comp):
    """channel-slice input / output views and Root's multi-source 1x1 (dla.py:86-94) in f32"""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 96, 12, 20, generator=g)
    w = torch.randn(32, 64, 3, 3, generator=g) / 24
    ref = F.conv2d(x[:, 16:80], w, None, 1, 1)
    pc = ops.PackedConv(w.to(dev), None, None, stride=1, pad=1, compute=comp)
    out = torch.zeros(2, 12, 20, 48, device=dev)
    ops.conv2d(nhwc(x).to(dev)[..., 16:80], pc, out=out[..., 8:40])
    assert (nchw(out[..., 8:40].cpu()) - ref).abs().max() < TOL * ref.abs().max()
    assert out[..., :8].abs().max() == 0 and out[..., 40:].abs().max() == 0
    rng = np.random.RandomState(9)
    for it in range(8):
        n = int(rng.randint(1, 5))
        cins = [int(rng.choice([4, 8, 12, 16, 32, 64])) for _ in range(n)]
        Cout = int(rng.choice([16, 40, 64, 128]))
        B, H, W = int(rng.randint(1, 3)), int(rng.randint(3, 24)), int(rng.randint(3, 24))
        xs = [torch.randn(B, c, H, W, generator=g) for c in cins]
        w = torch.randn(Cout, sum(cins), 1, 1, generator=g) / sum(cins) ** 0.5
        ref = F.conv2d(torch.cat(xs, 1), w).relu()
        pc = ops.PackedConv(w.to(dev), None, None, compute=comp)
        y = ops.conv1x1_cat([nhwc(t).to(dev) for t in xs], pc, act=ops.ACT_RELU)
        err = (nchw(y[..., :Cout].cpu()) - ref).abs().max().item()
        assert err < TOL * max(1.0, ref.abs().max().item()), f"cat case {it}: {cins}->{Cout}: {err}"


def test_conv_transpose_f32(ops, dev, comp):
    """dense ConvTranspose2d 4x4 s2 p1 (centernet.py:268-293) in f32: the input is read as zero-stuffed in place"""
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 32, 9, 11, generator=g)
    w = torch.randn(32, 24, 4, 4, generator=g) / 16
    ref = F.conv_transpose2d(x, w, None, stride=2, padding=1)
    y = ops.conv_transpose2d(nhwc(x).to(dev), w.to(dev), None, None, 2, 1, comp)
    assert (nchw(y[..., :24].cpu()) - ref).abs().max() < TOL * ref.abs().max()


def test_dcnv2_f32_random_shapes(ops, dev, comp):
    """DCNv2 on the f32 matrix pipe vs the oracle (deform_conv_cuda_kernel.cu:666-868 restated): partial tiles, every cout
    tile, offsets from zero to far outside the image, exact-integer and border coordinates; Cin % 16 != 0 -> scalar kernel"""
    rng = np.random.RandomState(77)
    for it in range(14):
        Cin = int(rng.choice([16, 32, 64, 128, 24]))
        Cout = int(rng.choice([8, 16, 40, 64, 128, 132]))
        B, H, W = int(rng.randint(1, 3)), int(rng.randint(3, 30)), int(rng.randint(3, 40))
        if it % 6 == 0:
            B, H, W = 6, 48, 64
        off_std = float(rng.choice([0.0, 0.7, 2.0, 5.0, 30.0]))
        g = torch.Generator().manual_seed(700 + it)
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
        om = torch.randn(B, 27, H, W, generator=g)
        om[:, :18] *= off_std
        om[:, 0, 0, 0] = -0.0
        om[:, 1, 0, 0] = -1.0
        om[:, 2, -1, -1] = 1.0
        om[:, 3, 1, 1] = 0.5
        bias = torch.randn(Cout, generator=g)
        ref = O.dcnv2_forward(x, om[:, :18], torch.sigmoid(om[:, 18:]), w, bias, 1, 1, 1).relu()
        pc = ops.PackedConv(w.to(dev), None, bias.to(dev), stride=1, pad=1, compute=comp)
        om_d = torch.zeros(B, H, W, 28)
        om_d[..., :27] = nhwc(om)
        y = ops.dcnv2(nhwc(x).to(dev), om_d.to(dev), pc, act=ops.ACT_RELU)
        err = (nchw(y[..., :Cout].cpu()) - ref).abs().max().item()
        assert err <= 2e-5 * max(1.0, ref.abs().max().item()), f"case {it}: B{B} {H}x{W} {Cin}->{Cout} std {off_std}: {err}"


@pytest.mark.parametrize("case", [(7, 8, 16, 1, 3, 2, 32, 128), (7, 8, 16, 1, 0, 1, 16, 64), (3, 16, 16, 1, 1, 3, 24, 64),
                                  (3, 16, 32, 2, 1, 2, 32, 128)])
def test_conv_f32_window_kernel(ops, dev, case, comp):
    """conv_f32_win_kernel (the narrow DLA base layers in f32): stem 7x7 on the 8-channel image with its padding in the
    kernel (pad 3) or as a zero frame in memory (pad 0), level0 3x3 16->16, level1 3x3 16->32 stride 2 -- vs torch"""
    k, Cin, Cout, stride, pad, B, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g)
    if Cin == 8:
        x[:, 3:] = 0
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    scale, bias = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    ref = (F.conv2d(x, w, None, stride, k // 2) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)).relu()
    pc = ops.PackedConv(w.to(dev), scale.to(dev), bias.to(dev), stride=stride, pad=pad, compute=comp)
    xin = nhwc(x)
    if pad == 0 and k > 1:      # zero frame in memory
        xin = F.pad(xin, (0, 0, k // 2, k // 2, k // 2, k // 2))
    y = ops.conv2d(xin.to(dev), pc, act=ops.ACT_RELU)
    got = nchw(y[..., :Cout].cpu())
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < TOL * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("case", [(1, 8, 16, 16, 64, 0.0), (2, 16, 32, 32, 64, 0.7), (1, 24, 48, 64, 128, 2.0), (2, 8, 32, 16, 40, 5.0),
                                  (1, 16, 16, 128, 132, 30.0), (3, 40, 48, 64, 64, 3.0)])
def test_dcnv2_f32_window_kernel(ops, dev, case, comp):
    """dcn_f32_window_kernel (8x16 tiles sampling an 18x26 LDS window; far samples from global memory) vs the oracle, and
    bit-for-bit against nothing less: the gather kernel (TUNE_NO_F32_DCN_WINDOW) must agree with it to f32 rounding"""
    from detectron2_centernet_amd import _lib

    B, H, W, Cin, Cout, off_std = case
    g = torch.Generator().manual_seed(int(sum(case[:5])))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    om = torch.randn(B, 27, H, W, generator=g)
    om[:, :18] *= off_std
    om[:, 0, 0, 0], om[:, 1, 0, 0], om[:, 2, -1, -1], om[:, 3, 1, 1] = -0.0, -1.0, 1.0, 0.5
    om[:, 4, 2, 2], om[:, 5, 2, 2] = -4.0, 4.0        # exactly the window margin
    om[:, 6, 3, 3], om[:, 7, 3, 3] = -5.25, 5.5       # just beyond it
    om[:, 8, 0, 5], om[:, 9, 0, 5] = -float(H), float(W) + 3   # outside the image: contributes 0
    bias = torch.randn(Cout, generator=g)
    ref = O.dcnv2_forward(x, om[:, :18], torch.sigmoid(om[:, 18:]), w, bias, 1, 1, 1).relu()
    pc = ops.PackedConv(w.to(dev), None, bias.to(dev), stride=1, pad=1, compute=comp)
    om_d = torch.zeros(B, H, W, 28)
    om_d[..., :27] = nhwc(om)
    xd, omd = nhwc(x).to(dev), om_d.to(dev)
    y = ops.dcnv2(xd, omd, pc, act=ops.ACT_RELU)
    with _lib.tuning(_lib.TUNE_NO_F32_DCN_WINDOW):
        y_gather = ops.dcnv2(xd, omd, pc, act=ops.ACT_RELU)
    scale = max(1.0, ref.abs().max().item())
    err = (nchw(y[..., :Cout].cpu()) - ref).abs().max().item()
    assert err <= 2e-5 * scale, f"{case}: window kernel vs oracle {err}"
    assert (y - y_gather).abs().max().item() <= 2e-5 * scale
    assert y[..., Cout:].abs().max().item() == 0 if y.shape[-1] > Cout else True


def test_f16x3_small_and_large_magnitudes(ops, dev):
    """operand scales from 1e-4 to 1e3.  Weight rows are scaled by a power of two at pack time, so their magnitude does not
    matter; a small ACTIVATION's lo half reaches the f16 subnormals (spacing 6e-8), i.e. an absolute floor of 3e-8 per
    element on top of the 2^-22 relative accuracy of the split: error measured against sum |x||w| in f64"""
    g = torch.Generator().manual_seed(11)
    for sx, sw in [(1e-4, 1.0), (1.0, 1e-4), (1e-3, 1e-3), (1.0, 1.0), (1e3, 1.0), (30.0, 30.0)]:
        x = torch.randn(2, 64, 16, 32, generator=g) * sx
        w = torch.randn(64, 64, 3, 3, generator=g) * sw
        ref = F.conv2d(x.double(), w.double(), None, 1, 1)
        mag = F.conv2d(x.double().abs(), w.double().abs(), None, 1, 1)
        pc = ops.PackedConv(w.to(dev), None, None, stride=1, pad=1, compute=ops.F16X3)
        y = nchw(ops.conv2d(nhwc(x).to(dev), pc).cpu()).double()
        rel = ((y - ref).abs() / mag).max().item()
        assert rel < 5e-7 + 4e-8 / sx, f"scales {sx} {sw}: relative error {rel}"
