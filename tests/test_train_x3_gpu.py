"""GPU parity of the f16x3 TRAINING mode (round 4): f32 tensors, statistics and gradients, every contraction of the step --
forward, input gradient, weight gradient, DCNv2's column GEMMs -- as hi*hi + lo*hi + hi*lo on the f16 matrix pipe with f32
accumulation.  The kernels are checked one by one against torch's fp32 ops on f32 (not f16-representable) data at f32-grade
tolerances, then the whole step against torch autograd through the fp32 oracle at the bounds of the f32 mode's test
(losses 1e-3 -- north_star --, every gradient group at cosine >= 0.999 with matching norms)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import ctdet_oracle as O
from oracle import model_ref as MR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import detectron2_centernet_amd.ops as ops
    import detectron2_centernet_amd.ops_train as ot

    return ops, ot


@pytest.fixture()
def x3(T):
    """the f32-tensor nodes contract in f16x3 inside the test, whatever ran before"""
    ops, ot = T
    prev = ot.F32_COMPUTE
    ot.F32_COMPUTE = ops.F16X3
    yield ops.F16X3
    ot.F32_COMPUTE = prev


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def close(got, ref, tol, what=""):
    err = (got - ref).abs().max().item()
    scale = max(1e-30, ref.abs().max().item())
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.2e})"


# ------------------------------------------------------------------------------------------ weight packing
@pytest.mark.parametrize("case", [(64, 64, 3, False, None), (27, 64, 3, False, None), (128, 48, 3, False, None), (16, 3, 7, False, 8),
                                  (256, 64, 1, False, None), (64, 32, 3, True, None), (2, 256, 1, True, 8), (80, 256, 1, False, None),
                                  (64, 128, 3, True, 72), (40, 16, 3, False, None)])
def test_pack_weights_x3_equals_the_torch_chain(T, dev, case):
    """ctdet_pack_weights_x3 (one launch per layout: row scale, split, tap-pair interleave) produces bit for bit what the
    torch chain of round 3 produced (permute / pad / row maximum / power-of-two scale / ctdet_split_weights / _pack_pairs): the
    tap-major split image, the tap-pair images (korder 2 and 3) and the epilogue scale; forward and input-gradient operands"""
    ops, ot = T
    O_, I, k, transposed, cin_pad = case
    g = torch.Generator().manual_seed(sum(c for c in case if isinstance(c, int) and not isinstance(c, bool)))
    w = (torch.randn(O_, I, k, k, generator=g) * torch.rand(O_, 1, 1, 1, generator=g) * 0.3).to(dev)
    w[min(3, O_ - 1)] = 0            # an all-zero row: scale 1, zeros
    pad = k // 2
    if transposed:
        wt = w.flip(2, 3).permute(1, 0, 2, 3).contiguous()
        ref = ops.PackedConv(wt.double(), None, None, stride=1, pad=pad, compute=ops.F16X3, cin_pad=cin_pad)    # f64 in: the torch chain
        got = ops.PackedConv(w, None, None, stride=1, pad=pad, compute=ops.F16X3, cin_pad=cin_pad, transposed=True)
    else:
        ref = ops.PackedConv(w.double(), None, None, stride=1, pad=pad, compute=ops.F16X3, cin_pad=cin_pad)
        got = ops.PackedConv(w, None, None, stride=1, pad=pad, compute=ops.F16X3, cin_pad=cin_pad)
    assert got._x3_src is not None and ref._x3_src is None
    assert (got.Cout_pad, got.Kpad, got.Cout_eff, got.Cin) == (ref.Cout_pad, ref.Kpad, ref.Cout_eff, ref.Cin)
    assert torch.equal(got.w.view(torch.int32), ref.w.view(torch.int32)), "split tap-major image"
    assert torch.equal(got.scale, ref.scale), "epilogue scale"
    if k == 3 and got.Cin % 16 == 0:
        x = torch.zeros(1, 16, 32, got.Cin, device=dev)
        assert got.pair_ok(x) and ref.pair_ok(x)
        assert got.pair_korder == ref.pair_korder == (3 if (got.Cin // 16) % 2 == 0 else 2)
        assert torch.equal(got.w_pair.view(torch.int32), ref.w_pair.view(torch.int32)), "tap-pair image"
        assert torch.equal(got.scale_pair, ref.scale)


def test_pack_weights_x3_dcn_column_operand(T, dev):
    """the d(columns) operand of DCNv2's backward (rows = 9*Cin column channels, tap-major or chunk-major; channels = the couts
    of dY, zero-padded) straight from the [Cout, Cin, 3, 3] parameter against the permuted weight matrix through the torch chain"""
    ops, ot = T
    g = torch.Generator().manual_seed(11)
    w = (torch.randn(48, 64, 3, 3, generator=g) * 0.05).to(dev)
    for chunked in (False, True):
        wm = ot.dcn_weight_matrix(w, 48, chunked)                               # [48, 9*64, 1, 1]: d(columns) = dY . wm
        wt = wm.flip(2, 3).permute(1, 0, 2, 3).contiguous()                     # the conv that maps dY to d(columns)
        ref = ops.PackedConv(wt.double(), None, None, stride=1, pad=0, compute=ops.F16X3)
        got = ops.PackedConv(w, None, None, stride=1, pad=0, compute=ops.F16X3, cin_pad=48,
                             transposed="dcn_cols_chunked" if chunked else "dcn_cols")
        assert (got.Cout_pad, got.Kpad, got.Cout_eff) == (ref.Cout_pad, ref.Kpad, ref.Cout_eff)
        assert torch.equal(got.w.view(torch.int32), ref.w.view(torch.int32)) and torch.equal(got.scale, ref.scale)


# ------------------------------------------------------------------------------------------ pointwise f32 kernels
@pytest.mark.parametrize("C,res,relu", [(16, False, True), (64, True, True), (128, False, False), (80, False, True), (28, False, False),
                                        (512, True, True)])
def test_bn_train_fwd_bwd_f32(T, dev, C, res, relu):
    """the BatchNorm kernels instantiated for f32 tensors (16-byte vectors of 4 channels; power-of-two and other channel-vector
    counts) against torch's batch_norm autograd on f32 data"""
    ops, ot = T
    g = torch.Generator().manual_seed(C)
    y = (torch.randn(3, C, 10, 12, generator=g) * 2 + 0.5).requires_grad_(True)
    r = torch.randn(3, C, 10, 12, generator=g).requires_grad_(True) if res else None
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(C, generator=g).requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    ref = F.batch_norm(y, rm, rv, gamma, beta, True, 0.1, 1e-5)
    if res:
        ref = ref + r
    if relu:
        ref = ref.relu()
    dz = torch.randn(ref.shape, generator=g)
    ref.backward(dz)
    rm_d, rv_d = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    z, mean, invstd, scale = ot.bn_train_fwd(nhwc(y.detach()).to(dev), gamma.detach().to(dev), beta.detach().to(dev),
                                             rm_d, rv_d, 1e-5, 0.1, res=nhwc(r.detach()).to(dev) if res else None, relu=relu)
    assert z.dtype == torch.float32
    close(nchw(z.cpu()), ref.detach(), 2e-6, "bn fwd")
    close(rm_d.cpu(), rm, 1e-5, "running_mean")
    close(rv_d.cpu(), rv, 1e-5, "running_var")
    dy, dres, dgamma, dbeta = ot.bn_train_bwd(nhwc(dz).to(dev), z, nhwc(y.detach()).to(dev), mean, invstd, scale, relu=relu,
                                              want_dres=res, grad_mult=1.0)
    close(nchw(dy.cpu()), y.grad, 2e-5, "bn dy")
    close(dgamma.cpu(), gamma.grad, 2e-5, "dgamma")
    close(dbeta.cpu(), beta.grad, 2e-5, "dbeta")
    if res:
        close(nchw(dres.cpu()), r.grad, 1e-6, "dres")


def test_maxpool_dwconvT_depth_to_space_f32(T, dev):
    ops, ot = T
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 8, 12, generator=g).requires_grad_(True)
    y = F.max_pool2d(x, 2, 2)
    dz = torch.randn(y.shape, generator=g)
    y.backward(dz)
    dx = ot.maxpool2x2_bwd(nhwc(x.detach()).to(dev), nhwc(dz).to(dev))
    assert dx.dtype == torch.float32 and torch.equal(nchw(dx.cpu()), x.grad)
    for f in (2, 4):
        x = torch.randn(2, 36, 6, 5, generator=g).requires_grad_(True)
        w = torch.rand(36, 1, 2 * f, 2 * f, generator=g).requires_grad_(True)
        y = F.conv_transpose2d(x, w, None, stride=f, padding=f // 2, groups=36)
        dz = torch.randn(y.shape, generator=g)
        y.backward(dz)
        dx, dw = ot.dwconvT_bwd(nhwc(x.detach()).to(dev), nhwc(dz).to(dev), w.detach().to(dev), f)
        close(nchw(dx.cpu()), x.grad, 2e-6, "dwconvT dx")
        close(dw.cpu(), w.grad, 1e-5, "dwconvT dw")


# ------------------------------------------------------------------------------------------ weight / input gradients
WG_CASES = [(2, 12, 14, 64, 64, 3, 1, 1), (1, 16, 16, 32, 64, 3, 2, 1), (2, 9, 9, 128, 128, 3, 1, 1), (1, 8, 8, 256, 64, 1, 1, 0),
            (2, 20, 20, 16, 32, 3, 2, 1), (1, 16, 24, 8, 16, 7, 1, 3), (2, 6, 6, 576, 64, 1, 1, 0),
            # window form (3x3 / stride 1, maps divisible by 8x32, Cin % 32 == 0); Cout 24 / 40: partial cout tiles
            (2, 8, 32, 64, 64, 3, 1, 1), (1, 16, 64, 32, 24, 3, 1, 1), (3, 24, 32, 96, 40, 3, 1, 1),
            # narrow window form: the 7x7 stem on 8 channels, level0's 3x3 16 -> 16
            (1, 8, 32, 8, 16, 7, 1, 3), (2, 24, 64, 8, 16, 7, 1, 3), (2, 16, 64, 16, 16, 3, 1, 1), (3, 8, 32, 16, 8, 3, 1, 1),
            (1, 40, 96, 32, 32, 3, 1, 1), (1, 40, 96, 8, 16, 7, 1, 3), (1, 40, 96, 16, 16, 3, 1, 1)]


@pytest.mark.parametrize("case", WG_CASES)
def test_conv_wgrad_and_dgrad_x3(T, dev, x3, case):
    """dW (generic, window and narrow kernels: f32 operands split into hi + lo on the way to LDS, three MFMAs per product) and dX
    (the f16x3 forward kernels on the transposed / flipped operand packed from the parameter; stride 2 through the four-phase
    form) against torch's fp32 conv autograd on f32 data.  Bound: 2e-5 of the largest element -- two orders of magnitude
    below what one f16 product per term gives (2e-3)"""
    ops, ot = T
    B, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    dw = ot.conv_wgrad(nhwc(x.detach()).to(dev), nhwc(dy).to(dev), Cout, k, k, s, p, scale=1.0, comp=x3)
    got = dw.view(Cout, k, k, Cin).permute(0, 3, 1, 2).cpu()
    close(got, w.grad, 2e-5, "dW")
    if Cin % 4 == 0 and Cout % 8 == 0:
        dx = ot.conv_dgrad(nhwc(dy).to(dev), w.detach().to(dev), s, p, (H, W), comp=x3)
        assert dx.dtype == torch.float32
        close(nchw(dx.cpu()), x.grad, 2e-5, "dX")


@pytest.mark.parametrize("mag", [1e-3, 1.0, 30.0])
def test_conv_wgrad_x3_magnitudes(T, dev, x3, mag):
    """gradient magnitudes from 1e-3 to 30 (what reaches the kernels is the gradient times the step's loss scale of 1024): 1e-5
    of the largest element while the lo halves are normal f16 numbers (|dy| >= 0.125); below that lo runs into the f16
    subnormals (quantum 6e-8: at |dy| ~ 1e-3 an element keeps 14 bits) and the bound is 5e-5 -- still 40 times tighter than one
    f16 product per term.  The window kernel on a BASELINE-shaped layer"""
    ops, ot = T
    g = torch.Generator().manual_seed(3)
    B, H, W, Cin, Cout = 2, 16, 64, 64, 64
    x = torch.randn(B, Cin, H, W, generator=g).relu() * 1.5
    dy = torch.randn(B, Cout, H, W, generator=g) * mag
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 3, 3), dy.double(), stride=1, padding=1).float()
    dw = ot.conv_wgrad(nhwc(x).to(dev), nhwc(dy).to(dev), Cout, 3, 3, 1, 1, scale=1.0, comp=x3)
    close(dw.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2).cpu(), ref, 1e-5 if mag >= 1.0 else 5e-5, f"dW at |dy| ~ {mag}")


def test_conv_wgrad_x3_into_parameter_layout_and_slices(T, dev, x3):
    """the f16x3 weight-gradient kernels accumulating straight into an OIHW gradient with padded channels dropped, and reading
    x / dy as channel slices of wider NHWC buffers"""
    ops, ot = T
    g = torch.Generator().manual_seed(17)
    for (B, H, W, Cin, Cout, k, p) in [(2, 16, 32, 64, 48, 3, 1), (2, 16, 32, 8, 16, 7, 3), (1, 8, 8, 256, 80, 1, 0)]:
        cin_real = 3 if Cin == 8 else Cin
        x = torch.randn(B, Cin, H, W, generator=g)
        x[:, cin_real:] = 0
        dy = torch.randn(B, Cout, H, W, generator=g)
        ref = torch.nn.grad.conv2d_weight(x[:, :cin_real], (Cout, cin_real, k, k), dy, stride=1, padding=p)
        Cw = (Cout + 7) // 8 * 8
        xw = torch.randn(B, H, W, Cin + 16, generator=g).to(dev)
        dyw = torch.zeros(B, H, W, Cw + 8).to(dev)
        xw[..., 8:8 + Cin] = nhwc(x).to(dev)
        dyw[..., :Cout] = nhwc(dy).to(dev)
        prior = torch.randn(Cout, cin_real, k, k, generator=g)
        slot = prior.clone().to(dev)
        ot.conv_wgrad(xw[..., 8:8 + Cin], dyw[..., :Cw], Cw, k, k, 1, p, scale=0.5, into=(slot, k * k, Cin), comp=x3)
        close(slot.cpu() - prior, 0.5 * ref, 3e-5, f"dW {Cin}->{Cout} k{k} in the parameter's layout, from slices")


def test_conv_bias_relu_fn_x3(T, dev, x3):
    """the head pattern (3x3 + bias + ReLU, then 1x1 + bias to 2 channels) as ConvFn nodes in the f16x3 mode"""
    ops, ot = T
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 64, 16, 32, generator=g).requires_grad_(True)
    w = (torch.randn(256, 64, 3, 3, generator=g) / 24).requires_grad_(True)
    b = torch.randn(256, generator=g).requires_grad_(True)
    w2 = (torch.randn(2, 256, 1, 1, generator=g) / 16).requires_grad_(True)
    b2 = torch.randn(2, generator=g).requires_grad_(True)
    out = F.conv2d(F.conv2d(x, w, b, 1, 1).relu(), w2, b2)
    dz = torch.randn(out.shape, generator=g) * 1e-2
    out.backward(dz)
    leaves = [t.detach().to(dev).requires_grad_(True) for t in (w, b, w2, b2)]
    xd = nhwc(x.detach()).to(dev).requires_grad_(True)
    hid = ot.ConvFn.apply(xd, leaves[0], leaves[1], 1, 1, True, False)
    o = ot.ConvFn.apply(hid, leaves[2], leaves[3], 1, 0, False, True)
    close(nchw(o[..., :2].cpu()), out.detach(), 1e-5, "fwd")
    o.backward(torch.nn.functional.pad(nhwc(dz), (0, 2)).to(dev) * ot.GRAD_SCALE)
    for got, ref, name in zip(leaves, (w, b, w2, b2), ("w", "b", "w2", "b2")):
        close(got.grad.cpu(), ref.grad, 3e-5, name)
    close(nchw(xd.grad.cpu()) / ot.GRAD_SCALE, x.grad, 3e-5, "dx")


# ------------------------------------------------------------------------------------------ DCNv2
@pytest.mark.parametrize("case", [(2, 10, 12, 64, 64, 2.0), (1, 8, 8, 128, 64, 1.0), (2, 16, 32, 64, 64, 1.5), (1, 8, 16, 128, 64, 7.0),
                                  (1, 16, 16, 64, 128, 0.0)])
def test_dcn_training_fwd_bwd_x3(T, dev, x3, case):
    """DCNv2 forward and backward (columns, both column GEMMs, the LDS-window scatter with f32 columns) in the f16x3 mode against
    the oracle's restatement on f32 data.  The scatter accumulates in per-tile fixed point (2^-20 of the tile's largest
    d(columns) magnitude per contribution), hence 1e-4 on d(input)"""
    ops, ot = T
    B, H, W, Cin, Cout, off_std = case
    g = torch.Generator().manual_seed(Cin + H)
    x = torch.randn(B, Cin, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).requires_grad_(True)
    bias = torch.randn(Cout, generator=g).requires_grad_(True)
    om = torch.randn(B, 27, H, W, generator=g)
    om[:, :18] *= off_std
    om.requires_grad_(True)
    y = O.dcnv2_forward(x, om[:, :18], torch.sigmoid(om[:, 18:]), w, bias, 1, 1, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xd = nhwc(x.detach()).to(dev).requires_grad_(True)
    omd = torch.zeros(B, H, W, 28)
    omd[..., :27] = nhwc(om.detach())
    omd = omd.to(dev).requires_grad_(True)
    wd = w.detach().to(dev).requires_grad_(True)
    bd = bias.detach().to(dev).requires_grad_(True)
    yd = ot.DCNFn.apply(xd, omd, wd, bd)
    close(nchw(yd.cpu()), y.detach(), 2e-5, "dcn fwd")
    yd.backward(nhwc(dy).to(dev) * ot.GRAD_SCALE)
    S = ot.GRAD_SCALE
    close(nchw(xd.grad.cpu()) / S, x.grad, 1e-4, "dcn dx")
    close(nchw(omd.grad[..., :27].cpu()) / S, om.grad, 1e-4, "dcn d(offset, mask)")
    close(wd.grad.cpu(), w.grad, 3e-5, "dcn dW")
    close(bd.grad.cpu(), bias.grad, 2e-5, "dcn dbias")


@pytest.mark.parametrize("shape", [(2, 16, 32, 64), (12, 32, 32, 32), (40, 32, 48, 32)])
def test_dcn_col2im_window_f32_vs_atomics(T, dev, x3, shape):
    """the LDS-window scatter on f32 columns / inputs (tap split over 3 / 2 / 1 workgroups per tile) against the generic f32
    kernel with plain atomics, tap-major and chunk-major columns"""
    ops, ot = T
    B, H, W, Cin = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, H, W, Cin, generator=g).to(dev)
    dcol = torch.randn(B, H, W, 9 * Cin, generator=g).to(dev)
    om = torch.randn(B, H, W, 28, generator=g)
    om[..., :18] *= 2.0
    om = om.to(dev)
    dx_w, dom_w = ot.dcn_col2im_coord(dcol, x, om, comp=x3)
    dx_a, dom_a = ot.dcn_col2im_coord(dcol, x, om, comp=ops.F32)
    mx = dx_a.abs().max().item()
    assert (dx_w - dx_a).abs().max().item() <= mx * 2.0 ** -16
    assert (dom_w - dom_a).abs().max().item() <= dom_a.abs().max().item() * 1e-5
    dcol_c = dcol.view(B, H, W, 9, Cin // 32, 32).permute(0, 1, 2, 4, 3, 5).reshape(B, H, W, 9 * Cin).contiguous()
    dx_c, dom_c = ot.dcn_col2im_coord(dcol_c, x, om, dcol_chunked=True, comp=x3)
    assert (dx_c - dx_w).abs().max().item() <= mx * 1e-6 and torch.equal(dom_c, dom_w)
    _, dom_p = ot.dcn_col2im_coord(dcol, x, om, dom_channels=32, comp=x3)
    assert dom_p.dtype == torch.float32 and dom_p.shape[3] == 32 and dom_p[..., 27:].abs().max().item() == 0
    assert torch.equal(dom_p[..., :27], dom_w[..., :27])


@pytest.mark.parametrize("shape", [(2, 16, 32, 64, 64), (12, 32, 32, 32, 256), (40, 32, 48, 64, 128), (1, 16, 16, 512, 256)])
def test_dcn_col2im_fused_dcol_equals_two_step(T, dev, x3, shape):
    """ctdet_dcn_col2im_fused (d(columns) = dY . W computed per tile on the matrix pipe inside the scatter kernel; tap split over
    3 / 2 / 1 workgroups) against the two-step path (f16x3 1x1 GEMM writing d(columns), then the scatter kernel reading it):
    same products, another summation order"""
    ops, ot = T
    B, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, H, W, Cin, generator=g).to(dev)
    dy = torch.randn(B, H, W, Cout, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * torch.rand(Cout, 1, 1, 1, generator=g) / (Cin * 9) ** 0.5).to(dev)
    om = torch.randn(B, H, W, 28, generator=g)
    om[..., :18] *= 2.0
    om = om.to(dev)
    fused = ot.dcn_col2im_fused(dy, w, x, om, dom_channels=32)       # (called directly: the training step's default is two-step)
    assert fused is not None
    dx_f, dom_f = fused
    dcol = ot.dcn_dcol(dy, w, True, x3)
    dx_t, dom_t = ot.dcn_col2im_coord(dcol, x, om, dom_channels=32, dcol_chunked=True, comp=x3)
    close(dx_f.cpu(), dx_t.cpu(), 2e-5, "dx")
    close(dom_f.cpu(), dom_t.cpu(), 2e-5, "dom")
    assert dom_f[..., 27:].abs().max().item() == 0
    # and against the exact f64 product for d(columns) feeding the generic f32 kernel
    wm = w.double().reshape(Cout, Cin // 32, 32, 9).permute(0, 1, 3, 2).reshape(Cout, 9 * Cin)
    dcol64 = (dy.double().reshape(-1, Cout) @ wm).float().reshape(B, H, W, 9 * Cin)
    dx_r, dom_r = ot.dcn_col2im_coord(dcol64, x, om, dom_channels=32, dcol_chunked=True, comp=ops.F32)
    close(dx_f.cpu(), dx_r.cpu(), 3e-5, "dx vs f64 columns")
    close(dom_f.cpu(), dom_r.cpu(), 3e-5, "dom vs f64 columns")


@pytest.mark.parametrize("case", [(2, 16, 32, 64, 64, 1.5), (1, 16, 16, 128, 256, 5.0), (3, 8, 48, 32, 128, 0.0)])
def test_dcn_forward_writes_the_columns(T, dev, x3, case):
    """the f16x3 DCNv2 window kernel (64-, 128-cout tiles; two cout tiles for 256 couts: the first one writes) stores the sampled
    columns as a by-product: bit-identical outputs with and without, columns == the stand-alone sampling kernel's (the same f32
    blend) and == the oracle's bilinear sampling"""
    ops, ot = T
    B, H, W, Cin, Cout, off_std = case
    g = torch.Generator().manual_seed(int(sum(case[:5])))
    x = torch.randn(B, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).to(dev)
    om = torch.randn(B, 27, H, W, generator=g)
    om[:, :18] *= off_std
    om[:, 0, 0, 0], om[:, 1, 0, 0], om[:, 4, 3, 5], om[:, 5, 3, 5] = -0.0, -1.0, 4.0, -4.0
    xd = nhwc(x).to(dev)
    omd = torch.zeros(B, H, W, 28)
    omd[..., :27] = nhwc(om)
    omd = omd.to(dev)
    p = ops.PackedConv(w, None, None, stride=1, pad=1, compute=ops.F16X3)
    y0 = ops.dcnv2(xd, omd, p)
    y1, cols = ops.dcnv2(xd, omd, p, want_cols=True)
    assert cols is not None and torch.equal(y0, y1)
    ref = ot.dcn_cols(xd, omd)
    close(cols.cpu(), ref.cpu(), 1e-6, "columns vs the sampling kernel")
    t, c = 4, 3
    wsel = torch.zeros(1, Cin, 3, 3)
    wsel[0, c, t // 3, t % 3] = 1.0
    refo = O.dcnv2_forward(x, om[:, :18], torch.sigmoid(om[:, 18:]), wsel, None, 1, 1, 1)[:, 0]
    close(cols[..., t * Cin + c].cpu(), refo, 1e-5, "columns vs the oracle")


def test_deform_conv_node_x3(T, dev, x3):
    """DeformConvFn (offset conv + DCNv2 as one node) in the f16x3 mode against autograd through the oracle"""
    ops, ot = T
    g = torch.Generator().manual_seed(21)
    B, H, W, Cin, Cout = 2, 16, 32, 64, 64
    x = torch.randn(B, Cin, H, W, generator=g).requires_grad_(True)
    w_off = (torch.randn(27, Cin, 3, 3, generator=g) * 0.03).requires_grad_(True)
    b_off = (torch.randn(27, generator=g) * 0.1).requires_grad_(True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 24).requires_grad_(True)
    b = torch.randn(Cout, generator=g).requires_grad_(True)
    om = F.conv2d(x, w_off, b_off, 1, 1)
    y = O.dcnv2_forward(x, om[:, :18], torch.sigmoid(om[:, 18:]), w, b, 1, 1, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    leaves = [t.detach().to(dev).requires_grad_(True) for t in (w_off, b_off, w, b)]
    xd = nhwc(x.detach()).to(dev).requires_grad_(True)
    yd = ot.DeformConvFn.apply(xd, *leaves)
    close(nchw(yd.cpu()), y.detach(), 3e-5, "fwd")
    yd.backward(nhwc(dy).to(dev) * ot.GRAD_SCALE)
    # returned parameter gradients carry PARAM_GRAD_MULT = 1 / GRAD_SCALE (stand-alone: no flat buffer, autograd accumulates)
    for got, ref, name in zip(leaves, (w_off, b_off, w, b), ("w_off", "b_off", "w", "b")):
        close(got.grad.cpu(), ref.grad, 2e-4, name)
    close(nchw(xd.grad.cpu()) / ot.GRAD_SCALE, x.grad, 2e-4, "dx")


# ------------------------------------------------------------------------------------------ the whole step
def _oracle_step(model, cfg, inputs, sd0):
    sd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
          for k, v in sd0.items()}
    x_ref, _ = O.preprocess([d["image"] for d in inputs], cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD, 32)
    z = MR.centernet_forward(sd, x_ref, training=True, f16_activations=False)
    targets = [O.gen_heatmap(d["instances"].gt_boxes.tensor, d["instances"].gt_classes, 64, 64, 80) for d in inputs]
    ref = MR.centernet_losses(z, targets, [1.0])
    sum(ref.values()).backward()
    return sd, ref


def test_full_training_step_f16x3_matches_fp32_oracle(tmp_path, dev):
    """HIP_PRECISION f16x3 trains in f16x3 (round 3 fell back to the f32 kernels): the whole step -- targets, forward, losses,
    every backward kernel -- against torch autograd through the fp32 oracle at the f32 mode's bounds: losses within 1e-3
    (north_star), every parameter-gradient group at cosine >= 0.999 with norms within 1 %"""
    from test_model_gpu import cpu_state_dict, make_model
    from detectron2_centernet_amd import ops, ops_train
    from detectron2_centernet_amd.data.catalog import synthetic_sample
    from detectron2_centernet_amd.structures import Boxes, Instances

    model, cfg = make_model(tmp_path, "f16x3", seed=11)
    model.train()
    sd0 = cpu_state_dict(model)
    inputs = []
    for i in range(2):
        smp = synthetic_sample(i, size=256, num_classes=80, max_boxes=6)
        inst = Instances((256, 256))
        inst.gt_boxes = Boxes(smp["boxes"])
        inst.gt_classes = smp["classes"]
        inputs.append({"image": smp["image"], "instances": inst})
    calls = {"x3": 0, "other": 0}
    real = ops_train.conv_wgrad

    def counting(x, dy, *a, comp=None, **kw):
        calls["x3" if comp == ops.F16X3 else "other"] += 1
        return real(x, dy, *a, comp=comp, **kw)
    ops_train.conv_wgrad = counting
    try:
        losses = model(inputs)
        assert ops_train.F32_COMPUTE == ops.F16X3
        sum(losses.values()).backward()
    finally:
        ops_train.conv_wgrad = real
    assert calls["x3"] > 60 and calls["other"] == 0, calls        # every weight gradient of the step ran in f16x3
    sd, ref = _oracle_step(model, cfg, inputs, sd0)
    for k in ("hm_loss", "wh_loss", "off_loss"):
        got, want = losses[k].item(), ref[k].item()
        print(k, got, want)
        assert abs(got - want) <= 1e-3 * max(1.0, abs(want)), (k, got, want)
    worst = {}
    for name, p in model.named_parameters():
        gref = sd[name].grad
        if gref is None:
            assert p.grad is None or p.grad.abs().max() == 0, name
            continue
        assert p.grad is not None, name
        if name.endswith("conv.bias") and ".conv_offset_mask" not in name:
            continue  # a bias in front of BatchNorm has zero true gradient; both sides are rounding noise
        gg = p.grad.float().cpu()
        if gref.abs().max() == 0:
            continue
        cos = torch.nn.functional.cosine_similarity(gg.flatten(), gref.flatten(), dim=0).item()
        ratio = (gg.norm() / gref.norm()).item()
        group = "hm" if name.startswith("hm.") else "heads" if name.split(".")[0] in ("wh", "reg") else \
            ("ida_up" if name.startswith("backbone.ida_up") else ("dla_up" if "dla_up" in name else "base"))
        w = worst.setdefault(group, [1.0, 1.0, 1.0, ""])
        if cos < w[0]:
            w[0], w[3] = cos, name
        w[1], w[2] = min(w[1], ratio), max(w[2], ratio)
    print("f16x3 worst (cos, min ratio, max ratio, worst name) per group:", worst)
    for grp, (cos, rmin, rmax, name) in worst.items():
        assert cos >= 0.999, (grp, cos, name)
        assert 0.99 < rmin and rmax < 1.01, (grp, rmin, rmax)
    bn = model.backbone.base.base_layer[1]
    assert int(bn.num_batches_tracked) == 1


def test_training_f16x3_graph_replay_and_planned_packs(tmp_path, dev):
    """the f16x3 step under the trainer: captured as one HIP graph from the third call on, the split operands of every weight
    refreshed by ONE batched pack launch after the optimizer update (ops.PackPlan: after each step every planned f16x3 image
    equals a fresh pack of the live parameter), losses finite and falling on a fixed batch"""
    from test_model_gpu import make_model
    from detectron2_centernet_amd import ops
    from detectron2_centernet_amd.engine.bench_train import synthetic_batch
    from detectron2_centernet_amd.engine.train_loop import SimpleTrainer

    model, cfg = make_model(tmp_path, "f16x3", seed=4)
    cfg.SOLVER.IMS_PER_BATCH = 2
    cfg.SOLVER.BASE_LR = 2e-3
    tr = SimpleTrainer(model, None, cfg)
    batch = synthetic_batch(2, 128, 0, dev)
    hist = []
    for i in range(7):
        l = tr.run_step_tensors(*batch)
        hist.append({k: float(v) for k, v in l.items()})
        assert all(torch.isfinite(torch.tensor(list(hist[-1].values())))), hist
    assert tr.graph_state == "captured", tr._graphs
    for g in (g for g in tr._graphs.values() if g["graph"] is not None):   # kernels only (engine/graph_nodes.py)
        assert g["nodes"].get("kernel", 0) > 0 and set(g["nodes"]) <= {"kernel", "empty"}, g["nodes"]
    plan = ops.PACK_PLAN
    x3_entries = [e for e in plan.entries.values() if e[4] is not None]
    assert len(x3_entries) > 100, len(x3_entries)            # forward + input-gradient operands of ~55 convs, DCN column operands
    torch.cuda.synchronize()
    for w, packed, args, ver, scale in x3_entries[::7]:      # (replays refresh the buffers from inside the graph: the host-side
        fresh = torch.empty_like(packed)                     #  version bookkeeping only moves in eager steps)
        fs = torch.empty_like(scale)
        from detectron2_centernet_amd import _lib
        from detectron2_centernet_amd.ops import _ptr, _stream
        _lib.check(_lib.lib().ctdet_pack_weights_x3(_ptr(w), _ptr(fresh), _ptr(fs), *args, _stream()), "pack")
        assert torch.equal(fresh.view(torch.int32), packed.view(torch.int32)) and torch.equal(fs, scale)
    assert hist[-1]["hm_loss"] < hist[0]["hm_loss"], hist
