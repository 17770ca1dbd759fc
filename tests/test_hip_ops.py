"""GPU parity tests of every HIP kernel, called through the C ABI, against the CPU oracle.

Tolerances: index / class / mask outputs bit-exact; f32 kernels 1e-4 relative to the oracle's f32 result
(different summation order only); f16-MFMA kernels are fed f16-representable inputs so the only differences
are f32 accumulation order and the final f16 rounding (2e-3 relative).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ctdet_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import detectron2_centernet_amd.ops as ops

    return ops


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def h16(t):  # round to f16-representable f32
    return t.half().float()


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad, residual, relu
    (2, 20, 24, 16, 16, 3, 1, 1, False, True),
    (2, 20, 24, 16, 32, 3, 2, 1, False, True),
    (1, 18, 22, 32, 64, 3, 2, 1, False, True),
    (2, 16, 16, 64, 64, 3, 1, 1, True, True),
    (1, 12, 20, 128, 128, 3, 1, 1, True, True),
    (1, 8, 8, 256, 256, 3, 1, 1, False, False),
    (3, 9, 7, 448, 128, 1, 1, 0, False, True),
    (1, 16, 16, 64, 768, 3, 1, 1, False, True),
    (1, 16, 16, 256, 80, 1, 1, 0, False, False),
    (1, 10, 10, 64, 27, 3, 1, 1, False, False),
    (5, 40, 40, 64, 64, 3, 1, 1, False, True),   # > 512 tiles -> 256-pixel tile variant
    (4, 64, 64, 128, 128, 3, 1, 1, True, True),
    # 16-pixel-wide maps (DLA-34's 512-channel level at 512^2 input)
    (4, 16, 16, 128, 128, 3, 1, 1, True, True),
    (2, 8, 16, 512, 27, 3, 1, 1, False, False),
    (6, 24, 16, 64, 40, 3, 1, 1, False, True),
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("mode", ["f16", "f32"])
def test_conv2d(ops, dev, case, mode):
    B, H, W, Cin, Cout, k, s, p, use_res, relu = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = h16(torch.randn(B, Cin, H, W, generator=g))
    w = h16(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
    scale = torch.rand(Cout, generator=g) + 0.5
    bias = torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, None, s, p) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    res = None
    if use_res:
        res = h16(torch.randn(ref.shape, generator=g))
        ref = ref + res
    if relu:
        ref = ref.relu()
    comp = ops.F16 if mode == "f16" else ops.F32
    tdt = torch.float16 if mode == "f16" else torch.float32
    pc = ops.PackedConv(w.to(dev), scale.to(dev), bias.to(dev), stride=s, pad=p, compute=comp)
    res_d = None
    if res is not None:
        res_d = torch.zeros(ref.shape[0], ref.shape[2], ref.shape[3], pc.Cout_eff, dtype=tdt)
        res_d[..., :Cout] = nhwc(res).to(tdt)
        res_d = res_d.to(dev)
    y = ops.conv2d(nhwc(x).to(tdt).to(dev), pc, act=ops.ACT_RELU if relu else ops.ACT_NONE, residual=res_d)
    got = nchw(y[..., :Cout].float().cpu())
    tol = 3e-3 if mode == "f16" else 1e-4
    err = (got - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), f"max err {err}"


def test_conv2d_random_shapes(ops, dev):
    """seeded sweep over shapes the fixed cases do not hit: every dispatch branch (halo / uniform-K / generic DMA / small-channel
    kernels, 128- and 256-pixel tiles, ragged maps, padded couts) must agree with torch"""
    rng = np.random.RandomState(20261004)
    for it in range(48):
        k = int(rng.choice([1, 3, 3]))
        stride = int(rng.choice([1, 1, 2]))
        Cin = int(rng.choice([8, 16, 24, 32, 64, 96, 128, 160]))
        Cout = int(rng.choice([3, 4, 16, 27, 32, 40, 64, 80, 128, 132]))
        B = int(rng.randint(1, 4))
        H, W = int(rng.randint(3, 41)), int(rng.randint(3, 41))
        if rng.rand() < 0.3:                       # tile-divisible maps: the halo-resident kernel's domain
            H, W = 8 * int(rng.randint(1, 4)), 32 * int(rng.randint(1, 3))
        use_res, relu = bool(rng.rand() < 0.4), bool(rng.rand() < 0.6)
        g = torch.Generator().manual_seed(1000 + it)
        x = h16(torch.randn(B, Cin, H, W, generator=g))
        w = h16(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
        scale = torch.rand(Cout, generator=g) + 0.5
        bias = torch.randn(Cout, generator=g)
        ref = F.conv2d(x, w, None, stride, k // 2) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
        pc = ops.PackedConv(w.to(dev), scale.to(dev), bias.to(dev), stride=stride, pad=k // 2, compute=ops.F16)
        res_d = None
        if use_res:
            res = h16(torch.randn(ref.shape, generator=g))
            ref = ref + res
            res_d = torch.zeros(ref.shape[0], ref.shape[2], ref.shape[3], pc.Cout_eff, dtype=torch.float16)
            res_d[..., :Cout] = nhwc(res).half()
            res_d = res_d.to(dev)
        if relu:
            ref = ref.relu()
        y = ops.conv2d(nhwc(x).half().to(dev), pc, act=ops.ACT_RELU if relu else ops.ACT_NONE, residual=res_d)
        got = nchw(y[..., :Cout].float().cpu())
        assert got.shape == ref.shape, (it, got.shape, ref.shape)
        err = (got - ref).abs().max().item()
        assert err < 4e-3 * max(1.0, ref.abs().max().item()), \
            f"case {it}: B{B} {H}x{W} {Cin}->{Cout} k{k} s{stride} res={use_res} relu={relu}: max err {err}"


def test_dcnv2_random_shapes(ops, dev):
    """seeded sweep for the DCNv2 window kernel: partial edge tiles, 64- and 128-cout tiles, padded couts, offsets from inside
    to far outside the LDS window"""
    rng = np.random.RandomState(7)
    for it in range(16):
        Cin = int(rng.choice([32, 64, 128]))
        Cout = int(rng.choice([8, 40, 64, 128, 132, 200]))
        B, H, W = int(rng.randint(1, 3)), int(rng.randint(3, 30)), int(rng.randint(3, 40))
        off_std = float(rng.choice([0.0, 0.7, 2.0, 5.0, 9.0]))
        g = torch.Generator().manual_seed(500 + it)
        x = h16(torch.randn(B, Cin, H, W, generator=g))
        w = h16(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
        om = torch.randn(B, 27, H, W, generator=g)
        om[:, :18] *= off_std
        ref = O.dcnv2_forward(x, om[:, :18], torch.sigmoid(om[:, 18:]), w, None, 1, 1, 1).relu()
        pc = ops.PackedConv(w.to(dev), None, None, stride=1, pad=1, compute=ops.F16, cout_align=64)
        om_d = torch.zeros(B, H, W, 28)
        om_d[..., :27] = nhwc(om)
        y = ops.dcnv2(nhwc(x).half().to(dev), om_d.to(dev), pc, act=ops.ACT_RELU)
        err = (nchw(y[..., :Cout].float().cpu()) - ref).abs().max().item()
        assert err <= 6e-3 * max(1.0, ref.abs().max().item()), f"case {it}: B{B} {H}x{W} {Cin}->{Cout} std {off_std}: {err}"
    # weights packed for an ordinary conv of a narrow layer (16-row tile) are refused with a message that says what to do
    narrow = ops.PackedConv(torch.randn(8, 32, 3, 3).to(dev), None, None, stride=1, pad=1, compute=ops.F16)
    with pytest.raises(ValueError, match="cout_align=64"):
        ops.dcnv2(torch.zeros(1, 8, 16, 32, dtype=torch.float16, device=dev), torch.zeros(1, 8, 16, 28, device=dev), narrow)


def test_conv1x1_cat_random_splits(ops, dev):
    """Root (dla.py:86-94) reads its children in place: 1..4 sources with channel counts that are / are not multiples of 32,
    odd maps, with and without residual"""
    rng = np.random.RandomState(31)
    for it in range(12):
        n = int(rng.randint(1, 5))
        cins = [int(rng.choice([8, 16, 24, 32, 64, 128])) for _ in range(n)]
        Cout = int(rng.choice([16, 40, 64, 128, 256]))
        B, H, W = int(rng.randint(1, 3)), int(rng.randint(3, 24)), int(rng.randint(3, 24))
        g = torch.Generator().manual_seed(600 + it)
        xs = [h16(torch.randn(B, c, H, W, generator=g)) for c in cins]
        w = h16(torch.randn(Cout, sum(cins), 1, 1, generator=g) / sum(cins) ** 0.5)
        scale, bias = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
        ref = (F.conv2d(torch.cat(xs, 1), w) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)).relu()
        pc = ops.PackedConv(w.to(dev), scale.to(dev), bias.to(dev), compute=ops.F16)
        y = ops.conv1x1_cat([nhwc(x).half().to(dev) for x in xs], pc, act=ops.ACT_RELU)
        err = (nchw(y[..., :Cout].float().cpu()) - ref).abs().max().item()
        assert err < 4e-3 * max(1.0, ref.abs().max().item()), f"case {it}: {cins}->{Cout} {H}x{W}: {err}"


def test_dla_base_fused_random_sizes(ops, dev):
    """seeded sweep of the fused base kernel: image smaller than / equal to the padded size, tall and wide maps, 1..3 images,
    byte and f32 images, against torch with f16-rounded intermediate maps; the pooled output equals maxpool2x2 of the main one"""
    rng = np.random.RandomState(41)
    mean, std = [0.408, 0.447, 0.470], [0.289, 0.274, 0.278]
    g = torch.Generator().manual_seed(77)
    ws = [h16(torch.randn(16, 3, 7, 7, generator=g) / 147 ** 0.5), h16(torch.randn(16, 16, 3, 3, generator=g) / 12),
          h16(torch.randn(32, 16, 3, 3, generator=g) / 12)]
    sb = [(torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3) for c in (16, 16, 32)]
    args = []
    for w, (sc, bi) in zip(ws, sb):
        args += [w.to(dev), (sc.to(dev), bi.to(dev))]
    pb = ops.PackedDlaBase(*args)
    for it in range(8):
        Hp, Wp = 16 * int(rng.randint(1, 12)), 32 * int(rng.randint(1, 7))
        H, W = Hp - int(rng.randint(0, 16)), Wp - int(rng.randint(0, 32))
        B = int(rng.randint(1, 4))
        dt = torch.uint8 if rng.rand() < 0.7 else torch.float32
        img = torch.randint(0, 256, (B, 3, H, W), generator=g).to(dt)
        x = (img.float() / 255 - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
        ref = h16(F.pad(x, (0, Wp - W, 0, Hp - H)))
        for w, (sc, bi), (st, pd) in zip(ws, sb, ((1, 3), (1, 1), (2, 1))):
            ref = h16((F.conv2d(ref, w, None, st, pd) * sc.view(1, -1, 1, 1) + bi.view(1, -1, 1, 1)).relu())
        pooled = torch.empty(B, Hp // 4, Wp // 4, 32, dtype=torch.float16, device=dev)
        y = ops.dla_base_fused(img.to(dev), mean, std, Hp, Wp, pb, pooled=pooled)
        err = (nchw(y.float().cpu()) - ref).abs().max().item()
        assert err <= 3e-3 * max(1.0, ref.abs().max().item()), f"case {it}: B{B} {H}x{W} in {Hp}x{Wp} {dt}: {err}"
        assert torch.equal(pooled, ops.maxpool2x2(y)), f"case {it}: pooled output"


def test_conv_f16_f32_output_and_slices(ops, dev):
    """f32 output from the f16 MFMA kernel, reading a channel slice and writing into a slice of a wider buffer."""
    g = torch.Generator().manual_seed(3)
    B, H, W = 2, 12, 12
    xw = h16(torch.randn(B, H, W, 96, generator=g))
    w = h16(torch.randn(32, 64, 1, 1, generator=g) / 8)
    pc = ops.PackedConv(w.to(dev), None, None, compute=ops.F16)
    big = torch.zeros(B, H, W, 64, dtype=torch.float32, device=dev)
    ops.conv2d(xw.half().to(dev)[..., 32:96], pc, out=big[..., 16:48])
    ref = nhwc(F.conv2d(nchw(xw[..., 32:96]), w))
    assert (big[..., 16:48].cpu() - ref).abs().max() < 2e-3
    assert big[..., :16].abs().max() == 0 and big[..., 48:].abs().max() == 0


def test_stem_7x7(ops, dev):
    g = torch.Generator().manual_seed(4)
    x = h16(torch.randn(2, 3, 32, 40, generator=g))
    w = h16(torch.randn(16, 3, 7, 7, generator=g) / 12)
    ref = F.conv2d(x, w, None, 1, 3).relu()
    x8 = torch.zeros(2, 32, 40, 8)
    x8[..., :3] = nhwc(x)
    for comp, tdt, tol in ((ops.F16, torch.float16, 3e-3), (ops.F32, torch.float32, 1e-4)):
        pc = ops.PackedConv(w.to(dev), None, None, stride=1, pad=3, compute=comp, cin_pad=8)
        y = ops.conv2d(x8.to(tdt).to(dev), pc, act=ops.ACT_RELU)
        assert (nchw(y.float().cpu()) - ref).abs().max() < tol * ref.abs().max()


# LDS-window small-channel kernel (conv_win_kernel): the three DLA base layers at tile-divisible sizes, including
# image borders (zero fill by the DMA source select) and the pre-padded stem (pad 0 on a framed input)
@pytest.mark.parametrize("case", [("stem", 7, 3, 8, 16, 1, 3), ("stem_prepad", 7, 3, 8, 16, 1, 0),
                                  ("level0", 3, 16, 16, 16, 1, 1), ("level1", 3, 16, 16, 32, 2, 1)])
def test_conv_window_small_channels(ops, dev, case):
    name, R, cin, cin_pad, cout, stride, pad = case
    g = torch.Generator().manual_seed(R + cout + stride)
    B, H, W = 2, 32 * stride, 128 * stride
    x = h16(torch.randn(B, cin, H, W, generator=g))
    w = h16(torch.randn(cout, cin, R, R, generator=g) / (cin * R * R) ** 0.5)
    scale = torch.rand(cout, generator=g) + 0.5
    bias = torch.randn(cout, generator=g)
    ref = (F.conv2d(x, w, None, stride, R // 2) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)).relu()
    fr = R // 2 if pad == 0 else 0           # frame carried in memory by the pre-padded variant
    xn = torch.zeros(B, H + 2 * fr, W + 2 * fr, cin_pad)
    xn[:, fr:fr + H, fr:fr + W, :cin] = nhwc(x)
    pc = ops.PackedConv(w.to(dev), scale.to(dev), bias.to(dev), stride=stride, pad=pad, compute=ops.F16,
                        cin_pad=cin_pad, tap_major=True)
    y = ops.conv2d(xn.half().to(dev), pc, act=ops.ACT_RELU)
    got = nchw(y[..., :cout].float().cpu())
    assert got.shape == ref.shape
    err = (got - ref).abs().max().item()
    assert err < 3e-3 * max(1.0, ref.abs().max().item()), f"{name}: max err {err}"


@pytest.mark.parametrize("case", [("u8_ragged", torch.uint8, 2, 50, 70, 64, 96), ("u8_full", torch.uint8, 3, 64, 64, 64, 64),
                                  ("f32", torch.float32, 1, 32, 64, 32, 64), ("u8_tall", torch.uint8, 1, 130, 40, 144, 64),
                                  ("u8_interior", torch.uint8, 2, 128, 160, 128, 160),
                                  ("u8_interior_ragged", torch.uint8, 1, 121, 150, 128, 160),
                                  ("f32_interior", torch.float32, 1, 96, 128, 96, 128)])
def test_dla_base_fused(ops, dev, case):
    """normalisation + 7x7 stem + level0 + level1 in one launch vs (a) torch with the maps rounded to f16 where the
    layer-by-layer path rounds them and (b) that layer-by-layer HIP path itself; image smaller than the padded input,
    tiles on every border, uint8 and f32 images"""
    name, dt, B, H, W, Hp, Wp = case
    g = torch.Generator().manual_seed(H + W)
    img = torch.randint(0, 256, (B, 3, H, W), generator=g).to(dt)
    mean, std = [103.5 / 255, 116.3 / 255, 123.7 / 255], [0.225, 0.224, 0.229]
    ws = [h16(torch.randn(16, 3, 7, 7, generator=g) / 147 ** 0.5), h16(torch.randn(16, 16, 3, 3, generator=g) / 12),
          h16(torch.randn(32, 16, 3, 3, generator=g) / 12)]
    sb = [(torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3) for c in (16, 16, 32)]
    x = (img.float() / 255 - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    x = h16(F.pad(x, (0, Wp - W, 0, Hp - H)))
    ref = x
    for w, (sc, bi), (st, pd) in zip(ws, sb, ((1, 3), (1, 1), (2, 1))):
        ref = h16((F.conv2d(ref, w, None, st, pd) * sc.view(1, -1, 1, 1) + bi.view(1, -1, 1, 1)).relu())
    args = []
    for w, (sc, bi) in zip(ws, sb):
        args += [w.to(dev), (sc.to(dev), bi.to(dev))]
    pb = ops.PackedDlaBase(*args)
    pooled = torch.empty(B, Hp // 4, Wp // 4, 32, dtype=torch.float16, device=dev)
    y = ops.dla_base_fused(img.to(dev), mean, std, Hp, Wp, pb, pooled=pooled)
    assert torch.equal(pooled, ops.maxpool2x2(y)), f"{name}: fused 2x2 max-pool"
    assert torch.equal(y, ops.dla_base_fused(img.to(dev), mean, std, Hp, Wp, pb)), f"{name}: with / without the pooled output"
    got = nchw(y.float().cpu())
    assert got.shape == ref.shape
    err = (got - ref).abs().max().item()
    assert err <= 3e-3 * max(1.0, ref.abs().max().item()), f"{name}: max err {err} vs torch"
    # the layer-by-layer HIP path on the same operands: identical rounding points -> at most an f16 ulp or two apart
    t = ops.preprocess(img.to(dev), mean, std, Hp, Wp)
    p0 = ops.PackedConv(ws[0].to(dev), sb[0][0].to(dev), sb[0][1].to(dev), stride=1, pad=3, compute=ops.F16, cin_pad=8,
                        tap_major=True)
    t = ops.conv2d(t, p0, act=ops.ACT_RELU)
    t = ops.conv2d(t, pb.p1, act=ops.ACT_RELU)
    t = ops.conv2d(t, pb.p2, act=ops.ACT_RELU)
    d = (t.float() - y.float()).abs().max().item()
    assert d <= 4e-3 * max(1.0, ref.abs().max().item()), f"{name}: {d} vs the layer-by-layer path"
    assert (t != y).float().mean().item() < 0.02


@pytest.mark.parametrize("case", [("u8_ragged", torch.uint8, 2, 50, 70, 64, 96), ("u8_full", torch.uint8, 3, 64, 64, 64, 64),
                                  ("f32", torch.float32, 1, 32, 64, 32, 64), ("u8_tall", torch.uint8, 1, 130, 40, 144, 64),
                                  ("u8_interior", torch.uint8, 2, 128, 160, 128, 160),
                                  ("u8_interior_ragged", torch.uint8, 1, 121, 150, 128, 160),
                                  ("f32_interior", torch.float32, 1, 96, 128, 96, 128)])
def test_dla_base_fused_f16x3(ops, dev, case):
    """the f16x3 form of the fused base (f32 tensors, hi*hi + lo*hi + hi*lo on the f16 matrix pipe): against torch fp32 on the
    same f32 weights (f32-grade: 2e-5 of the output range through three layers, tolerance written here) and against the
    layer-by-layer f16x3 HIP path on the same packed operands (same split points, same products); pooled output == 2x2 max-pool"""
    name, dt, B, H, W, Hp, Wp = case
    g = torch.Generator().manual_seed(H + W + 1)
    img = torch.randint(0, 256, (B, 3, H, W), generator=g).to(dt)
    mean, std = [103.5 / 255, 116.3 / 255, 123.7 / 255], [0.225, 0.224, 0.229]
    ws = [torch.randn(16, 3, 7, 7, generator=g) / 147 ** 0.5, torch.randn(16, 16, 3, 3, generator=g) / 12,
          torch.randn(32, 16, 3, 3, generator=g) / 12]
    sb = [(torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3) for c in (16, 16, 32)]
    x = (img.float() / 255 - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    ref = F.pad(x, (0, Wp - W, 0, Hp - H)).double()
    for w, (sc, bi), (st, pd) in zip(ws, sb, ((1, 3), (1, 1), (2, 1))):
        ref = (F.conv2d(ref, w.double(), None, st, pd) * sc.double().view(1, -1, 1, 1) + bi.double().view(1, -1, 1, 1)).relu()
    args = []
    for w, (sc, bi) in zip(ws, sb):
        args += [w.to(dev), (sc.to(dev), bi.to(dev))]
    pb = ops.PackedDlaBaseX3(*args)
    pooled = torch.empty(B, Hp // 4, Wp // 4, 32, dtype=torch.float32, device=dev)
    y = ops.dla_base_fused(img.to(dev), mean, std, Hp, Wp, pb, pooled=pooled)
    assert y.dtype == torch.float32
    assert torch.equal(pooled, ops.maxpool2x2(y)), f"{name}: fused 2x2 max-pool"
    assert torch.equal(y, ops.dla_base_fused(img.to(dev), mean, std, Hp, Wp, pb)), f"{name}: with / without the pooled output"
    got = nchw(y.cpu()).double()
    assert got.shape == ref.shape
    err = (got - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), f"{name}: max err {err} vs torch fp64"
    # the layer-by-layer f16x3 path on the same operands
    t = ops.preprocess(img.to(dev), mean, std, Hp, Wp, out=torch.empty(B, Hp, Wp, 4, dtype=torch.float32, device=dev))
    p0 = ops.PackedConv(ws[0].to(dev), sb[0][0].to(dev), sb[0][1].to(dev), stride=1, pad=3, compute=ops.F16X3, cin_pad=4)
    t = ops.conv2d(t, p0, act=ops.ACT_RELU)
    t = ops.conv2d(t, pb.p1, act=ops.ACT_RELU)
    t = ops.conv2d(t, pb.p2, act=ops.ACT_RELU)
    d = (t[..., :32] - y).abs().max().item()
    assert d <= 1e-5 * max(1.0, ref.abs().max().item()), f"{name}: {d} vs the layer-by-layer path"


def test_dla_base_fused_rejects_bad_shapes(ops, dev):
    g = torch.Generator().manual_seed(1)
    args = [torch.randn(16, 3, 7, 7, generator=g).to(dev), (torch.ones(16, device=dev), torch.zeros(16, device=dev)),
            torch.randn(16, 16, 3, 3, generator=g).to(dev), (torch.ones(16, device=dev), torch.zeros(16, device=dev)),
            torch.randn(32, 16, 3, 3, generator=g).to(dev), (torch.ones(32, device=dev), torch.zeros(32, device=dev))]
    pb = ops.PackedDlaBase(*args)
    img = torch.zeros(1, 3, 40, 40, dtype=torch.uint8, device=dev)
    with pytest.raises(RuntimeError, match="multiple"):
        ops.dla_base_fused(img, [0, 0, 0], [1, 1, 1], 40, 40, pb)


@pytest.mark.parametrize("case", [(2, 16, 32, 64, (80, 2, 2)), (1, 8, 16, 64, (5,)), (1, 24, 16, 128, (20, 2, 2, 7))])
def test_heads_fused(ops, dev, case):
    """fused 3x3 + ReLU + 1x1 heads vs torch (f16 operands, f32 accumulation; the hidden map is rounded to f16 as in
    the unfused path), hm head with sigmoid + clamp, image borders included"""
    B, H, W, Cin, couts = case
    g = torch.Generator().manual_seed(Cin + len(couts))
    x = h16(torch.randn(B, Cin, H, W, generator=g))
    w1 = [h16(torch.randn(256, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5) for _ in couts]
    b1 = [torch.randn(256, generator=g) * 0.3 for _ in couts]
    w2 = [h16(torch.randn(c, 256, 1, 1, generator=g) / 16) for c in couts]
    b2 = [torch.randn(c, generator=g) for c in couts]
    acts = [ops.ACT_SIGMOID_CLAMP if i == 0 else ops.ACT_NONE for i in range(len(couts))]
    ph = ops.PackedHeads([w.to(dev) for w in w1], [b.to(dev) for b in b1], [w.to(dev) for w in w2],
                         [b.to(dev) for b in b2], acts)
    outs = ops.heads_fused(nhwc(x).half().to(dev), ph, clamp=(1e-4, 1 - 1e-4))
    for i, c in enumerate(couts):
        hid = h16(F.conv2d(x, w1[i], b1[i], 1, 1).relu())
        ref = F.conv2d(hid, w2[i], b2[i])
        if i == 0:
            ref = torch.clamp(torch.sigmoid(ref), 1e-4, 1 - 1e-4)
        got = nchw(outs[i][..., :c].cpu())
        assert got.shape == ref.shape
        err = (got - ref).abs().max().item()
        assert err <= 3e-3 * max(1.0, ref.abs().max().item()), f"head {i}: max err {err}"


DCN_CASES = [(2, 12, 14, 64, 64, 2.0), (1, 9, 9, 128, 64, 0.0), (2, 8, 10, 128, 128, 4.0), (1, 6, 6, 256, 256, 1.0),
             (1, 7, 5, 512, 256, 8.0)]


@pytest.mark.parametrize("case", DCN_CASES)
@pytest.mark.parametrize("mode", ["f16", "f32"])
def test_dcnv2(ops, dev, case, mode):
    B, H, W, Cin, Cout, off_std = case
    g = torch.Generator().manual_seed(Cin + Cout + H)
    x = h16(torch.randn(B, Cin, H, W, generator=g))
    w = h16(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    om = torch.randn(B, 27, H, W, generator=g)
    om[:, :18] *= off_std
    # put a few samples exactly on integer / border coordinates
    om[:, 0, 0, 0] = -0.0
    om[:, 1, 0, 0] = -1.0
    om[:, 2, -1, -1] = 1.0
    bias = torch.randn(Cout, generator=g)
    scale = torch.rand(Cout, generator=g) + 0.5
    ref = O.dcnv2_forward(x, om[:, :18], torch.sigmoid(om[:, 18:]), w, None, 1, 1, 1)
    ref = (ref * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)).relu()
    comp = ops.F16 if mode == "f16" else ops.F32
    tdt = torch.float16 if mode == "f16" else torch.float32
    pc = ops.PackedConv(w.to(dev), scale.to(dev), bias.to(dev), stride=1, pad=1, compute=comp)
    om_d = torch.zeros(B, H, W, 32)
    om_d[..., :27] = nhwc(om)
    y = ops.dcnv2(nhwc(x).to(tdt).to(dev), om_d.to(dev), pc, act=ops.ACT_RELU)
    got = nchw(y[..., :Cout].float().cpu())
    # f16 mode additionally rounds the sampled*mask operand to f16 before the MFMA
    tol = 6e-3 if mode == "f16" else 2e-4
    err = (got - ref).abs().max().item()
    assert err <= tol * max(1.0, ref.abs().max().item()), f"max err {err}"


# LDS-window kernel: off_std 0/1/2 stay inside the +-4 px window margin (fast path), 6 and 12 push samples outside it so
# some / all waves take the gather-from-global path for those taps; both must agree with the oracle.  (test_dcnv2 above
# covers maps the 8x16 tile does not divide: partial edge tiles.)
DCN_WINDOW_CASES = [(2, 16, 32, 64, 64, 1.0), (1, 8, 16, 128, 64, 0.0), (2, 16, 16, 64, 64, 6.0),
                    (1, 8, 32, 256, 128, 2.0), (1, 24, 16, 128, 128, 3.0), (1, 8, 16, 512, 256, 12.0),
                    (1, 16, 48, 64, 40, 2.5)]


@pytest.mark.parametrize("variant", ["default", "v1", "mixed"])
@pytest.mark.parametrize("case", DCN_WINDOW_CASES)
def test_dcnv2_window(ops, dev, case, variant):
    """default: dcn_window_rows_kernel for the 64-cout layers, dcn_window_kernel for wider ones; v1: dcn_window_kernel for all
    (CTDET_TUNING_DCN_WINDOW_V1); mixed: its CTDET_TUNING_DCN_MIXED instantiation (only the lanes that left the window
    gather from global memory)"""
    from detectron2_centernet_amd import _lib
    B, H, W, Cin, Cout, off_std = case
    g = torch.Generator().manual_seed(Cin + Cout + H + 1)
    x = h16(torch.randn(B, Cin, H, W, generator=g))
    w = h16(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    om = torch.randn(B, 27, H, W, generator=g)
    om[:, :18] *= off_std
    om[:, 0, 0, 0] = -0.0
    om[:, 1, 0, 0] = -1.0
    om[:, 2, -1, -1] = 1.0
    om[:, 4, 3, 5] = 4.0      # exactly on the window edge
    om[:, 5, 3, 5] = -4.0
    bias = torch.randn(Cout, generator=g)
    scale = torch.rand(Cout, generator=g) + 0.5
    ref = O.dcnv2_forward(x, om[:, :18], torch.sigmoid(om[:, 18:]), w, None, 1, 1, 1)
    ref = (ref * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)).relu()
    pc = ops.PackedConv(w.to(dev), scale.to(dev), bias.to(dev), stride=1, pad=1, compute=ops.F16)
    assert pc.korder == 1
    om_d = torch.zeros(B, H, W, 28)
    om_d[..., :27] = nhwc(om)
    xd = nhwc(x).half().to(dev)
    with _lib.tuning({"default": 0, "v1": _lib.TUNE_DCN_WINDOW_V1, "mixed": _lib.TUNE_DCN_MIXED}[variant]):
        y = ops.dcnv2(xd, om_d.to(dev), pc, act=ops.ACT_RELU)
    got = nchw(y[..., :Cout].float().cpu())
    err = (got - ref).abs().max().item()
    assert err <= 6e-3 * max(1.0, ref.abs().max().item()), f"max err {err}"


@pytest.mark.parametrize("case", [(2, 16, 32, 64, 64, 1.0), (1, 8, 16, 128, 64, 0.3), (2, 16, 16, 64, 40, 4.0), (1, 24, 48, 256, 64, 2.0),
                                  (3, 8, 32, 32, 16, 1.0)])
def test_dcnv2_with_fused_offset_conv(ops, dev, case):
    """ctdet_dcnv2_offset_fwd: the 3x3 offset / mask conv evaluated inside the DCNv2 kernel from the same LDS window.  Against
    the two-kernel path on the same weights: offsets / mask logits (om_out) to f32 rounding of a different summation order,
    the output to the f16 tolerance of the two-kernel path against the oracle; and against the oracle itself"""
    B, H, W, Cin, Cout, off_scale = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = h16(torch.randn(B, Cin, H, W, generator=g))
    w = h16(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    w_off = h16(torch.randn(27, Cin, 3, 3, generator=g) * (off_scale / (Cin * 9) ** 0.5))
    b_off = torch.randn(27, generator=g) * 0.5
    bias, scale = torch.randn(Cout, generator=g), torch.rand(Cout, generator=g) + 0.5
    om_ref = F.conv2d(x, w_off, b_off, 1, 1)
    ref = O.dcnv2_forward(x, om_ref[:, :18], torch.sigmoid(om_ref[:, 18:]), w, None, 1, 1, 1)
    ref = (ref * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)).relu()
    pc = ops.PackedConv(w.to(dev), scale.to(dev), bias.to(dev), stride=1, pad=1, compute=ops.F16, cout_align=64)
    po = ops.PackedConv(w_off.to(dev), None, b_off.to(dev), stride=1, pad=1, compute=ops.F16)
    xd = nhwc(x).half().to(dev)
    assert ops.dcnv2_offset_supported(xd, po, pc)
    om_out = torch.full((B, H, W, 28), float("nan"), device=dev)
    y = ops.dcnv2_offset(xd, po, pc, act=ops.ACT_RELU, om_out=om_out)
    om2 = ops.conv2d(xd, po, out_dtype=torch.float32)
    y2 = ops.dcnv2(xd, om2, pc, act=ops.ACT_RELU)
    assert (om_out[..., :27] - om2[..., :27]).abs().max().item() <= 2e-5 * max(1.0, om2.abs().max().item())
    assert (nchw(om_out[..., :27].cpu()) - om_ref).abs().max().item() <= 1e-4 * max(1.0, om_ref.abs().max().item())
    tol = 6e-3 * max(1.0, ref.abs().max().item())
    assert (y.float() - y2.float()).abs().max().item() <= tol
    assert (nchw(y[..., :Cout].float().cpu()) - ref).abs().max().item() <= tol
    y3 = ops.dcnv2_offset(xd, po, pc, act=ops.ACT_RELU)          # without om_out
    assert torch.equal(y3, y)
    # geometry the fused kernel does not serve is reported as such
    assert not ops.dcnv2_offset_supported(xd[:, :7], po, pc)


@pytest.mark.parametrize("tdt", [torch.float16, torch.float32])
def test_maxpool_and_dwconvT(ops, dev, tdt):
    g = torch.Generator().manual_seed(7)
    x = h16(torch.randn(2, 64, 12, 16, generator=g))
    y = ops.maxpool2x2(nhwc(x).to(tdt).to(dev))
    assert torch.equal(nchw(y.float().cpu()), F.max_pool2d(x, 2, 2))
    tol = 2e-3 if tdt == torch.float16 else 1e-5
    # C = 64: the row-mapped kernel (f = 2, 4, 8, channel vectors a power of two); C = 24: the generic one
    for Cc, f in ((64, 2), (64, 4), (64, 8), (24, 2), (24, 4)):
        xc = x[:, :Cc]
        w = torch.rand(Cc, 1, 2 * f, 2 * f, generator=g)
        skip = h16(torch.randn(2, Cc, 12 * f, 16 * f, generator=g))
        up = F.conv_transpose2d(xc, w, None, stride=f, padding=f // 2, groups=Cc)
        out = ops.dwconvT_add(nhwc(xc).to(tdt).to(dev), w.to(dev), f, skip=nhwc(skip).to(tdt).to(dev))
        assert (nchw(out.float().cpu()) - (up + skip)).abs().max() < tol * (up + skip).abs().max(), (Cc, f)
        out = ops.dwconvT_add(nhwc(xc).to(tdt).to(dev), w.to(dev), f)
        assert (nchw(out.float().cpu()) - up).abs().max() < tol * up.abs().max(), (Cc, f, "no skip")


def test_preprocess(ops, dev):
    g = torch.Generator().manual_seed(11)
    img = torch.randint(0, 256, (3, 3, 50, 70), generator=g, dtype=torch.uint8)
    mean, std = [0.408, 0.447, 0.470], [0.289, 0.274, 0.278]
    ref, _ = O.preprocess([i for i in img], mean, std, 32)
    out = ops.preprocess(img.to(dev), mean, std, 64, 96, out_dtype=torch.float32)
    assert torch.allclose(nchw(out[..., :3].cpu()), ref, atol=2e-6, rtol=1e-6)
    assert out[..., 3:].abs().max() == 0
    out16 = ops.preprocess(img.float().to(dev), mean, std, 64, 96, out_dtype=torch.float16)
    assert (nchw(out16[..., :3].float().cpu()) - ref).abs().max() < 2e-3


def _rand_heat(B, C, H, W, seed, lo=1e-4, hi=1 - 1e-4):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(B, C, H, W, generator=g) * 1.5 - 2.19
    return torch.clamp(torch.sigmoid(logits), lo, hi)


@pytest.mark.parametrize("case", [(1, 8, 32, 64, 64), (2, 16, 64, 128, 27), (3, 24, 32, 64, 80), (2, 8, 64, 256, 256), (1, 16, 32, 192, 128)])
def test_f16_halo_two_tap_kernel(ops, dev, case):
    """3x3 / s1 / p1, Cin % 64 == 0 on tile-divisible maps in the f16 mode: conv3x3_halo_tap2_kernel (two taps per K step, tap 8 of a
    32-channel chunk paired with tap 8 of the next) against torch on f16-representable operands (products exact, f32 sums: 1e-5)
    and against the per-tap kernel it replaces (same f32 values up to the summation order; f16 outputs within one ulp)"""
    from detectron2_centernet_amd import _lib
    B, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    x = h16(torch.randn(B, Cin, H, W, generator=g))
    w = h16(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    bias = torch.randn(Cout, generator=g)
    ref = (F.conv2d(x, w, None, 1, 1) + bias.view(1, -1, 1, 1)).relu()
    pc = ops.PackedConv(w.to(dev), None, bias.to(dev), stride=1, pad=1, compute=ops.F16)
    xd = nhwc(x).half().to(dev)
    outs = {}
    for name, flag in (("tap2", 0), ("per_tap", _lib.TUNE_NO_HALO_TAP2)):
        with _lib.tuning(flag):
            outs[name] = [ops.conv2d(xd, pc, act=ops.ACT_RELU, out_dtype=od)[..., :Cout].float().cpu().permute(0, 3, 1, 2)
                          for od in (torch.float32, torch.float16)]
    scale = max(1.0, ref.abs().max().item())
    assert (outs["tap2"][0] - ref).abs().max().item() < 1e-5 * scale
    assert (outs["tap2"][0] - outs["per_tap"][0]).abs().max().item() < 1e-5 * scale
    assert (outs["tap2"][1] - outs["per_tap"][1]).abs().max().item() <= 2.0 ** -9 * scale


@pytest.mark.parametrize("shape", [(1, 80, 128, 128), (3, 80, 32, 48), (2, 4, 16, 16), (2, 8, 128, 128)])
def test_decode_matches_oracle(ops, dev, shape):
    B, C, H, W = shape
    heat = _rand_heat(B, C, H, W, seed=sum(shape))
    g = torch.Generator().manual_seed(1)
    wh = torch.rand(B, 2, H, W, generator=g) * 20
    reg = torch.rand(B, 2, H, W, generator=g)
    K = 100
    rb, rs, rc, ri = O.ctdet_decode(heat, wh, reg, down_ratio=4, K=K)
    whreg = torch.cat([nhwc(wh), nhwc(reg)], dim=3).to(dev)
    b, s, c, i = ops.decode(nhwc(heat).to(dev), whreg[..., 0:2], whreg[..., 2:4], K, 4.0, check_status=True)
    assert torch.equal(s.cpu(), rs), "scores differ"
    assert torch.equal(c.cpu(), rc), "classes differ"
    assert torch.equal(i.cpu().long(), ri), "peak indices differ"
    assert torch.allclose(b.cpu(), rb, atol=1e-4, rtol=1e-6)


def test_decode_random_shapes(ops, dev):
    """seeded sweep: odd map sizes (partial 16x16 tiles), class counts from 4 to 92, K from 1 to 300, narrow-band and wide
    heat maps -- scores / classes / peak indices bit-exact with the oracle"""
    rng = np.random.RandomState(11)
    for it in range(14):
        B = int(rng.randint(1, 4))
        C = 4 * int(rng.randint(1, 24))
        H, W = int(rng.randint(5, 70)), int(rng.randint(5, 70))
        K = int(rng.choice([1, 7, 100, 300]))
        g = torch.Generator().manual_seed(300 + it)
        spread = float(rng.choice([0.02, 0.5, 1.5]))        # 0.02: nearly constant map, every value in a few fine bins
        heat = torch.clamp(torch.sigmoid(torch.randn(B, C, H, W, generator=g) * spread - 2.19), 1e-4, 1 - 1e-4)
        wh = torch.rand(B, 2, H, W, generator=g) * 20
        reg = torch.rand(B, 2, H, W, generator=g)
        if K > C * H * W:
            continue
        rb, rs, rc, ri = O.ctdet_decode(heat, wh, reg, down_ratio=4, K=K)
        whreg = torch.cat([nhwc(wh), nhwc(reg)], dim=3).to(dev)
        b, sc, c, i = ops.decode(nhwc(heat).contiguous().to(dev), whreg[..., 0:2], whreg[..., 2:4], K, 4.0, check_status=True)
        tag = f"case {it}: B{B} C{C} {H}x{W} K{K} spread {spread}"
        assert torch.equal(sc.cpu(), rs), tag + ": scores"
        assert torch.equal(c.cpu(), rc), tag + ": classes"
        assert torch.equal(i.cpu().long(), ri), tag + ": peak indices"
        assert torch.allclose(b.cpu(), rb, atol=1e-4, rtol=1e-6), tag + ": boxes"


def test_heads_fused_random_shapes(ops, dev):
    """seeded sweep for the fused heads: class counts 1..91, 32..256 input channels (other map sizes take the layer-by-layer
    heads in the model, see CenterNet._head_outputs)"""
    rng = np.random.RandomState(23)
    for it in range(8):
        B, H, W = int(rng.randint(1, 3)), 8 * int(rng.randint(1, 5)), 16 * int(rng.randint(1, 4))   # the kernel's domain:
        Cin = int(rng.choice([32, 64, 128, 256]))                                                    # maps of 8x16 tiles
        couts = (int(rng.choice([1, 3, 20, 80, 91])), 2, 2)
        g = torch.Generator().manual_seed(800 + it)
        x = h16(torch.randn(B, Cin, H, W, generator=g))
        w1 = [h16(torch.randn(256, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5) for _ in couts]
        b1 = [torch.randn(256, generator=g) * 0.3 for _ in couts]
        w2 = [h16(torch.randn(c, 256, 1, 1, generator=g) / 16) for c in couts]
        b2 = [torch.randn(c, generator=g) for c in couts]
        acts = [ops.ACT_SIGMOID_CLAMP, ops.ACT_NONE, ops.ACT_NONE]
        ph = ops.PackedHeads([w.to(dev) for w in w1], [b.to(dev) for b in b1], [w.to(dev) for w in w2],
                             [b.to(dev) for b in b2], acts)
        outs = ops.heads_fused(nhwc(x).half().to(dev), ph, clamp=(1e-4, 1 - 1e-4))
        for i, c in enumerate(couts):
            hid = h16(F.conv2d(x, w1[i], b1[i], 1, 1).relu())
            ref = F.conv2d(hid, w2[i], b2[i])
            if i == 0:
                ref = torch.clamp(torch.sigmoid(ref), 1e-4, 1 - 1e-4)
            got = nchw(outs[i][..., :c].cpu())
            err = (got - ref).abs().max().item()
            assert err <= 3e-3 * max(1.0, ref.abs().max().item()), f"case {it} head {i}: B{B} {H}x{W} Cin{Cin} cout{c}: {err}"


def test_decode_plateau_and_sparse(ops, dev):
    """plateaus: `hmax == heat` keeps every plateau cell; ties resolve by flat NCHW index ascending.
    sparse: fewer than K positive peaks -> remaining slots have score 0."""
    B, C, H, W = 1, 4, 16, 16
    heat = torch.full((B, C, H, W), 0.01)
    heat[0, 1, 4:6, 4:6] = 0.9      # 2x2 plateau: all four kept
    heat[0, 2, 10, 10] = 0.95
    wh = torch.ones(B, 2, H, W)
    reg = torch.zeros(B, 2, H, W)
    rb, rs, rc, ri = O.ctdet_decode(heat, wh, reg, down_ratio=4, K=20)
    whreg = torch.cat([nhwc(wh), nhwc(reg)], dim=3).to(dev)
    b, s, c, i = ops.decode(nhwc(heat).to(dev), whreg[..., 0:2], whreg[..., 2:4], 20, 4.0, check_status=True)
    assert torch.equal(s.cpu(), rs) and torch.equal(c.cpu(), rc) and torch.equal(i.cpu().long(), ri)
    assert s[0, 0].item() == pytest.approx(0.95) and (s[0, 1:5].cpu() == 0.9).all()
    # sparse
    heat2 = torch.zeros(B, C, H, W)
    heat2[0, 0, 3, 3] = 0.5
    heat2[0, 3, 8, 9] = 0.7
    b, s, c, i = ops.decode(nhwc(heat2).to(dev), whreg[..., 0:2], whreg[..., 2:4], 20, 4.0, check_status=True)
    assert s[0, :2].tolist() == pytest.approx([0.7, 0.5]) and (s[0, 2:] == 0).all()
    assert c[0, :2].tolist() == [3, 0] and i[0, :2].tolist() == [8 * W + 9, 3 * W + 3]


def _trained_like(B, C, H, W, n_blobs, seed):
    """a map like a trained network's: background exactly on the 1e-4 clamp, a few Gaussian-ish blobs above it"""
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(B, C, H, W, generator=g) - 14.0
    for b in range(B):
        for _ in range(n_blobs):
            c, y, x = int(torch.randint(0, C, (1,), generator=g)), int(torch.randint(0, H, (1,), generator=g)), int(torch.randint(0, W, (1,), generator=g))
            amp = float(torch.rand(1, generator=g)) * 8 + 6
            for dy in range(-2, 3):
                for dx in range(-2, 3):
                    if 0 <= y + dy < H and 0 <= x + dx < W:
                        logits[b, c, y + dy, x + dx] += amp * 0.5 ** (dy * dy + dx * dx) + 0.01 * float(torch.rand(1, generator=g))
    return torch.clamp(torch.sigmoid(logits), 1e-4, 1 - 1e-4)


@pytest.mark.parametrize("shape,n_blobs,K", [((2, 80, 128, 128), 300, 100), ((2, 80, 128, 128), 12, 100), ((3, 20, 24, 40), 5, 100),
                                             ((1, 4, 9, 7), 2, 100), ((2, 4, 5, 5), 0, 60), ((1, 80, 128, 128), 0, 100)])
def test_decode_with_clamp_floor_equals_plain_decode(ops, dev, shape, n_blobs, K):
    """heat_floor = 1e-4 on trained-like maps (background exactly on the clamp): same scores / classes / indices / boxes as
    the oracle's plain canonical decode and as the kernel without the promise -- more than K real peaks (the floor is never
    consulted), fewer than K (the floor peaks of lowest flat index fill the rest; blobs touching the first rows make some of
    those positions non-peaks), tiny maps that run out of floor peaks too (zeros of the NMS-ed map follow), no peak at all"""
    B, C, H, W = shape
    heat = _trained_like(B, C, H, W, n_blobs, seed=sum(shape) + n_blobs)
    if n_blobs:   # one blob next to flat index 0 so that the fill has to step over non-peak floor cells
        heat[0, 0, 0, 1] = 0.3
    assert (heat == 1e-4).float().mean() > 0.5
    g = torch.Generator().manual_seed(1)
    wh = torch.rand(B, 2, H, W, generator=g) * 20
    reg = torch.rand(B, 2, H, W, generator=g)
    rb, rs, rc, ri = O.ctdet_decode(heat, wh, reg, down_ratio=4, K=K)
    whreg = torch.cat([nhwc(wh), nhwc(reg)], dim=3).to(dev)
    hm = nhwc(heat).to(dev)
    for floor in (1e-4, 0.0):
        b, s, c, i = ops.decode(hm, whreg[..., 0:2], whreg[..., 2:4], K, 4.0, check_status=True, heat_floor=floor)
        assert torch.equal(s.cpu(), rs), f"floor {floor}: scores differ"
        assert torch.equal(c.cpu(), rc), f"floor {floor}: classes differ"
        assert torch.equal(i.cpu().long(), ri), f"floor {floor}: peak indices differ"
        assert torch.allclose(b.cpu(), rb, atol=1e-4, rtol=1e-6)


def test_decode_clamp_floor_on_spread_maps_and_broken_promise(ops, dev):
    """maps that never touch the floor decode the same with the promise; a positive value below the promised floor is
    reported by the status call"""
    heat = _rand_heat(2, 80, 64, 64, seed=5)
    wh = torch.ones(2, 2, 64, 64)
    whreg = torch.cat([nhwc(wh), nhwc(torch.zeros(2, 2, 64, 64))], dim=3).to(dev)
    rb, rs, rc, ri = O.ctdet_decode(heat, wh, torch.zeros(2, 2, 64, 64), down_ratio=4, K=100)
    b, s, c, i = ops.decode(nhwc(heat).to(dev), whreg[..., 0:2], whreg[..., 2:4], 100, 4.0, check_status=True, heat_floor=1e-4)
    assert torch.equal(s.cpu(), rs) and torch.equal(c.cpu(), rc) and torch.equal(i.cpu().long(), ri)
    heat[1, 3, 10, 10] = 5e-5
    with pytest.raises(RuntimeError, match="below the promised heat_floor"):
        ops.decode(nhwc(heat).to(dev), whreg[..., 0:2], whreg[..., 2:4], 100, 4.0, check_status=True, heat_floor=1e-4)


def test_decode_big_tie_block(ops, dev):
    """a constant heatmap (every cell a peak, all equal) needs the deep radix levels: top-K = first K flat indices."""
    B, C, H, W = 2, 8, 32, 32
    heat = torch.full((B, C, H, W), 0.25)
    wh = torch.ones(B, 2, H, W)
    whreg = torch.cat([nhwc(wh), nhwc(torch.zeros(B, 2, H, W))], dim=3).to(dev)
    b, s, c, i = ops.decode(nhwc(heat).to(dev), whreg[..., 0:2], whreg[..., 2:4], 100, 4.0, check_status=True)
    assert (s == 0.25).all() and (c == 0).all()
    assert i[0].tolist() == list(range(100)) and i[1].tolist() == list(range(100))


def test_gaussian_radius_grid(ops, dev):
    hw = torch.stack(torch.meshgrid(torch.arange(1, 129), torch.arange(1, 129), indexing="ij"), -1).reshape(-1, 2)
    r, ri = ops.gaussian_radius(hw.int().to(dev))
    ref = np.array([O.gaussian_radius((int(h), int(w))) for h, w in hw.tolist()])
    assert np.array_equal(r.cpu().numpy(), ref), "f64 radius not bit-identical"
    assert np.array_equal(ri.cpu().numpy(), np.maximum(0, ref.astype(np.int64)).astype(np.int32))


def _rand_instances(B, n_max, size, C, seed):
    g = torch.Generator().manual_seed(seed)
    boxes = torch.zeros(B, n_max, 4)
    classes = torch.zeros(B, n_max, dtype=torch.int64)
    counts = torch.randint(1, n_max + 1, (B,), generator=g).int()
    for b in range(B):
        n = counts[b].item()
        wh = torch.rand(n, 2, generator=g) * 248 + 8
        ctr = torch.rand(n, 2, generator=g) * (size - wh) + wh / 2
        boxes[b, :n] = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
        classes[b, :n] = torch.randint(0, C, (n,), generator=g)
    return boxes, classes, counts


def test_gaussian_targets(ops, dev):
    B, C, size = 4, 80, 512
    boxes, classes, counts = _rand_instances(B, 32, size, C, seed=5)
    # edge cases: zero-area box, box touching the border, duplicate centre/class, class at the last index
    boxes[0, 0] = torch.tensor([10.0, 10.0, 10.0, 50.0])
    boxes[0, 1] = torch.tensor([0.0, 0.0, 37.0, 23.0])
    boxes[0, 2] = torch.tensor([400.0, 380.0, 511.9, 511.9])
    boxes[0, 3] = boxes[0, 2]
    classes[0, 3] = classes[0, 2]
    classes[0, 1] = C - 1
    counts[0] = max(counts[0].item(), 4)
    out = ops.gaussian_targets(boxes.to(dev), classes.to(dev), counts.to(dev), size // 4, size // 4, C)
    for b in range(B):
        n = counts[b].item()
        ref = O.gen_heatmap(boxes[b, :n], classes[b, :n], size // 4, size // 4, C)
        hm = nchw(out["hm"][b:b + 1].cpu())[0].numpy()
        assert np.array_equal(out["ind"][b].cpu().numpy(), ref["ind"])
        assert np.array_equal(out["reg_mask"][b].cpu().numpy(), ref["reg_mask"])
        assert np.array_equal(out["wh"][b].cpu().numpy(), ref["wh"])
        assert np.array_equal(out["reg"][b].cpu().numpy(), ref["reg"])
        assert np.array_equal(hm == 1.0, ref["hm"] == 1.0)
        assert np.array_equal(hm > 0, ref["hm"] > 0)
        assert np.abs(hm - ref["hm"]).max() <= 1e-6


def test_gaussian_targets_more_than_128(ops, dev):
    B, C = 1, 8
    g = torch.Generator().manual_seed(9)
    n = 150
    ctr = torch.rand(n, 2, generator=g) * 400 + 50
    boxes = torch.cat([ctr - 12, ctr + 12], 1).view(1, n, 4)
    classes = torch.randint(0, C, (1, n), generator=g)
    out = ops.gaussian_targets(boxes.to(dev), classes.to(dev), torch.tensor([n], dtype=torch.int32, device=dev), 128, 128, C)
    ref = O.gen_heatmap(boxes[0], classes[0], 128, 128, C)
    assert np.array_equal(out["ind"][0].cpu().numpy(), ref["ind"])
    assert np.abs(nchw(out["hm"].cpu())[0].numpy() - ref["hm"]).max() <= 1e-6


@pytest.mark.parametrize("alpha", [[1.0], [0.25]])
@pytest.mark.parametrize("with_pos", [True, False])
def test_focal_loss(ops, dev, alpha, with_pos):
    B, C, H, W = 2, 8, 32, 32
    g = torch.Generator().manual_seed(21)
    logits = (torch.randn(B, C, H, W, generator=g) * 3 - 2).requires_grad_(True)
    logits.data[0, 0, 0, 0] = 12.0   # sigmoid above the clamp: zero gradient there
    logits.data[0, 0, 0, 1] = -12.0
    gt = torch.rand(B, C, H, W, generator=g) ** 4
    if with_pos:
        gt.view(-1)[torch.randint(0, gt.numel(), (30,), generator=g)] = 1.0
        gt[0, 0, 0, 0] = 1.0
    ref = O.focal_loss_from_logits(logits, gt, alpha)
    ref.backward()
    a = torch.tensor(alpha * C if len(alpha) == 1 else alpha, dtype=torch.float32)
    loss, stats, grad = ops.focal_loss(nhwc(logits.detach()).to(dev), nhwc(gt).to(dev), a.to(dev))
    assert loss.item() == pytest.approx(ref.item(), rel=1e-5, abs=1e-6)
    assert stats[2].item() == float((gt == 1).sum())
    gref = logits.grad
    assert (nchw(grad.cpu()) - gref).abs().max() <= 1e-5 * max(1.0, gref.abs().max().item())


def test_reg_l1_loss(ops, dev):
    B, H, W, N = 3, 16, 16, 128
    g = torch.Generator().manual_seed(31)
    out = torch.randn(B, 2, H, W, generator=g, requires_grad=True)
    mask = (torch.rand(B, N, generator=g) < 0.2).to(torch.uint8)
    ind = torch.randint(0, H * W, (B, N), generator=g)
    ind[0, 1] = ind[0, 0]
    mask[0, 0] = mask[0, 1] = 1     # duplicate index: gradients accumulate
    tgt = torch.randn(B, N, 2, generator=g)
    ref = O.reg_l1_loss(out, mask, ind, tgt)
    ref.backward()
    loss, grad = ops.reg_l1_loss(nhwc(out.detach()).to(dev), mask.to(dev), ind.to(dev), tgt.to(dev))
    assert loss.item() == pytest.approx(ref.item(), rel=1e-5)
    assert (nchw(grad.cpu()) - out.grad).abs().max() < 1e-6
    # all-masked-out: loss 0 / 1e-4
    loss0, _ = ops.reg_l1_loss(nhwc(out.detach()).to(dev), torch.zeros_like(mask).to(dev), ind.to(dev), tgt.to(dev))
    assert loss0.item() == 0.0


def test_sgd_matches_torch(ops, dev):
    g = torch.Generator().manual_seed(41)
    p = torch.randn(10007, generator=g)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.SGD([pr], lr=0.01, momentum=0.9, weight_decay=1e-4)
    pd, buf = p.to(dev), torch.zeros(10007, device=dev)
    lr = torch.tensor([0.01], device=dev)
    for step in range(3):
        gr = torch.randn(10007, generator=g)
        pr.grad = gr.clone()
        opt.step()
        ops.sgd_momentum_(pd, gr.to(dev), buf, lr, 0.9, 1e-4, first_step=(step == 0))
    assert torch.allclose(pd.cpu(), pr.detach(), atol=1e-6, rtol=1e-6)


def test_sgd_runs_matches_per_run_launches(ops, dev):
    """one launch over a run table == one launch per run, bit for bit (runs of 1 element, odd lengths, two lr factors)"""
    g = torch.Generator().manual_seed(43)
    lens = [1, 255, 256, 257, 3, 70001, 1, 1, 4096, 9]
    n = sum(lens)
    ends = torch.tensor(lens).cumsum(0)
    wds = [1e-4, 0.0, 1e-4, 0.0, 0.0, 1e-4, 0.0, 1e-4, 0.0, 1e-4]
    lidx = [0, 1, 0, 0, 1, 0, 1, 1, 0, 0]
    lrt = torch.tensor([0.01, 0.02], device=dev)
    p0 = torch.randn(n, generator=g).to(dev)
    pa, pb = p0.clone(), p0.clone()
    ma, mb = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for step in range(3):
        gr = torch.randn(n, generator=g).to(dev)
        ops.sgd_momentum_runs_(pa, gr, ma, ends.to(dev), torch.tensor(lidx, dtype=torch.int32, device=dev),
                               torch.tensor(wds, device=dev), lrt, 0.9, step == 0)
        a = 0
        for r, b in enumerate(ends.tolist()):
            ops.sgd_momentum_(pb[a:b], gr[a:b], mb[a:b], lrt[lidx[r]:lidx[r] + 1], 0.9, wds[r], first_step=(step == 0))
            a = b
    assert torch.equal(pa, pb) and torch.equal(ma, mb)


def test_cpu_tensors_are_rejected(ops):
    with pytest.raises(NotImplementedError):
        ops.maxpool2x2(torch.zeros(1, 4, 4, 8))
