"""Pins the CPU oracle (oracle/) against golden vectors produced by the reference's own Python functions
(tests/golden/make_golden.py, run in the build container against /root/reference).  CPU only.

The reference has no tests for this path, so these vectors -- outputs of the reference code
itself -- are what anchors parity.  DCNv2 arithmetic is the one part with no runnable reference ("parity unpinned").
"""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import ctdet_oracle as O
from oracle import model_ref as MR

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, G)
from weights import fill_state_dict  # noqa: E402


def load(name):
    return np.load(os.path.join(G, name))


def test_g1_gaussian_radius_bit_exact():
    d = load("g1_gaussian_radius.npz")
    got = np.array([O.gaussian_radius((int(h), int(w))) for h, w in zip(d["h"], d["w"])])
    assert np.array_equal(got, d["radius"])


def test_g2_gen_heatmap_bit_exact():
    d = load("g2_gen_heatmap.npz")
    for i in range(int(d["n_cases"])):
        r = O.gen_heatmap(d[f"boxes{i}"], d[f"classes{i}"], 128, 128, 80)
        hm = np.zeros(80 * 128 * 128, dtype=np.float32)
        hm[d[f"hm_idx{i}"]] = d[f"hm_val{i}"]
        assert np.array_equal(r["hm"].ravel(), hm), f"case {i}: heatmap differs"
        for k in ("wh", "reg", "ind", "reg_mask"):
            assert np.array_equal(r[k], d[f"{k}{i}"]), f"case {i}: {k} differs"
            assert r[k].dtype == d[f"{k}{i}"].dtype


def test_g3_neg_loss_value_and_grad():
    d = load("g3_neg_loss.npz")
    for i in range(int(d["n_cases"])):
        logits = torch.from_numpy(d[f"logits{i}"]).requires_grad_(True)
        loss = O.focal_loss_from_logits(logits, torch.from_numpy(d[f"gt{i}"]), d[f"alpha{i}"].tolist())
        loss.backward()
        assert np.array_equal(loss.detach().numpy(), d[f"loss{i}"])
        assert np.array_equal(logits.grad.numpy(), d[f"grad{i}"])


def test_g4_reg_l1():
    d = load("g4_reg_l1.npz")
    out = torch.from_numpy(d["output"]).requires_grad_(True)
    loss = O.reg_l1_loss(out, torch.from_numpy(d["mask"]), torch.from_numpy(d["ind"]), torch.from_numpy(d["target"]))
    loss.backward()
    assert np.array_equal(loss.detach().numpy(), d["loss"])
    assert np.array_equal(out.grad.numpy(), d["grad"])


def test_g5_decode_bit_exact_and_postprocess():
    d = load("g5_decode.npz")
    for i in range(int(d["n_cases"])):
        heat, wh, reg = (torch.from_numpy(d[f"{k}{i}"]) for k in ("heat", "wh", "reg"))
        b, s, c, _ = O.ctdet_decode(heat, wh, reg, down_ratio=4, K=100)
        assert np.array_equal(s[0].numpy(), d[f"scores{i}"])
        assert np.array_equal(c[0].numpy(), d[f"classes{i}"]) and c.dtype == torch.int32
        assert np.array_equal(b[0].numpy(), d[f"boxes{i}"])
        H, W = heat.shape[2:]
        bb, ss, cc = O.inference_single_image(b[0], s[0], c[0], 50, float(d[f"pp_thresh{i}"]))
        bb, keep = O.detector_postprocess(bb, (H * 4, W * 4), H * 8, W * 6)
        assert np.array_equal(bb[keep].numpy(), d[f"pp_boxes{i}"])
        assert np.array_equal(ss[keep].numpy(), d[f"pp_scores{i}"])
        assert np.array_equal(cc[keep].numpy(), d[f"pp_classes{i}"])
        assert len(d[f"pp_scores{i}"]) > 0
    kept = O.nms_keep(torch.from_numpy(d["plateau_heat"]))
    assert np.array_equal(kept.numpy(), d["plateau_kept"])
    assert (kept[0, 1, 4:6, 4:6] == 0.9).all()  # every plateau cell survives `hmax == heat`


def test_g6_preprocess_and_padding():
    d = load("g6_preprocess.npz")
    out, sizes = O.preprocess([torch.from_numpy(d["img0"]), torch.from_numpy(d["img1"])], [0.408, 0.447, 0.470],
                              [0.289, 0.274, 0.278], 32)
    assert np.array_equal(out.numpy(), d["batch"])
    assert [list(s) for s in sizes] == d["sizes"].tolist()


def _golden_state_dict():
    shapes = {}
    with open(os.path.join(G, "g8_dla34_state_dict_keys.txt")) as f:
        for line in f:
            k, shp = line.split(" ", 1)
            shapes[k] = eval(shp)
    sd = {k: torch.zeros(s) for k, s in shapes.items()}
    # the bilinear up-conv weights are kept by fill_state_dict, so they must be the reference initialiser's
    d = load("g7_dla34.npz")
    for k, s in shapes.items():
        if "up_" in k and len(s) == 4 and s[1] == 1:
            f = s[2] // 2
            sd[k] = torch.from_numpy(d[f"up_w{f}"])[:1].repeat(s[0], 1, 1, 1)
    return fill_state_dict(sd, seed=7), shapes


def test_g7_dla34_topology_matches_reference_modules():
    """the functional oracle reproduces the reference's DLA-34 module graph (base levels, Tree/Root recursion,
    DLAUp/IDAUp ordering, depthwise up-convs); DCN slots use this repo's CPU DCNv2 on both sides."""
    d = load("g7_dla34.npz")
    sd, _ = _golden_state_dict()
    n = MR.Net({"backbone." + k: v for k, v in sd.items()})
    x = torch.from_numpy(d["x"])
    with torch.no_grad():
        base = MR.dla_base(n, "backbone.base", x, (1, 1, 1, 2, 2, 1))
        y = MR.dla34(n, "backbone", x)
    for i, m in enumerate(base):
        ref = d[f"base{i}"]
        assert m.shape == ref.shape
        assert np.abs(m.numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), f"base level {i}"
    for i, m in enumerate(y):
        ref = d[f"y{i}"]
        assert np.abs(m.numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), f"ida_up output {i}"


def test_g7_fill_up_weights():
    from detectron2_centernet_amd.modeling.backbone.dla import fill_up_weights

    d = load("g7_dla34.npz")
    for f in (2, 4):
        up = torch.nn.ConvTranspose2d(4, 4, f * 2, stride=f, padding=f // 2, groups=4, bias=False)
        fill_up_weights(up)
        assert np.array_equal(up.weight.detach().numpy(), d[f"up_w{f}"])


def test_g8_state_dict_keys_match_reference():
    """checkpoint compatibility: the build's DLA34 exposes exactly the reference's keys and shapes."""
    from detectron2_centernet_amd.config import get_cfg
    from detectron2_centernet_amd.modeling.backbone.dla import DLA34

    _, shapes = _golden_state_dict()
    model = DLA34(get_cfg())
    mine = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert mine == shapes


def test_g9_resnet50_res4_and_deconv_match_reference():
    """oracle restatement of ResNet-50 (FrozenBN, stride in 1x1) + the CenterNet deconv layers vs the outputs of the
    reference's own modules (tests/golden/make_golden.py G9), and the state-dict key list"""
    import types

    import torch

    from oracle import model_ref as MR
    from weights import fill_state_dict

    g = np.load(os.path.join(G, "g9_resnet50.npz"))
    shapes = {}
    for line in open(os.path.join(G, "g9_resnet50_state_dict_keys.txt")):
        k, shp = line.split(" ", 1)
        shapes[k] = tuple(int(v) for v in shp.strip().strip("()").split(",") if v.strip())
    bb = {k[len("backbone."):]: torch.zeros(s) for k, s in shapes.items() if k.startswith("backbone.")}
    dc = {k[len("deconv_layers."):]: torch.zeros(s) for k, s in shapes.items() if k.startswith("deconv_layers.")}
    sd = {"backbone." + k: v for k, v in fill_state_dict(bb, seed=9).items()}
    sd.update({"deconv_layers." + k: v for k, v in fill_state_dict(dc, seed=10).items()})
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        res4 = MR.resnet_features(sd, "backbone", x)
        up = MR.deconv_layers(sd, "deconv_layers", res4)
    assert res4.shape == g["res4"].shape and up.shape == g["up"].shape
    assert np.abs(res4.numpy() - g["res4"]).max() <= 1e-5 * max(1.0, np.abs(g["res4"]).max())
    assert np.abs(up.numpy() - g["up"]).max() <= 1e-5 * max(1.0, np.abs(g["up"]).max())


def test_g10_resnet18_res4_and_deconv_match_reference():
    """the BasicBlock ResNet of the ctdet_res_18 / 34 configs (G10, from the reference's own modules)"""
    from oracle import model_ref as MR

    g = np.load(os.path.join(G, "g10_resnet18.npz"))
    shapes = {}
    for line in open(os.path.join(G, "g10_resnet18_state_dict_keys.txt")):
        k, shp = line.split(" ", 1)
        shapes[k] = tuple(int(v) for v in shp.strip().strip("()").split(",") if v.strip())
    bb = {k[len("backbone."):]: torch.zeros(s) for k, s in shapes.items() if k.startswith("backbone.")}
    dc = {k[len("deconv_layers."):]: torch.zeros(s) for k, s in shapes.items() if k.startswith("deconv_layers.")}
    sd = {"backbone." + k: v for k, v in fill_state_dict(bb, seed=11).items()}
    sd.update({"deconv_layers." + k: v for k, v in fill_state_dict(dc, seed=12).items()})
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        res4 = MR.resnet_features(sd, "backbone", x, blocks=(2, 2, 2), bottleneck=False)
        up = MR.deconv_layers(sd, "deconv_layers", res4)
    assert np.abs(res4.numpy() - g["res4"]).max() <= 1e-5 * max(1.0, np.abs(g["res4"]).max())
    assert np.abs(up.numpy() - g["up"]).max() <= 1e-5 * max(1.0, np.abs(g["up"]).max())


# ---------------------------------------------------------------------------------------------------------------
# G11: hand-derived known answers for DCNv2 (tests/golden/make_dcn_known_answers.py) -- the only independent anchor of
# the "parity unpinned" DCN oracle -- and the kernel-by-kernel restatement of the backward against autograd of the forward
def _g11():
    return np.load(os.path.join(G, "g11_dcn_known_answers.npz"))


G11_CASES = ["zero", "negzero", "int_1_0", "int_0_m2", "int_m1_3", "int_2_2", "half_h", "half_w_masked", "border"]


@pytest.mark.parametrize("case", G11_CASES)
def test_g11_dcn_known_answers_forward_and_backward(case):
    g = _g11()
    x, w, bias, go = (torch.from_numpy(g[k]) for k in ("x", "w", "bias", "grad_out"))
    off, mask = torch.from_numpy(g[f"{case}_offset"]), torch.from_numpy(g[f"{case}_mask"])
    out = O.dcnv2_forward(x, off, mask, w, bias, 1, 1, 1)
    assert torch.allclose(out, torch.from_numpy(g[f"{case}_out"]), rtol=0, atol=1e-12), case
    gx, goff, gm, gw, gb = O.dcnv2_backward(x, off, mask, w, go)
    for key, got in (("grad_input", gx), ("grad_weight", gw), ("grad_bias", gb)):
        if f"{case}_{key}" in g.files:
            assert torch.allclose(got, torch.from_numpy(g[f"{case}_{key}"]), rtol=0, atol=1e-11), (case, key)
    if f"{case}_grad_offset_h" in g.files:
        assert torch.allclose(goff[:, 0::2], torch.from_numpy(g[f"{case}_grad_offset_h"]), rtol=0, atol=1e-11), case


def test_dcn_backward_restatement_matches_autograd_of_forward():
    """deform_conv_cuda_kernel.cu:871-1066 restated line by line (oracle.dcnv2_backward) == torch autograd through the
    restated forward, away from the measure-zero kinks (integer coordinates); incl. samples far outside the image"""
    gen = torch.Generator().manual_seed(5)
    for B, C, Co, H, W, std in ((2, 5, 4, 7, 9, 2.0), (1, 8, 3, 5, 6, 6.0), (1, 3, 2, 4, 4, 0.3)):
        x = torch.randn(B, C, H, W, generator=gen, dtype=torch.float64, requires_grad=True)
        off = (torch.randn(B, 18, H, W, generator=gen, dtype=torch.float64) * std).requires_grad_(True)
        mask = torch.sigmoid(torch.randn(B, 9, H, W, generator=gen, dtype=torch.float64)).requires_grad_(True)
        w = torch.randn(Co, C, 3, 3, generator=gen, dtype=torch.float64, requires_grad=True)
        bias = torch.randn(Co, generator=gen, dtype=torch.float64, requires_grad=True)
        go = torch.randn(B, Co, H, W, generator=gen, dtype=torch.float64)
        O.dcnv2_forward(x, off, mask, w, bias, 1, 1, 1).backward(go)
        got = O.dcnv2_backward(x.detach(), off.detach(), mask.detach(), w.detach(), go)
        for name, a, b in zip(("input", "offset", "mask", "weight", "bias"), got, (x.grad, off.grad, mask.grad, w.grad, bias.grad)):
            assert torch.allclose(a, b, rtol=0, atol=1e-11), name


def test_dcn_backward_sentinel_and_truncation_cases():
    """the branches autograd never exercises as such: the `inv = -2` sentinel of the coordinate kernel (:1027-1029) gives
    exactly zero offset / mask gradients for samples outside (-1, H) x (-1, W); `(int)` truncation toward zero in col2im
    (:927-928) for coordinates in (-1, 0): cur = 0, the dy = -1 neighbour is skipped by the bounds test and the weight of
    pixel 0 is 1 + h_im"""
    x = torch.arange(1, 13, dtype=torch.float64).reshape(1, 1, 3, 4)
    w = torch.zeros(1, 1, 3, 3, dtype=torch.float64)
    w[0, 0, 1, 1] = 2.0                     # only the centre tap (k = 4) contributes
    off = torch.zeros(1, 18, 3, 4, dtype=torch.float64)
    mask = torch.ones(1, 9, 3, 4, dtype=torch.float64)
    off[0, 8, 0, 0] = -0.25                 # centre tap of pixel (0, 0): h_im = -0.25
    off[0, 8, 2, 3] = 5.0                   # centre tap of pixel (2, 3): h_im = 7 -> outside
    off[0, 9, 1, 1] = -9.0                  # centre tap of pixel (1, 1): w_im = -8 -> outside
    go = torch.ones(1, 1, 3, 4, dtype=torch.float64)
    out = O.dcnv2_forward(x, off, mask, w, None, 1, 1, 1)
    assert out[0, 0, 0, 0].item() == pytest.approx(2.0 * 0.75 * 1.0)        # (1 - 0.25) * x[0, 0], upper corner row -1 -> 0
    assert out[0, 0, 2, 3].item() == 0.0 and out[0, 0, 1, 1].item() == 0.0
    gx, goff, gm, gw, gb = O.dcnv2_backward(x, off, mask, w, go, with_bias=False)
    assert gb is None
    assert goff[0, 8, 2, 3].item() == 0.0 and goff[0, 9, 2, 3].item() == 0.0 and gm[0, 4, 2, 3].item() == 0.0
    assert goff[0, 8, 1, 1].item() == 0.0 and goff[0, 9, 1, 1].item() == 0.0 and gm[0, 4, 1, 1].item() == 0.0
    # pixel (0, 0) of the input receives 2 * 0.75 from output (0, 0) (truncated cur_h = 0, weight h + 1 - h_im with h = -1
    # excluded) on top of nothing else (the outside samples scatter nothing)
    assert gx[0, 0, 0, 0].item() == pytest.approx(2.0 * 0.75)
    assert gx[0, 0, 2, 3].item() == 0.0 and gx[0, 0, 1, 1].item() == 0.0
    # d out / d h at h_im = -0.25: x[0,0] * (+1) (only the lower corner exists): coordinate weight = + v3 ... here
    # low = -1 (guarded), high = 0: weight = (w_low + 1 - w) * x[high, low] = x[0, 0]
    assert goff[0, 8, 0, 0].item() == pytest.approx(2.0 * 1.0)


def test_g12_vovnet_oracle_matches_reference_module():
    """oracle.model_ref.vovnet_features == the reference's own VoVNet (V-19-slim-eSE, FrozenBN) on a 70x100 input
    (ceil-mode pooling, eSE attention, OSA concat), and the local module's state-dict keys == the reference's"""
    g = np.load(os.path.join(G, "g12_vovnet19slim.npz"))
    keys = [l.split(" ")[0] for l in open(os.path.join(G, "g12_vovnet19slim_state_dict_keys.txt"))]
    shapes = {l.split(" ")[0]: eval(l.split(" ", 1)[1]) for l in open(os.path.join(G, "g12_vovnet19slim_state_dict_keys.txt"))}
    sd = fill_state_dict({k[len("backbone."):]: torch.zeros(shapes[k]) for k in keys}, seed=13)
    sd = {"backbone." + k: v for k, v in sd.items()}
    outs = MR.vovnet_features(sd, "backbone", torch.from_numpy(g["x"]))
    for name in ("stage2", "stage3", "stage4", "stage5"):
        ref = torch.from_numpy(g[name])
        assert outs[name].shape == ref.shape, name
        assert (outs[name] - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item()), name
    # the local VoVNet has the same parameters / buffers under the same names
    from types import SimpleNamespace as NS
    from detectron2_centernet_amd.modeling.backbone.vovnet import VoVNet
    cfg = NS(MODEL=NS(VOVNET=NS(NORM="FrozenBN", CONV_BODY="V-19-slim-eSE"), BACKBONE=NS(FREEZE_AT=0)))
    m = VoVNet(cfg, 3, out_features=["stage2", "stage3", "stage4", "stage5"])
    mine = {"backbone." + k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert mine == shapes
