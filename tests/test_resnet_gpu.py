"""SURVEY 8a row a21 on the GPU: the ResNet-50 CenterNet config (`ctdet_res_50_1x.yaml`: res4 with FrozenBN, two
ConvTranspose stages, 256-channel heads) through the HIP kernels vs the CPU oracle (oracle/model_ref.py, pinned to
the reference's own modules by tests/golden/g9_resnet50.npz).  Plus the two ops this path adds."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ctdet_oracle as O
from oracle import model_ref as MR

pytestmark = pytest.mark.gpu

RES50_YAML = """
_BASE_: "./Base-CenterNet.yaml"
MODEL:
  BACKBONE:
    NAME: "build_resnet_backbone"
  CENTERNET:
    FOCAL_LOSS_ALPHA: [1]
DATASETS:
  TRAIN: ("bulb_train",)
  TEST: ("bulb_val",)
INPUT:
  FORMAT: "RGB"
  MIN_SIZE_TRAIN: (640, 672, 704, 736, 768, 800)
SOLVER:
  IMS_PER_BATCH: 2
  BASE_LR: 2.5e-4
  STEPS: (225100, 337650)
  MAX_ITER: 450200
  CHECKPOINT_PERIOD: 9004
TEST:
  EVAL_PERIOD: 18008
OUTPUT_DIR: "./output"
VERSION: 2
"""
BASE = """
MODEL:
  META_ARCHITECTURE: "CenterNet"
  PIXEL_MEAN: [0.408, 0.447, 0.470]
  PIXEL_STD: [0.289, 0.274, 0.278]
VERSION: 2
"""


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops(dev):
    from detectron2_centernet_amd import ops as _ops
    return _ops


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("tdt", [torch.float16, torch.float32])
@pytest.mark.parametrize("hw", [(16, 24), (15, 9)])
def test_maxpool3x3s2(ops, dev, tdt, hw):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 16, *hw, generator=g).half().float()
    y = ops.maxpool3x3s2(nhwc(x).to(tdt).to(dev))
    assert torch.equal(nchw(y.float().cpu()), F.max_pool2d(x, 3, 2, 1))


@pytest.mark.parametrize("mode", ["f16", "f32"])
def test_conv_transpose2d_dense(ops, dev, mode):
    g = torch.Generator().manual_seed(12)
    B, Cin, Cout, H, W = 2, 64, 32, 6, 10
    x = torch.randn(B, Cin, H, W, generator=g).half().float()
    w = (torch.randn(Cin, Cout, 4, 4, generator=g) / (Cin * 4) ** 0.5).half().float()
    scale, bias = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    ref = (F.conv_transpose2d(x, w, None, stride=2, padding=1) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)).relu()
    comp = ops.F16 if mode == "f16" else ops.F32
    tdt = torch.float16 if mode == "f16" else torch.float32
    y = ops.conv_transpose2d(nhwc(x).to(tdt).to(dev), w.to(dev), scale.to(dev), bias.to(dev), 2, 1, comp,
                             act=ops.ACT_RELU)
    got = nchw(y[..., :Cout].float().cpu())
    assert got.shape == ref.shape == (B, Cout, 2 * H, 2 * W)
    tol = 3e-3 if mode == "f16" else 1e-5
    assert (got - ref).abs().max() <= tol * max(1.0, ref.abs().max())


RES18_YAML = RES50_YAML.replace("""  CENTERNET:
    FOCAL_LOSS_ALPHA: [1]
""", """  CENTERNET:
    FOCAL_LOSS_ALPHA: [1]
    HEAD_CONV: 64
  RESNETS:
    DEPTH: 18
    RES2_OUT_CHANNELS: 64
""")


def _make(tmp_path, precision, yaml_text=None):
    from detectron2_centernet_amd.config import get_cfg
    from detectron2_centernet_amd.data.catalog import register_synthetic
    from detectron2_centernet_amd.modeling import build_model
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from weights import fill_state_dict

    (tmp_path / "Base-CenterNet.yaml").write_text(BASE)
    (tmp_path / "ctdet_res_50_1x.yaml").write_text(yaml_text or RES50_YAML)
    cfg = get_cfg()
    cfg.merge_from_file(str(tmp_path / "ctdet_res_50_1x.yaml"))
    cfg.MODEL.CENTERNET.HIP_PRECISION = precision
    register_synthetic("bulb_train", num_classes=80)
    model = build_model(cfg).eval()
    # name-keyed deterministic weights: non-trivial FrozenBN statistics; the deconv / final head weights get a
    # realistic scale (the reference initialises them with std 0.001, which would hide errors)
    sd = fill_state_dict({k: v.cpu() for k, v in model.state_dict().items()}, seed=21)
    model.load_state_dict({k: v.to(model.device) for k, v in sd.items()})
    return model, cfg, sd


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_resnet50_centernet_eval_matches_oracle(tmp_path, dev, precision):
    model, cfg, sd = _make(tmp_path, precision)
    assert model.size_divisibility == 16 and model.backbone.down_ratio == 4
    g = torch.Generator().manual_seed(5)
    img = torch.randint(0, 256, (2, 3, 96, 128), generator=g, dtype=torch.uint8)
    model.score_threshold = 0.0
    out = model([{"image": img[b]} for b in range(2)])
    eng = next(iter(model._engines.values()))
    assert eng.graph_nodes.get("kernel", 0) > 0 and set(eng.graph_nodes) <= {"kernel", "empty"}, eng.graph_nodes      # engine/graph_nodes.py: no memset / memcpy node
    hm, wh, reg = [t.float().cpu().permute(0, 3, 1, 2) for t in eng.out]
    assert hm.shape == (2, 80, 24, 32)
    x, sizes = O.preprocess([i for i in img], cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD, 16)
    sdf = {k: v.float() for k, v in sd.items()}
    with torch.no_grad():
        z = MR.centernet_resnet_forward(sdf, x)
    hm_ref = torch.clamp(torch.sigmoid(z["hm"]), 1e-4, 1 - 1e-4)
    err_hm = (hm - hm_ref).abs().max().item()
    err_wh = (wh - z["wh"]).abs().max().item() / max(1.0, z["wh"].abs().max().item())
    err_reg = (reg - z["reg"]).abs().max().item() / max(1.0, z["reg"].abs().max().item())
    print(precision, "resnet50 heatmap err", err_hm, "wh rel", err_wh, "reg rel", err_reg)
    if precision == "f32":
        assert err_hm <= 1e-5 and err_wh <= 2e-4 and err_reg <= 2e-4
    else:
        assert err_hm <= 1e-3 and err_wh <= 2e-2 and err_reg <= 2e-2
    # decode + postprocess parity on the HIP heatmap (bit-exact indices / scores)
    rb, rs, rc, _ = O.ctdet_decode(hm, wh, reg, down_ratio=4, K=100)
    for b in range(2):
        inst = out[b]["instances"]
        bb, ss, cc = O.inference_single_image(rb[b], rs[b], rc[b], 100, 0.0)
        bb, keep = O.detector_postprocess(bb, (96, 128), 96, 128)
        assert torch.equal(inst.scores.cpu(), ss[keep]) and torch.equal(inst.pred_classes.cpu(), cc[keep])
        assert torch.allclose(inst.pred_boxes.tensor.cpu(), bb[keep], atol=1e-4, rtol=1e-6)


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_resnet18_centernet_eval_matches_oracle(tmp_path, dev, precision):
    """`ctdet_res_18_1x.yaml`: BasicBlock ResNet-18, HEAD_CONV 64 (the unfused head path), deconv 256->256->256"""
    model, cfg, sd = _make(tmp_path, precision, RES18_YAML)
    keys = sorted(k for k in model.state_dict() if k.startswith(("backbone.", "deconv_layers.")))
    import os
    ref_keys = [l.split(" ")[0] for l in open(os.path.join(os.path.dirname(__file__), "golden",
                                                           "g10_resnet18_state_dict_keys.txt"))]
    assert keys == sorted(ref_keys)
    g = torch.Generator().manual_seed(6)
    img = torch.randint(0, 256, (2, 3, 80, 112), generator=g, dtype=torch.uint8)
    model.score_threshold = 0.0
    model([{"image": img[b]} for b in range(2)])
    eng = next(iter(model._engines.values()))
    hm, wh, reg = [t.float().cpu().permute(0, 3, 1, 2) for t in eng.out]
    x, _ = O.preprocess([i for i in img], cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD, 16)
    sdf = {k: v.float() for k, v in sd.items()}
    with torch.no_grad():
        y = MR.deconv_layers(sdf, "deconv_layers", MR.resnet_features(sdf, "backbone", x, blocks=(2, 2, 2), bottleneck=False))
        z = MR.centernet_heads(MR.Net(sdf), y)
    hm_ref = torch.clamp(torch.sigmoid(z["hm"]), 1e-4, 1 - 1e-4)
    err = (hm - hm_ref).abs().max().item()
    print(precision, "resnet18 heatmap err", err)
    assert err <= (1e-5 if precision == "f32" else 1e-3)
    assert (wh - z["wh"]).abs().max().item() <= (2e-4 if precision == "f32" else 2e-2) * max(1.0, z["wh"].abs().max().item())


def test_resnet50_larger_batch_is_deterministic_and_matches_oracle(tmp_path, dev):
    """8 x 384 x 512 through the ResNet-50 config: grids of several rounds of workgroups per CU for the kernels this path
    adds (7x7 s2 stem on the generic LDS-DMA kernel, strided 1x1 / 3x3, input-dilated ConvTranspose); two runs are
    bit-identical and image 0 / 7 agree with the CPU oracle"""
    model, cfg, sd = _make(tmp_path, "f16")
    g = torch.Generator().manual_seed(8)
    img = torch.randint(0, 256, (8, 3, 384, 512), generator=g, dtype=torch.uint8).to(dev)
    model.score_threshold = 0.0
    model.infer_batch_tensor(img)
    eng = next(iter(model._engines.values()))
    first = [t.clone() for t in eng.out]
    model.infer_batch_tensor(img)
    for a, b in zip(first, eng.out):
        assert torch.equal(a, b)
    hm = first[0].float().cpu().permute(0, 3, 1, 2)
    sdf = {k: v.float() for k, v in sd.items()}
    for b in (0, 7):
        x, _ = O.preprocess([img[b].cpu()], cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD, 16)
        with torch.no_grad():
            z = MR.centernet_resnet_forward(sdf, x)
        ref = torch.clamp(torch.sigmoid(z["hm"]), 1e-4, 1 - 1e-4)
        err = (hm[b:b + 1] - ref).abs().max().item()
        assert err <= 1e-3, (b, err)


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_resnet50_training_step_matches_oracle(tmp_path, dev, precision):
    """config 5's training path (ctdet_res_50_1x.yaml): frozen stem + res2 (FREEZE_AT 2), res3 / res4 with FrozenBatchNorm
    as autograd nodes, dense ConvTranspose2d 4x4 s2 + BatchNorm (batch statistics) + ReLU, 256-channel heads, the three
    losses -- against torch autograd through the oracle (pinned to the reference's own ResNet / deconv modules by G9).
    f32: losses 1e-3, gradients cos >= 0.999; f16: rounding-level agreement."""
    from detectron2_centernet_amd.data.catalog import synthetic_sample
    from detectron2_centernet_amd.structures import Boxes, Instances

    model, cfg, sd0 = _make(tmp_path, precision)
    model.train()
    frozen = [n for n, p in model.named_parameters() if not p.requires_grad]
    assert any(n.startswith("backbone.stem") for n in frozen) and any(n.startswith("backbone.res2") for n in frozen)
    assert not any(n.startswith(("backbone.res3", "backbone.res4", "deconv_layers", "hm", "wh", "reg")) for n in frozen)
    inputs = []
    for i in range(2):
        smp = synthetic_sample(i, size=128, num_classes=80, max_boxes=6)
        inst = Instances((128, 128))
        inst.gt_boxes, inst.gt_classes = Boxes(smp["boxes"]), smp["classes"]
        inputs.append({"image": smp["image"], "instances": inst})
    losses = model(inputs)
    sum(losses.values()).backward()
    sd = {k: (v.float().clone().requires_grad_(True) if v.dtype.is_floating_point and not k.startswith(("backbone.stem", "backbone.res2"))
              and "running" not in k and ".norm." not in k else v.float().clone()) for k, v in sd0.items()}
    x_ref, _ = O.preprocess([d["image"] for d in inputs], cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD, 16)
    z = MR.centernet_resnet_forward(sd, x_ref, training=True)
    targets = [O.gen_heatmap(d["instances"].gt_boxes.tensor, d["instances"].gt_classes, 32, 32, 80) for d in inputs]
    ref = MR.centernet_losses(z, targets, [1.0])
    sum(ref.values()).backward()
    ltol = 1e-3 if precision == "f32" else 1e-2
    for k in ("hm_loss", "wh_loss", "off_loss"):
        got, want = losses[k].item(), ref[k].item()
        print(precision, k, got, want)
        assert abs(got - want) <= ltol * max(1.0, abs(want)), (k, got, want)
    worst, name_of = 1.0, ""
    for name, p in model.named_parameters():
        if not p.requires_grad:
            assert p.grad is None
            continue
        gref = sd[name].grad
        assert p.grad is not None and gref is not None, name
        if gref.abs().max() == 0:
            continue
        cos = torch.nn.functional.cosine_similarity(p.grad.float().cpu().flatten(), gref.flatten(), dim=0).item()
        ratio = (p.grad.float().cpu().norm() / gref.norm()).item()
        if cos < worst:
            worst, name_of = cos, name
        assert (0.99 if precision == "f32" else 0.9) < ratio < (1.01 if precision == "f32" else 1.1), (name, ratio)
    print(precision, "worst gradient cosine", worst, name_of)
    assert worst >= (0.999 if precision == "f32" else 0.98), (worst, name_of)
    bn = model.deconv_layers[1]
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("precision", ["f32", "f16x3", "f16"])
def test_export_split_matches_reference_inference_g15(tmp_path, dev, precision):
    """the serving split `export.CenterNetModel.inference` ({images} -> {hm after sigmoid + clamp, wh,
    reg}) against what the REFERENCE's own export module (detectron2/export/meta_modeling.py:151-201) computed for a
    ResNet-18 CenterNet built from the reference's modules with the same name-keyed weights (G15; input = G10's)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from weights import fill_state_dict
    from detectron2_centernet_amd.export import CenterNetModel

    model, cfg, _ = _make(tmp_path, precision, RES18_YAML)
    msd = {k: v.cpu() for k, v in model.state_dict().items()}
    sd = {}
    for prefix, seed in (("backbone.", 11), ("deconv_layers.", 12)):
        part = fill_state_dict({k[len(prefix):]: v for k, v in msd.items() if k.startswith(prefix)}, seed=seed)
        sd.update({prefix + k: v for k, v in part.items()})
    sd.update(fill_state_dict({k: v for k, v in msd.items() if k.split(".")[0] in ("hm", "wh", "reg")}, seed=15))
    missing = model.load_state_dict({k: v.to(dev) for k, v in sd.items()}, strict=False)
    assert set(missing.missing_keys) <= {"pixel_mean", "pixel_std"} and not missing.unexpected_keys
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g15_export_split.npz"))
    em = CenterNetModel(cfg, model)
    assert em.get_input_names() == ["images", "im_info"] and em.get_output_names() == list(d["output_names"])
    res = em.inference({"images": torch.from_numpy(d["x"]).to(dev)})
    tol = 1e-4 if precision != "f16" else 5e-3      # measured 2.8e-5 (f32 and f16x3 alike: summation order vs torch-CPU)
    for k in ("hm", "wh", "reg"):
        ref = torch.from_numpy(d[k])
        got = res[k].float().cpu()[:, :ref.shape[1]]
        assert got.shape == ref.shape
        err = (got - ref).abs().max().item()
        assert ref.std().item() > 1e-2 and err <= tol * max(1.0, ref.abs().max().item()), (k, err)
