"""VoVNet-v2 (eSE) CenterNet configs: `ctdet_vovnet2_19_slim_1x.yaml` ingested unchanged, eval forward on
the HIP kernels against the CPU oracle (pinned to the reference's own VoVNet module by G12), and the new pieces (ceil-mode
3x3/2 max pool, eSE attention) against torch."""
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import ctdet_oracle as O
from oracle import model_ref as MR

pytestmark = pytest.mark.gpu

VOV_YAML = """
_BASE_: "./Base-CenterNet.yaml"
MODEL:
  BACKBONE:
    NAME: "build_vovnet_backbone"
  WEIGHTS: "/autox-sz/users/chenxiaoniu/models/vovnet19_ese_slim_detectron2.pth"
  VOVNET:
    OUT_FEATURES: ["stage2", "stage3", "stage4", "stage5"]
    CONV_BODY: "V-19-slim-eSE"
  CENTERNET:
    HEAD_CONV: 64
    FOCAL_LOSS_ALPHA: [1]
DATASETS:
  TRAIN: ("bulb_train",)
  TEST: ("bulb_val",)
INPUT:
  FORMAT: "RGB"
  MIN_SIZE_TRAIN: (640, 672, 704, 736, 768, 800)
SOLVER:
  IMS_PER_BATCH: 24
  BASE_LR: 2.5e-4
VERSION: 2
"""
BASE = """
MODEL:
  META_ARCHITECTURE: "CenterNet"
  PIXEL_MEAN: [0.408, 0.447, 0.470]
  PIXEL_STD: [0.289, 0.274, 0.278]
VERSION: 2
"""


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("tdt", [torch.float16, torch.float32])
def test_vovnet_pool_and_ese_kernels(dev, tdt):
    import detectron2_centernet_amd.ops as ops
    g = torch.Generator().manual_seed(3)
    for H, W in ((17, 25), (16, 24), (35, 50), (3, 3), (4, 7)):
        x = torch.randn(2, 16, H, W, generator=g).half().float()
        y = ops.maxpool3x3s2_ceil(nhwc(x).to(tdt).to(dev))
        ref = F.max_pool2d(x, kernel_size=3, stride=2, ceil_mode=True)
        assert tuple(y.shape) == (2, ref.shape[2], ref.shape[3], 16), (H, W, y.shape, ref.shape)
        assert torch.equal(nchw(y.float().cpu()), ref), (H, W)
    x = torch.randn(3, 80, 9, 13, generator=g).half().float()
    idn = torch.randn(3, 80, 9, 13, generator=g).half().float()
    s = torch.randn(3, 80, generator=g) * 3
    pooled = ops.global_avgpool(nhwc(x).to(tdt).to(dev))
    assert torch.allclose(pooled.cpu(), x.mean((2, 3)), atol=1e-5)
    ref = x * (F.relu6(s + 3) / 6).view(3, 80, 1, 1)
    tol = 2e-3 if tdt == torch.float16 else 1e-6
    out = ops.ese_scale(nhwc(x).to(tdt).to(dev), s.to(dev))
    assert (nchw(out.float().cpu()) - ref).abs().max() <= tol * ref.abs().max()
    out = ops.ese_scale(nhwc(x).to(tdt).to(dev), s.to(dev), identity=nhwc(idn).to(tdt).to(dev))
    assert (nchw(out.float().cpu()) - (ref + idn)).abs().max() <= tol * (ref + idn).abs().max()


@pytest.mark.parametrize("tdt", [torch.float16, torch.float32])
def test_vovnet_pool_and_ese_backward(dev, tdt):
    """the training-side nodes of the VoVNet pieces (round 4: both directions on HIP kernels): MaxPool2d(3, 2, ceil_mode=True)
    and F.max_pool2d(x, 3, 2, 1) backward against torch's (ties included: the map is quantised so that windows hold equal
    maxima and the first one in scan order must take the gradient), eSE forward + backward against torch autograd of
    x * hsigmoid(fc(mean(x))) + identity, gradients of x, identity, fc.weight and fc.bias"""
    from detectron2_centernet_amd import ops_train
    g = torch.Generator().manual_seed(5)
    for ceil in (True, False):
        for H, W in ((17, 25), (16, 24), (5, 7), (3, 3)):
            x = (torch.randn(2, 16, H, W, generator=g) * 2).round().div(2)      # many ties
            xr = x.clone().requires_grad_(True)
            ref = F.max_pool2d(xr, 3, 2, ceil_mode=True) if ceil else F.max_pool2d(xr, 3, 2, 1)
            dz = torch.randn(ref.shape, generator=g).half().float()
            ref.backward(dz)
            xh = nhwc(x).to(tdt).to(dev).requires_grad_(True)
            y = ops_train.MaxPool3x3s2Fn.apply(xh, ceil)
            assert torch.equal(nchw(y.detach().float().cpu()), ref.detach()), (ceil, H, W)
            y.backward(nhwc(dz).to(tdt).to(dev))
            got = nchw(xh.grad.float().cpu())
            assert torch.allclose(got, xr.grad, atol=2e-3 if tdt == torch.float16 else 1e-6), (ceil, H, W, (got - xr.grad).abs().max())
    B, Cc, H, W = 3, 80, 9, 13
    x = torch.randn(B, Cc, H, W, generator=g).half().float()
    idn = torch.randn(B, Cc, H, W, generator=g).half().float()
    fw = (torch.randn(Cc, Cc, 1, 1, generator=g) * 0.5)
    fb = torch.randn(Cc, generator=g)
    dy = torch.randn(B, Cc, H, W, generator=g).half().float()
    for with_id in (False, True):
        xr, ir, wr, br = (t.clone().requires_grad_(True) for t in (x, idn, fw, fb))
        s = F.conv2d(xr.mean((2, 3), keepdim=True), wr, br)
        ref = xr * (F.relu6(s + 3) / 6) + (ir if with_id else 0)
        ref.backward(dy)
        xh, ih = (nhwc(t).to(tdt).to(dev).requires_grad_(True) for t in (x, idn))
        wh, bh = fw.to(dev).requires_grad_(True), fb.to(dev).requires_grad_(True)
        mult = ops_train.PARAM_GRAD_MULT
        y = ops_train.EseFn.apply(xh, wh, bh, ih if with_id else None)
        tol = 3e-3 if tdt == torch.float16 else 2e-6
        assert (nchw(y.detach().float().cpu()) - ref.detach()).abs().max() <= tol * ref.abs().max()
        y.backward(nhwc(dy).to(tdt).to(dev))
        assert (nchw(xh.grad.float().cpu()) - xr.grad).abs().max() <= tol * xr.grad.abs().max(), with_id
        if with_id:
            assert torch.equal(nchw(ih.grad.float().cpu()), dy)
        gtol = 2e-2 if tdt == torch.float16 else 1e-5
        assert (wh.grad.cpu() / mult - wr.grad).abs().max() <= gtol * wr.grad.abs().max()
        assert (bh.grad.cpu() / mult - br.grad).abs().max() <= gtol * br.grad.abs().max()


@pytest.mark.parametrize("precision", ["f32", "f16x3", "f16"])
def test_vovnet19_slim_centernet_eval_matches_oracle(tmp_path, dev, precision):
    import sys
    from detectron2_centernet_amd.config import get_cfg
    from detectron2_centernet_amd.data.catalog import register_synthetic
    from detectron2_centernet_amd.modeling import build_model
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from weights import fill_state_dict

    (tmp_path / "Base-CenterNet.yaml").write_text(BASE)
    (tmp_path / "ctdet_vovnet2_19_slim_1x.yaml").write_text(VOV_YAML)
    cfg = get_cfg()
    cfg.merge_from_file(str(tmp_path / "ctdet_vovnet2_19_slim_1x.yaml"))
    cfg.MODEL.CENTERNET.HIP_PRECISION = precision
    register_synthetic("bulb_train", num_classes=80)
    model = build_model(cfg).eval()
    assert model.backbone_type == "vovnet" and model.size_divisibility == 16 and model.head_conv == 64
    keys = sorted(k for k in model.state_dict() if k.startswith("backbone."))
    ref_keys = [l.split(" ")[0] for l in open(os.path.join(os.path.dirname(__file__), "golden",
                                                           "g12_vovnet19slim_state_dict_keys.txt"))]
    assert keys == sorted(ref_keys)
    sd = fill_state_dict({k: v.cpu() for k, v in model.state_dict().items()}, seed=31)
    model.load_state_dict({k: v.to(model.device) for k, v in sd.items()})
    g = torch.Generator().manual_seed(6)
    img = torch.randint(0, 256, (2, 3, 90, 120), generator=g, dtype=torch.uint8)      # padded to 96 x 128
    model.score_threshold = 0.0
    out = model([{"image": img[b]} for b in range(2)])
    eng = next(iter(model._engines.values()))
    assert eng.graph_nodes.get("kernel", 0) > 0 and set(eng.graph_nodes) <= {"kernel", "empty"}, eng.graph_nodes      # engine/graph_nodes.py
    hm, wh, reg = [t.float().cpu().permute(0, 3, 1, 2) for t in eng.out]
    assert hm.shape == (2, 80, 24, 32)
    x, _ = O.preprocess([i for i in img], cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD, 16)
    sdf = {k: v.float() for k, v in sd.items()}
    with torch.no_grad():
        z = MR.centernet_vovnet_forward(sdf, x)
    hm_ref = torch.clamp(torch.sigmoid(z["hm"]), 1e-4, 1 - 1e-4)
    err = (hm - hm_ref).abs().max().item()
    print(precision, "vovnet19-slim heatmap err", err)
    assert err <= (1e-3 if precision == "f16" else 5e-5)
    assert (wh - z["wh"]).abs().max().item() <= (2e-2 if precision == "f16" else 2e-4) * max(1.0, z["wh"].abs().max().item())
    rb, rs, rc, _ = O.ctdet_decode(hm, wh, reg, down_ratio=4, K=100)
    for b in range(2):
        inst = out[b]["instances"]
        bb, ss, cc = O.inference_single_image(rb[b], rs[b], rc[b], 100, 0.0)
        bb, keep = O.detector_postprocess(bb, (90, 120), 90, 120)
        assert torch.equal(inst.scores.cpu(), ss[keep]) and torch.equal(inst.pred_classes.cpu(), cc[keep])


V39_YAML = """
_BASE_: "./Base-CenterNet.yaml"
MODEL:
  BACKBONE:
    NAME: "build_vovnet_backbone"
  WEIGHTS: "/autox-sz/users/chenxiaoniu/models/vovnet39_ese_detectron2.pth"
  VOVNET:
    OUT_FEATURES: ["stage2", "stage3", "stage4", "stage5"]
  CENTERNET:
    FOCAL_LOSS_ALPHA: [1]
DATASETS:
  TRAIN: ("bulb_train",)
  TEST: ("bulb_val",)
INPUT:
  FORMAT: "RGB"
  MIN_SIZE_TRAIN: (640, 672, 704, 736, 768, 800)
SOLVER:
  IMS_PER_BATCH: 24
  BASE_LR: 2.5e-4
  STEPS: (136500, 273000)
  MAX_ITER: 364000
  CHECKPOINT_PERIOD: 3640
TEST:
  EVAL_PERIOD: 1820
OUTPUT_DIR: "./output/centernet-vovnet39-bulb"
VERSION: 2
"""


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_vovnet39_centernet_eval_matches_oracle(tmp_path, dev, precision):
    """`ctdet_vovnet2_39_1x.yaml` (text unchanged): VoVNet-39-eSE (the default CONV_BODY: 5 concat layers per block, 1/1/2/2
    blocks, 256-channel heads on stage4 -> two deconv stages) through the HIP kernels against the oracle whose VoVNet code is
    pinned to the reference module by G12; heat map within 5e-5, decode of the HIP heat map bit-exact"""
    import sys
    from detectron2_centernet_amd.config import get_cfg
    from detectron2_centernet_amd.data.catalog import register_synthetic
    from detectron2_centernet_amd.modeling import build_model
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from weights import fill_state_dict

    (tmp_path / "Base-CenterNet.yaml").write_text(BASE)
    (tmp_path / "ctdet_vovnet2_39_1x.yaml").write_text(V39_YAML)
    cfg = get_cfg()
    cfg.merge_from_file(str(tmp_path / "ctdet_vovnet2_39_1x.yaml"))
    cfg.MODEL.CENTERNET.HIP_PRECISION = precision
    register_synthetic("bulb_train", num_classes=80)
    model = build_model(cfg).eval()
    assert model.backbone_type == "vovnet" and cfg.MODEL.VOVNET.CONV_BODY == "V-39-eSE"
    sd = fill_state_dict({k: v.cpu() for k, v in model.state_dict().items()}, seed=39)
    model.load_state_dict({k: v.to(model.device) for k, v in sd.items()})
    g = torch.Generator().manual_seed(7)
    img = torch.randint(0, 256, (2, 3, 96, 128), generator=g, dtype=torch.uint8)
    model.score_threshold = 0.0
    model([{"image": img[b]} for b in range(2)])
    eng = next(iter(model._engines.values()))
    hm, wh, reg = [t.float().cpu().permute(0, 3, 1, 2) for t in eng.out]
    x, _ = O.preprocess([i for i in img], cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD, 16)
    with torch.no_grad():
        z = MR.centernet_vovnet_forward({k: v.float() for k, v in sd.items()}, x, body="V-39-eSE")
    hm_ref = torch.clamp(torch.sigmoid(z["hm"]), 1e-4, 1 - 1e-4)
    assert hm_ref.std().item() > 3e-3
    assert (hm - hm_ref).abs().max().item() <= 5e-5
    rb, rs, rc, ri = O.ctdet_decode(hm, wh, reg, down_ratio=4, K=100)
    boxes, scores, classes, inds = [t.cpu() for t in eng.dec]
    assert torch.equal(scores, rs) and torch.equal(classes, rc) and torch.equal(inds.long(), ri)


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_vovnet19_slim_training_step_matches_oracle(tmp_path, dev, precision):
    """the VoVNet configs' training path (`ctdet_vovnet2_19_slim_1x.yaml`): frozen stem + stage2 (FREEZE_AT 2) on the inference
    kernels, stage3 / stage4 with FrozenBatchNorm as autograd nodes (3x3 and concat 1x1 convs on the HIP kernels, eSE attention
    and the stage pooling through device-side torch ops), dense ConvTranspose2d 4x4 s2 + BatchNorm (batch statistics) + ReLU,
    64-channel heads, the three losses -- against torch autograd through the oracle (its VoVNet code pinned to the reference's
    module by G12).  f32: losses 1e-3, gradients cos >= 0.999; f16: rounding-level agreement."""
    import sys
    from detectron2_centernet_amd.config import get_cfg
    from detectron2_centernet_amd.data.catalog import register_synthetic, synthetic_sample
    from detectron2_centernet_amd.modeling import build_model
    from detectron2_centernet_amd.structures import Boxes, Instances
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from weights import fill_state_dict

    (tmp_path / "Base-CenterNet.yaml").write_text(BASE)
    (tmp_path / "ctdet_vovnet2_19_slim_1x.yaml").write_text(VOV_YAML)
    cfg = get_cfg()
    cfg.merge_from_file(str(tmp_path / "ctdet_vovnet2_19_slim_1x.yaml"))
    cfg.MODEL.CENTERNET.HIP_PRECISION = precision
    register_synthetic("bulb_train", num_classes=80)
    model = build_model(cfg)
    sd0 = fill_state_dict({k: v.cpu() for k, v in model.state_dict().items()}, seed=19)
    model.load_state_dict({k: v.to(model.device) for k, v in sd0.items()})
    model.train()
    frozen = [n for n, p in model.named_parameters() if not p.requires_grad]
    assert any(n.startswith("backbone.stem") for n in frozen) and any(n.startswith("backbone.stage2") for n in frozen)
    assert not any(n.startswith(("backbone.stage3", "backbone.stage4", "deconv_layers", "hm", "wh", "reg")) and "norm" not in n
                   for n in frozen)
    inputs = []
    for i in range(2):
        smp = synthetic_sample(i, size=128, num_classes=80, max_boxes=6)
        inst = Instances((128, 128))
        inst.gt_boxes, inst.gt_classes = Boxes(smp["boxes"]), smp["classes"]
        inputs.append({"image": smp["image"], "instances": inst})
    losses = model(inputs)
    sum(losses.values()).backward()
    trainable = {n for n, p in model.named_parameters() if p.requires_grad}
    sd = {k: (v.float().clone().requires_grad_(True) if k in trainable else v.float().clone()) for k, v in sd0.items()}
    x_ref, _ = O.preprocess([d["image"] for d in inputs], cfg.MODEL.PIXEL_MEAN, cfg.MODEL.PIXEL_STD, 16)
    z = MR.centernet_vovnet_forward(sd, x_ref, training=True)
    targets = [O.gen_heatmap(d["instances"].gt_boxes.tensor, d["instances"].gt_classes, 32, 32, 80) for d in inputs]
    ref = MR.centernet_losses(z, targets, [1.0])
    sum(ref.values()).backward()
    ltol = 1e-3 if precision == "f32" else 1e-2
    for k in ("hm_loss", "wh_loss", "off_loss"):
        got, want = losses[k].item(), ref[k].item()
        print(precision, k, got, want)
        assert abs(got - want) <= ltol * max(1.0, abs(want)), (k, got, want)
    worst, name_of, seen = 1.0, "", 0
    for name, p in model.named_parameters():
        if not p.requires_grad:
            assert p.grad is None, name
            continue
        gref = sd[name].grad
        if name.startswith("backbone.stage5"):       # not on the path to the loss (stage4 feeds the deconv layers)
            assert p.grad is None or p.grad.abs().max() == 0
            continue
        assert p.grad is not None and gref is not None, name
        if gref.abs().max() == 0:
            continue
        seen += 1
        cos = torch.nn.functional.cosine_similarity(p.grad.float().cpu().flatten(), gref.flatten(), dim=0).item()
        ratio = (p.grad.float().cpu().norm() / gref.norm()).item()
        if cos < worst:
            worst, name_of = cos, name
        assert (0.99 if precision == "f32" else 0.9) < ratio < (1.01 if precision == "f32" else 1.1), (name, ratio)
    print(precision, "worst gradient cosine", worst, name_of, "over", seen, "tensors")
    assert seen >= 20 and worst >= (0.999 if precision == "f32" else 0.98), (worst, name_of)
    # the same model through the trainer (flat-buffer SGD, HIP-graph capture of the step): parameters of the torch-op nodes
    # (eSE fc) and of the HIP nodes move, the frozen stem does not, losses stay finite over captured replays
    from detectron2_centernet_amd.engine.bench_train import synthetic_batch
    from detectron2_centernet_amd.engine.train_loop import SimpleTrainer
    model.zero_grad(set_to_none=True)
    cfg.SOLVER.IMS_PER_BATCH = 2
    tr = SimpleTrainer(model, None, cfg)
    fc0 = model.backbone.stage3.OSA3_1.ese.fc.weight.detach().clone()
    w0 = getattr(model.backbone.stage4.OSA4_1.layers[0], "OSA4_1_0/conv").weight.detach().clone()
    stem0 = getattr(model.backbone.stem, "stem_1/conv").weight.detach().clone()
    batch = synthetic_batch(2, 128, 0, dev)
    for _ in range(4):
        vals = {k: float(v) for k, v in tr.run_step_tensors(*batch).items()}
        assert all(v == v and abs(v) < 1e9 for v in vals.values()), vals
    assert (model.backbone.stage3.OSA3_1.ese.fc.weight.detach() - fc0).abs().max() > 0
    assert (getattr(model.backbone.stage4.OSA4_1.layers[0], "OSA4_1_0/conv").weight.detach() - w0).abs().max() > 0
    assert torch.equal(getattr(model.backbone.stem, "stem_1/conv").weight.detach(), stem0)
    print(precision, "trainer graph_state", tr.graph_state, vals)
    # round 3 ran this backbone's step eagerly: capturing it crashed hipStreamEndCapture.  Cause (round 4): the eager pass above
    # is still alive (`losses`), and with it AccumulateGrad nodes created on the default stream; the eSE parameters' gradients
    # went through them and forked the capture onto that stream.  Every parameter gradient of the step now goes straight into
    # the optimizer's flat buffer, so no AccumulateGrad node runs inside the capture
    assert tr.graph_state == "captured", tr._graphs
    for g in (g for g in tr._graphs.values() if g["graph"] is not None):   # kernels only (engine/graph_nodes.py)
        assert g["nodes"].get("kernel", 0) > 0 and set(g["nodes"]) <= {"kernel", "empty"}, g["nodes"]
