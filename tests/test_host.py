"""CPU tests of the host-side boundary: config/yaml ingestion, registries, structures, catalog, and that the
C-ABI library loads and exports every symbol include/ctdet_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BASE = """
MODEL:
  META_ARCHITECTURE: "CenterNet"
  BACKBONE:
    NAME: "build_dla34_backbone"
  PIXEL_MEAN: [0.408, 0.447, 0.470]
  PIXEL_STD: [0.289, 0.274, 0.278]
VERSION: 2
"""
DLA = """
_BASE_: "./Base-CenterNet.yaml"
MODEL:
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
  STEPS: (159000, 212000)
  MAX_ITER: 265000
  CHECKPOINT_PERIOD: 10600
#TEST:
#  EVAL_PERIOD: 1
OUTPUT_DIR: "./output/centernet-bulb-aug"
VERSION: 2
"""


def _cfg(tmp_path):
    from detectron2_centernet_amd.config import get_cfg

    (tmp_path / "Base-CenterNet.yaml").write_text(BASE)
    (tmp_path / "ctdet_dla_34_1x.yaml").write_text(DLA)
    cfg = get_cfg()
    cfg.merge_from_file(str(tmp_path / "ctdet_dla_34_1x.yaml"))
    return cfg


def test_yaml_with_base_and_tuple_strings(tmp_path):
    cfg = _cfg(tmp_path)
    assert cfg.MODEL.META_ARCHITECTURE == "CenterNet" and cfg.MODEL.BACKBONE.NAME == "build_dla34_backbone"
    assert cfg.DATASETS.TRAIN == ("bulb_train",) and cfg.INPUT.MIN_SIZE_TRAIN == (640, 672, 704, 736, 768, 800)
    assert cfg.SOLVER.STEPS == (159000, 212000) and cfg.SOLVER.BASE_LR == 2.5e-4
    assert cfg.MODEL.CENTERNET.FOCAL_LOSS_ALPHA == [1] and cfg.MODEL.PIXEL_STD == [0.289, 0.274, 0.278]
    assert cfg.MODEL.CENTERNET.TASK.HM == 80 and cfg.TEST.DETECTIONS_PER_IMAGE == 100
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", "128", "MODEL.DEVICE", "cpu"])
    assert cfg.SOLVER.IMS_PER_BATCH == 128 and cfg.MODEL.DEVICE == "cpu"
    with pytest.raises(KeyError):
        cfg.merge_from_list(["MODEL.NOPE", "1"])
    cfg.freeze()
    with pytest.raises(AttributeError):
        cfg.MODEL.DEVICE = "cuda"
    c2 = cfg.clone()
    c2.defrost()
    c2.MODEL.DEVICE = "cuda"
    assert cfg.MODEL.DEVICE == "cpu"


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree only exists in the build container")
def test_reference_yamls_load_unchanged():
    from detectron2_centernet_amd.config import get_cfg

    d = "/root/reference/projects/CenterNet/configs/COCO-Detection"
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(d, "ctdet_dla_34_1x.yaml"))
    assert cfg.MODEL.META_ARCHITECTURE == "CenterNet" and cfg.SOLVER.MAX_ITER == 265000
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(d, "ctdet_res_50_1x.yaml"))
    assert cfg.MODEL.BACKBONE.NAME == "build_resnet_backbone" and cfg.TEST.EVAL_PERIOD == 18008


def test_registries_and_model_construction(tmp_path):
    from detectron2_centernet_amd.data.catalog import DatasetCatalog, MetadataCatalog, register_synthetic
    from detectron2_centernet_amd.modeling import BACKBONE_REGISTRY, META_ARCH_REGISTRY, build_model
    from detectron2_centernet_amd.utils.registry import Registry

    assert META_ARCH_REGISTRY.get("CenterNet").__name__ == "CenterNet"
    assert callable(BACKBONE_REGISTRY.get("build_dla34_backbone"))
    with pytest.raises(KeyError):
        META_ARCH_REGISTRY.get("GeneralizedRCNN")
    r = Registry("T")

    @r.register()
    def foo():
        return 1

    assert r.get("foo") is foo
    with pytest.raises(AssertionError):
        r.register(foo)

    cfg = _cfg(tmp_path)
    cfg.MODEL.DEVICE = "cpu"
    with pytest.raises(KeyError):
        DatasetCatalog.get("definitely_not_registered")
    register_synthetic("bulb_train", num_classes=80)
    assert len(MetadataCatalog.get("bulb_train").thing_classes) == 80
    model = build_model(cfg)
    assert model.num_classes == 80 and model.backbone.down_ratio == 4 and model.size_divisibility == 32
    assert model.backbone.first_level == 2 and model.backbone.channels[2] == 64
    assert sum(p.numel() for p in model.parameters()) == 19675892          # SURVEY.md 2.5 / BASELINE.md
    assert float(model.hm[2].bias.detach()[0]) == pytest.approx(-2.19) and float(model.wh[2].bias.detach().abs().sum()) == 0.0
    sd = model.state_dict()
    for k in ("backbone.base.level2.tree1.conv1.weight", "backbone.dla_up.ida_0.proj_1.conv.conv_offset_mask.weight",
              "backbone.ida_up.node_2.actf.0.running_var", "hm.0.weight", "reg.2.bias", "pixel_mean"):
        assert k in sd
    # DCN parameterisation (third-party DCNv2 contract): zero-initialised offset/mask conv, zero bias
    dcn = model.backbone.ida_up.proj_1.conv
    assert dcn.conv_offset_mask.weight.abs().sum() == 0 and dcn.bias.abs().sum() == 0
    assert dcn.conv_offset_mask.out_channels == 27


def test_structures():
    from detectron2_centernet_amd.modeling.postprocessing import detector_postprocess
    from detectron2_centernet_amd.structures import Boxes, ImageList, Instances

    b = Boxes(torch.tensor([[0.0, 0.0, 10.0, 10.0], [5.0, 5.0, 5.0, 9.0], [-3.0, 2.0, 8.0, 30.0]]))
    assert b.nonempty().tolist() == [True, False, True] and b.area().tolist() == [100.0, 0.0, 308.0]
    inst = Instances((20, 20), pred_boxes=b, scores=torch.tensor([0.9, 0.8, 0.7]), pred_classes=torch.tensor([1, 2, 3]))
    out = detector_postprocess(inst, 40, 10)
    assert out.image_size == (40, 10) and len(out) == 2
    assert out.pred_boxes.tensor.tolist() == [[0.0, 0.0, 5.0, 20.0], [0.0, 4.0, 4.0, 40.0]]
    with pytest.raises(AssertionError):
        inst.set("bad", torch.zeros(5))
    il = ImageList.from_tensors([torch.ones(3, 50, 70), torch.ones(3, 64, 33)], 32)
    assert il.tensor.shape == (2, 3, 64, 96) and il.image_sizes == [(50, 70), (64, 33)]
    assert il.tensor[0, :, 50:].abs().sum() == 0 and il[1].shape == (3, 64, 33)
    assert ImageList.padded_size([(50, 70)], 32, 100, 100) == (128, 128)
    cat = Instances.cat([inst[[0]], inst[[2]]])
    assert len(cat) == 2 and cat.pred_classes.tolist() == [1, 3]


def test_library_exports_every_declared_symbol():
    from detectron2_centernet_amd import _lib

    header = open(os.path.join(ROOT, "include", "ctdet_hip.h")).read()
    declared = set(re.findall(r"\b(ctdet_[A-Za-z0-9_]+)\s*\(", header))
    declared -= {"ctdet_conv_desc"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    l = _lib.lib()
    assert l.ctdet_abi_version() == 7
    assert [l.ctdet_conv_cout_tile(c) for c in (4, 16, 28, 64, 80, 128, 768)] == [16, 16, 32, 64, 128, 128, 128]
    assert l.ctdet_decode_workspace_bytes(2, 128, 128, 80, 100) == 2 * l.ctdet_decode_workspace_bytes(1, 128, 128, 80, 100)
    # argument validation happens before any device work: a null descriptor is rejected with a message
    assert l.ctdet_conv2d_fwd(None, None, None, None, None, None, None, None) != 0
    assert b"null" in l.ctdet_last_error()


def test_ops_refuse_cpu_tensors():
    import detectron2_centernet_amd.ops as ops

    with pytest.raises(NotImplementedError):
        ops.maxpool2x2(torch.zeros(1, 4, 4, 8))
    with pytest.raises(NotImplementedError):
        ops.decode(torch.zeros(1, 4, 4, 4), torch.zeros(1, 4, 4, 2), None, 10, 4.0)


def test_synthetic_samples_are_deterministic():
    from detectron2_centernet_amd.data.catalog import synthetic_sample

    a, b = synthetic_sample(3), synthetic_sample(3)
    assert torch.equal(a["image"], b["image"]) and torch.equal(a["boxes"], b["boxes"])
    assert a["image"].dtype == torch.uint8 and a["image"].shape == (3, 512, 512)
    bx = a["boxes"]
    assert (bx[:, 0] >= 0).all() and (bx[:, 2] <= 512).all() and ((bx[:, 2] - bx[:, 0]) >= 8).all()


def test_checkpoint_roundtrip_reference_format(tmp_path):
    """`.pth` files in the reference's layout ({"model": state_dict, "iteration": n}, `module.`
    prefixes from DDP, `last_checkpoint` file) load key-for-key; shape mismatches are skipped and reported"""
    import torch

    from detectron2_centernet_amd.checkpoint import DetectionCheckpointer

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = torch.nn.Conv2d(3, 4, 3)
            self.bn = torch.nn.BatchNorm2d(4)
            self.register_buffer("pixel_mean", torch.zeros(3, 1, 1))

    torch.manual_seed(0)
    src, dst = Net(), Net()
    ck = DetectionCheckpointer(src, str(tmp_path))
    path = ck.save("model_0000009", iteration=9)
    assert open(tmp_path / "last_checkpoint").read() == "model_0000009.pth"
    raw = torch.load(path, weights_only=False)
    assert set(raw) == {"model", "iteration"} and set(raw["model"]) == set(src.state_dict())
    extra = DetectionCheckpointer(dst, str(tmp_path)).resume_or_load("", resume=True)
    assert extra == {"iteration": 9}
    for k, v in src.state_dict().items():
        assert torch.equal(v, dst.state_dict()[k])
    # a DDP-saved file with a wrong-shaped tensor, one missing and one unexpected key
    sd = {"module." + k: v.clone() for k, v in src.state_dict().items()}
    sd["module.conv.weight"] = torch.zeros(4, 3, 5, 5)
    del sd["module.bn.bias"], sd["module.pixel_mean"]
    sd["module.extra"] = torch.zeros(1)
    torch.save({"model": sd}, tmp_path / "ddp.pth")
    c2 = DetectionCheckpointer(Net())
    c2.load(str(tmp_path / "ddp.pth"))
    assert c2.incompatible.missing_keys == ["conv.weight", "bn.bias"] or set(c2.incompatible.missing_keys) == {"conv.weight", "bn.bias"}
    assert c2.incompatible.unexpected_keys == ["extra"]
    assert c2.incompatible.incorrect_shapes[0][0] == "conv.weight"


def test_coco_json_handoff():
    """XYWH boxes, python scalars, contiguous -> dataset category ids"""
    import json

    import torch

    from detectron2_centernet_amd.evaluation import instances_to_coco_json, results_to_coco_json
    from detectron2_centernet_amd.structures import Boxes, Instances

    inst = Instances((480, 640))
    inst.pred_boxes = Boxes(torch.tensor([[10.0, 20.0, 110.0, 70.0], [0.5, 1.5, 2.5, 4.0]]))
    inst.scores = torch.tensor([0.9, 0.25])
    inst.pred_classes = torch.tensor([3, 0])
    js = instances_to_coco_json(inst, 42)
    assert js == [{"image_id": 42, "category_id": 3, "bbox": [10.0, 20.0, 100.0, 50.0], "score": js[0]["score"]},
                  {"image_id": 42, "category_id": 0, "bbox": [0.5, 1.5, 2.0, 2.5], "score": 0.25}]
    assert abs(js[0]["score"] - 0.9) < 1e-6
    json.dumps(js)  # plain python types only
    empty = Instances((4, 4))
    empty.pred_boxes, empty.scores, empty.pred_classes = Boxes(torch.zeros(0, 4)), torch.zeros(0), torch.zeros(0, dtype=torch.long)
    flat = results_to_coco_json([{"instances": inst}, {"instances": empty}], [42, 43], {1: 0, 7: 1, 9: 2, 17: 3})
    assert [r["category_id"] for r in flat] == [17, 1] and all(r["image_id"] == 42 for r in flat)


def test_flat_sgd_accepts_torch_sgd_state_and_refuses_unimplemented_options():
    """a reference checkpoint's "optimizer" entry is a torch.optim.SGD state dict with one group per parameter in module
    order (solver/build.py:100-137): its momentum buffers land in the flat momentum buffer; NESTEROV / CLIP_GRADIENTS
    are refused instead of silently ignored"""
    import torch
    from detectron2_centernet_amd.config import get_cfg
    from detectron2_centernet_amd.solver.build import FlatSGD, build_optimizer, param_groups

    cfg = get_cfg()
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Conv2d(4, 2, 1))
    groups = param_groups(cfg, net)
    ref = torch.optim.SGD([{"params": [p], "lr": 0.1 * lf, "weight_decay": wd} for p, lf, wd in groups], lr=0.1, momentum=0.9)
    for p, _, _ in groups:
        p.grad = torch.randn_like(p)
    ref.step()
    want = [ref.state[p]["momentum_buffer"].clone() for p, _, _ in groups]
    opt = FlatSGD(groups, 0.1, 0.9)
    opt.load_state_dict(ref.state_dict())
    assert opt._first is False
    for p, buf in zip([g[0] for g in groups], want):
        i = [id(q) for q in opt.params].index(id(p))
        off, n = opt.offsets[i]
        assert torch.equal(opt.flat_mom[off:off + n], buf.reshape(-1))
    own = opt.state_dict()
    opt2 = FlatSGD(param_groups(cfg, net), 0.1, 0.9)
    opt2.load_state_dict(own)
    assert torch.equal(opt2.flat_mom, opt.flat_mom)
    with pytest.raises(KeyError):
        opt.load_state_dict({"foo": 1})
    cfg.SOLVER.NESTEROV = True
    with pytest.raises(NotImplementedError):
        build_optimizer(cfg, net)
    cfg.SOLVER.NESTEROV = False
    cfg.SOLVER.CLIP_GRADIENTS.ENABLED = True
    with pytest.raises(NotImplementedError):
        build_optimizer(cfg, net)


def test_coco_json_matches_reference_g13():
    """`instances_to_coco_json` against the records the REFERENCE's own function
    (detectron2/evaluation/coco_evaluation.py:321-382) produced for the same Instances (G13: three images, one empty):
    identical image ids, category ids, XYWH boxes and scores, as json numbers"""
    import os
    import numpy as np
    from detectron2_centernet_amd.evaluation import instances_to_coco_json
    from detectron2_centernet_amd.structures import Boxes, Instances

    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g13_coco_json.npz"))
    for i in range(3):
        inst = Instances((480, 640))
        inst.pred_boxes = Boxes(torch.from_numpy(d[f"boxes{i}"]))
        inst.scores = torch.from_numpy(d[f"scores{i}"])
        inst.pred_classes = torch.from_numpy(d[f"classes{i}"])
        recs = instances_to_coco_json(inst, 4100 + i)
        assert len(recs) == len(d[f"out_score{i}"])
        for k, r in enumerate(recs):
            assert set(r) == {"image_id", "category_id", "bbox", "score"}
            assert r["image_id"] == int(d[f"out_image_id{i}"][k]) and r["category_id"] == int(d[f"out_category_id{i}"][k])
            assert isinstance(r["category_id"], int) and isinstance(r["score"], float)
            assert r["bbox"] == d[f"out_bbox{i}"][k].tolist() and r["score"] == float(d[f"out_score{i}"][k])


def _gather_worker(outdir):
    import json
    from detectron2_centernet_amd.evaluation import COCOResultsWriter
    from detectron2_centernet_amd.structures import Boxes, Instances
    from detectron2_centernet_amd.utils import comm

    r = comm.get_rank()
    ev = COCOResultsWriter(outdir, {100 + c: c for c in range(4)})
    ev.reset()
    inst = Instances((10, 10))
    inst.pred_boxes = Boxes(torch.tensor([[1.0, 2.0, 4.0, 6.0]] * (r + 1)))
    inst.scores = torch.full((r + 1,), 0.5 + 0.1 * r)
    inst.pred_classes = torch.full((r + 1,), r, dtype=torch.int32)
    ev.process([{"image_id": 7 + r}], [{"instances": inst}])
    res = ev.evaluate()
    with open(os.path.join(outdir, f"res{r}.json"), "w") as f:
        json.dump(res, f)


def test_coco_results_writer_gathers_on_main_rank(tmp_path):
    """sharded inference under `launch`: every rank's records end up in ONE file written by the main process
    (coco_evaluation.py:130-135: gather to rank 0, {} elsewhere); two ranks over gloo"""
    import json
    from detectron2_centernet_amd.engine import launch

    launch(_gather_worker, 2, num_machines=1, machine_rank=0, dist_url="auto", args=(str(tmp_path),), backend="gloo")
    recs = json.load(open(tmp_path / "coco_instances_results.json"))
    assert sorted((r["image_id"], r["category_id"]) for r in recs) == [(7, 100), (8, 101), (8, 101)]
    assert json.load(open(tmp_path / "res0.json")) == {"bbox": {"num_detections": 3}}
    assert json.load(open(tmp_path / "res1.json")) == {}


@pytest.mark.parametrize("cin", [16, 32, 48, 64])
def test_tap_pair_weight_images_reproduce_the_convolution(cin):
    """host logic of the f16x3 3x3 kernels (include/ctdet_hip.h, ctdet_conv_desc.korder 2 / 3): `PackedConv._pack_pairs` lays a
    cout row out as 128-byte steps {X, Y}; a step multiplies two (tap, 16-channel chunk) operands.  Emulated here exactly as the
    kernel consumes it -- per step and channel group q: X.H + Y.H + X.L with H = {hi(a)[4q..], hi(b)[4q..]}, L = the lo halves --
    the sum over the steps must be the 3x3 convolution to the accuracy of the split (1e-6), for an odd number of chunks
    (korder 2: tap pairs within a chunk, a zero tenth tap) and an even one (korder 3: tap 8 pairs across two chunks)"""
    import torch.nn.functional as F
    from types import SimpleNamespace
    from detectron2_centernet_amd import ops
    g = torch.Generator().manual_seed(cin)
    cout, H, W = 8, 5, 6
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    x = torch.randn(1, cin, H, W, generator=g)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)[0]                         # [cout, H, W]
    wp = w.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous()                # tap-major rows, as PackedConv holds them
    img, korder = ops.PackedConv._pack_pairs(SimpleNamespace(Cout_pad=cout, Cin=cin), wp)
    nch = cin // 16
    assert korder == (3 if nch % 2 == 0 else 2) and img.shape == (32, nch // 2 * 288 if korder == 3 else nch * 160)
    h = img.view(torch.float16).double()                                          # [rows, steps * 64 halves]
    if korder == 3:
        steps = [(pair, [((2 * s, 0), (2 * s + 1, 0)) for s in range(4)] + [((8, 0), (8, 1))]
                  + [((2 * s, 1), (2 * s + 1, 1)) for s in range(4)]) for pair in range(nch // 2)]
        ops_of = [(t0, 2 * pair + c0, t1, 2 * pair + c1) for pair, ss in steps for (t0, c0), (t1, c1) in ss]
    else:
        ops_of = [(2 * s, c, 2 * s + 1 if s < 4 else None, c) for c in range(nch) for s in range(5)]
    assert h.shape[1] == len(ops_of) * 64
    xp = F.pad(x[0], (1, 1, 1, 1))                                                # [cin, H+2, W+2]
    x_hi = xp.to(torch.float16)
    x_lo = (xp - x_hi.float()).to(torch.float16)
    out = torch.zeros(cout, H, W, dtype=torch.float64)
    for si, (t0, c0, t1, c1) in enumerate(ops_of):
        blk = h[:cout, si * 64:(si + 1) * 64].reshape(cout, 2, 4, 2, 4)           # [row, X/Y, q, operand, j]
        for o, (t, c) in enumerate(((t0, c0), (t1, c1))):
            if t is None:                                                         # the zero tenth tap
                assert blk[:, :, :, o].abs().max() == 0
                continue
            r, s = divmod(t, 3)
            hi = x_hi[16 * c:16 * c + 16, r:r + H, s:s + W].double().reshape(4, 4, H, W)   # [q, j, y, x]
            lo = x_lo[16 * c:16 * c + 16, r:r + H, s:s + W].double().reshape(4, 4, H, W)
            X, Y = blk[:, 0, :, o], blk[:, 1, :, o]                               # [row, q, j]: w_hi, w_lo
            out += torch.einsum("rqj,qjyx->ryx", X, hi) + torch.einsum("rqj,qjyx->ryx", Y, hi) + torch.einsum("rqj,qjyx->ryx", X, lo)
    assert (out - ref).abs().max().item() < 1e-6 * max(1.0, ref.abs().max().item())
