#!/usr/bin/env python
"""Generates the golden vectors under tests/golden/ by running the REFERENCE's own Python functions
(/root/reference, read-only) in the build container.  Only data leaves this script: inputs, seeds and the
reference's outputs as .npz -- no reference source.

`import detectron2` does not work here (fvcore / yacs / pycocotools / torchvision / detectron2._C are absent --
ordinary ModuleNotFoundErrors), so the reference files are loaded one by one:
  * `detectron2` and its sub-packages are created as empty package objects that point at the real directories but
    whose `__init__.py` is never executed;
  * absent third-party modules are replaced by stubs whose attributes are dummy classes (none of them is touched
    by the functions exercised here);
  * `Tensor.cuda()` is made a no-op for `_neg_loss` (centernet.py:342-349 calls it unconditionally);
  * the missing third-party `DCN` class (deform_conv.py:13,505) is injected from this repo's oracle, so the DLA-34
    goldens pin everything *except* the DCN arithmetic.

Usage (build container only):  python tests/golden/make_golden.py
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types
import zlib

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

STUB_PREFIXES = ("fvcore", "yacs", "pycocotools", "termcolor", "torchvision", "iopath", "cv2", "tabulate",
                 "detectron2._C", "detectron2.data.transforms", "detectron2.utils.env", "detectron2.utils.comm",
                 "detectron2.utils.file_io", "detectron2.data.datasets", "detectron2.utils.logger",
                 "detectron2.evaluation.fast_eval_api", "detectron2.evaluation.evaluator", "onnx",
                 "detectron2.modeling.meta_arch.retinanet", "detectron2.export.patcher")
PACKAGES = ["detectron2", "detectron2.layers", "detectron2.structures", "detectron2.modeling",
            "detectron2.modeling.backbone", "detectron2.modeling.meta_arch", "detectron2.data", "detectron2.utils",
            "detectron2.config", "detectron2.evaluation", "detectron2.export"]


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, n):
        return _Dummy()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name == "Registry":
            return _Registry
        if name == "TORCH_VERSION":
            return tuple(int(v) for v in torch.__version__.split(".")[:2])
        return type(name, (_Dummy,), {})


class _Registry:
    """10-line stand-in for fvcore.common.registry.Registry (decorator + get)."""

    def __init__(self, name):
        self._name, self._map = name, {}

    def register(self, obj=None):
        if obj is None:
            def deco(f):
                self._map[f.__name__] = f
                return f
            return deco
        self._map[obj.__name__] = obj

    def get(self, name):
        return self._map[name]


class _LenientPackage(types.ModuleType):
    """package object whose __init__.py is never run; unknown names resolve to dummy classes"""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (_Dummy,), {})


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if any(fullname == p or fullname.startswith(p + ".") for p in STUB_PREFIXES):
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def install():
    sys.meta_path.insert(0, _StubFinder())
    for name in PACKAGES:
        m = _LenientPackage(name)
        m.__path__ = [os.path.join(REF, *name.split("."))]
        m.__package__ = name
        sys.modules[name] = m
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(sys.modules[parent], child, m)
    torch.Tensor.cuda = lambda self, *a, **k: self  # container-local shim for centernet.py:342-349


def load(name):
    return importlib.import_module(name)


from weights import fill_state_dict  # noqa: E402  (tests/golden/weights.py, shared with the tests)


def main():
    install()
    # ---- leaf files first so that packages expose the real classes ----
    shape_spec = load("detectron2.layers.shape_spec")
    sys.modules["detectron2.layers"].ShapeSpec = shape_spec.ShapeSpec
    wrappers = load("detectron2.layers.wrappers")
    for n in ("Conv2d", "ConvTranspose2d", "BatchNorm2d", "cat", "interpolate", "Linear", "nonzero_tuple"):
        setattr(sys.modules["detectron2.layers"], n, getattr(wrappers, n))
    blocks = load("detectron2.layers.blocks")
    sys.modules["detectron2.layers"].CNNBlockBase = blocks.CNNBlockBase
    boxes = load("detectron2.structures.boxes")
    instances = load("detectron2.structures.instances")
    image_list = load("detectron2.structures.image_list")
    st = sys.modules["detectron2.structures"]
    st.Boxes, st.BoxMode, st.Instances, st.ImageList = boxes.Boxes, boxes.BoxMode, instances.Instances, image_list.ImageList
    deform = load("detectron2.layers.deform_conv")

    from oracle import ctdet_oracle as O

    class OracleDCN(torch.nn.Module):
        """injected stand-in for the un-vendored DCNv2 `DCN` class (this repo's CPU restatement)."""

        def __init__(self, chi, cho, kernel_size, stride, padding, dilation, deformable_groups):
            super().__init__()
            self.weight = torch.nn.Parameter(torch.zeros(cho, chi, *kernel_size))
            self.bias = torch.nn.Parameter(torch.zeros(cho))
            self.conv_offset_mask = torch.nn.Conv2d(chi, 27, 3, 1, 1)

        def forward(self, x):
            return O.dcn_module_forward(x, self.conv_offset_mask.weight, self.conv_offset_mask.bias, self.weight,
                                        self.bias)

    deform.DCN = OracleDCN
    sys.modules["detectron2.layers"].DeformConvV2 = deform.DeformConvV2
    backbone_mod = load("detectron2.modeling.backbone.backbone")
    build_mod = load("detectron2.modeling.backbone.build")
    bb = sys.modules["detectron2.modeling.backbone"]
    bb.Backbone, bb.BACKBONE_REGISTRY, bb.build_backbone = backbone_mod.Backbone, build_mod.BACKBONE_REGISTRY, build_mod.build_backbone
    dla = load("detectron2.modeling.backbone.dla")
    bb.DLAUp, bb.IDAUp = dla.DLAUp, dla.IDAUp
    catalog = load("detectron2.data.catalog")
    du = load("detectron2.data.detection_utils")
    post = load("detectron2.modeling.postprocessing")
    sys.modules["detectron2.modeling"].postprocessing = post
    sys.modules["detectron2.modeling.meta_arch"].build = load("detectron2.modeling.meta_arch.build")
    cn = load("detectron2.modeling.meta_arch.centernet")
    Boxes, Instances, ImageList = boxes.Boxes, instances.Instances, image_list.ImageList

    out = {}
    # ---------------- G1: gaussian_radius grid ----------------
    hs, ws = np.meshgrid(np.arange(1, 129), np.arange(1, 129), indexing="ij")
    rad = np.array([du.gaussian_radius((int(h), int(w))) for h, w in zip(hs.ravel(), ws.ravel())], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "g1_gaussian_radius.npz"), h=hs.ravel().astype(np.int32),
                        w=ws.ravel().astype(np.int32), radius=rad)

    # ---------------- G2: gen_heatmap ----------------
    meta = types.SimpleNamespace(thing_classes=[str(i) for i in range(80)])
    g = torch.Generator().manual_seed(20260101)
    cases = []
    # random scenes
    for n in (1, 5, 32):
        wh = torch.rand(n, 2, generator=g) * 248 + 8
        ctr = torch.rand(n, 2, generator=g) * (512 - wh) + wh / 2
        cases.append((torch.cat([ctr - wh / 2, ctr + wh / 2], 1), torch.randint(0, 80, (n,), generator=g)))
    # hand-built edge cases: zero-area, border-touching, last row/col centre, same-class overlap
    cases.append((torch.tensor([[10.0, 10.0, 10.0, 50.0], [0.0, 0.0, 37.0, 23.0], [400.0, 380.0, 511.9, 511.9],
                                [400.0, 380.0, 511.9, 511.9], [100.0, 100.0, 180.0, 160.0], [120.0, 110.0, 200.0, 170.0],
                                [508.0, 508.0, 512.0, 512.0], [3.0, 300.0, 5.0, 302.0]]),
                  torch.tensor([3, 79, 7, 7, 11, 11, 0, 42])))
    # more than 128 objects
    ctr = torch.rand(150, 2, generator=g) * 400 + 50
    cases.append((torch.cat([ctr - 12, ctr + 12], 1), torch.randint(0, 80, (150,), generator=g)))
    g2 = {}
    for i, (bx, cl) in enumerate(cases):
        inst = Instances((512, 512))
        inst.gt_boxes = Boxes(bx)
        inst.gt_classes = cl
        r = du.gen_heatmap(inst, np.array([128, 128]), meta)
        hm = r["hm"].numpy()
        nz = np.nonzero(hm)
        g2[f"boxes{i}"], g2[f"classes{i}"] = bx.numpy(), cl.numpy()
        g2[f"hm_idx{i}"] = np.ravel_multi_index(nz, hm.shape).astype(np.int32)
        g2[f"hm_val{i}"] = hm[nz]
        for k in ("wh", "reg", "ind", "reg_mask"):
            g2[f"{k}{i}"] = r[k].numpy()
    g2["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "g2_gen_heatmap.npz"), **g2)

    # ---------------- G3: _neg_loss fwd + grads ----------------
    g3 = {}
    case = 0
    for alpha in ([1], [0.25]):
        for with_pos in (True, False):
            gg = torch.Generator().manual_seed(300 + case)
            logits = (torch.randn(2, 8, 32, 32, generator=gg) * 3 - 2)
            logits[0, 0, 0, 0], logits[0, 0, 0, 1] = 12.0, -12.0
            logits.requires_grad_(True)
            gt = torch.rand(2, 8, 32, 32, generator=gg) ** 4
            if with_pos:
                gt.view(-1)[torch.randint(0, gt.numel(), (30,), generator=gg)] = 1.0
                gt[0, 0, 0, 0] = 1.0
            pred = torch.clamp(logits.sigmoid(), min=1e-4, max=1 - 1e-4)
            loss = cn._neg_loss(pred, gt, list(alpha))
            loss.backward()
            g3[f"logits{case}"], g3[f"gt{case}"] = logits.detach().numpy(), gt.numpy()
            g3[f"alpha{case}"], g3[f"loss{case}"] = np.array(alpha, dtype=np.float32), loss.detach().numpy()
            g3[f"grad{case}"] = logits.grad.numpy()
            case += 1
    g3["n_cases"] = np.array(case)
    np.savez_compressed(os.path.join(HERE, "g3_neg_loss.npz"), **g3)

    # ---------------- G4: RegL1Loss ----------------
    gg = torch.Generator().manual_seed(400)
    outp = torch.randn(3, 2, 16, 16, generator=gg, requires_grad=True)
    mask = (torch.rand(3, 128, generator=gg) < 0.2).to(torch.uint8)
    ind = torch.randint(0, 256, (3, 128), generator=gg)
    ind[0, 1] = ind[0, 0]
    mask[0, 0] = mask[0, 1] = 1
    tgt = torch.randn(3, 128, 2, generator=gg)
    l1 = cn.RegL1Loss()(outp, mask, ind, tgt)
    l1.backward()
    np.savez_compressed(os.path.join(HERE, "g4_reg_l1.npz"), output=outp.detach().numpy(), mask=mask.numpy(),
                        ind=ind.numpy(), target=tgt.numpy(), loss=l1.detach().numpy(), grad=outp.grad.numpy())

    # ---------------- G5: ctdet_decode + inference_single_image + detector_postprocess ----------------
    g5 = {}
    for i, (C, H, W) in enumerate([(80, 128, 128), (8, 32, 48), (4, 16, 16)]):
        gg = torch.Generator().manual_seed(500 + i)
        # tie-free heat: distinct values (a permutation mapped into (1e-4, 1-1e-4))
        n = C * H * W
        perm = torch.randperm(n, generator=gg).double()
        heat = (1e-4 + (perm + 0.5) / n * (1 - 2e-4)).float().view(1, C, H, W)
        assert heat.unique().numel() == n
        wh = torch.rand(1, 2, H, W, generator=gg) * 20
        reg = torch.rand(1, 2, H, W, generator=gg)
        b, s, c = cn.ctdet_decode(heat, wh, reg=reg, down_ratio=4, K=100)
        g5[f"heat{i}"], g5[f"wh{i}"], g5[f"reg{i}"] = heat.numpy(), wh.numpy(), reg.numpy()
        g5[f"boxes{i}"], g5[f"scores{i}"], g5[f"classes{i}"] = b.numpy(), s.numpy(), c.numpy()
        # inference_single_image + detector_postprocess on the first case with a non-trivial output size
        ns = types.SimpleNamespace(backbone=types.SimpleNamespace(down_ratio=4), topk_candidates=100,
                                   max_detections_per_image=50, score_threshold=0.9993 if i == 0 else 0.5)
        res = cn.CenterNet.inference_single_image(ns, {"hm": heat, "wh": wh, "reg": reg}, (H * 4, W * 4))
        res = post.detector_postprocess(res, H * 8, W * 6)
        g5[f"pp_boxes{i}"], g5[f"pp_scores{i}"] = res.pred_boxes.tensor.numpy(), res.scores.numpy()
        g5[f"pp_classes{i}"] = res.pred_classes.numpy()
        g5[f"pp_thresh{i}"] = np.array(ns.score_threshold, dtype=np.float32)
    # plateau case: documents the `hmax == heat` semantics (every plateau cell is a peak)
    heat = torch.full((1, 4, 16, 16), 0.01)
    heat[0, 1, 4:6, 4:6] = 0.9
    heat[0, 2, 10, 10] = 0.95
    kept = cn._nms(heat)
    g5["plateau_heat"], g5["plateau_kept"] = heat.numpy(), kept.numpy()
    g5["n_cases"] = np.array(3)
    np.savez_compressed(os.path.join(HERE, "g5_decode.npz"), **g5)

    # ---------------- G6: ImageList.from_tensors + preprocess ----------------
    gg = torch.Generator().manual_seed(600)
    imgs = [torch.randint(0, 256, (3, 50, 70), generator=gg, dtype=torch.uint8),
            torch.randint(0, 256, (3, 64, 33), generator=gg, dtype=torch.uint8)]
    mean = torch.Tensor([0.408, 0.447, 0.470]).view(-1, 1, 1)
    std = torch.Tensor([0.289, 0.274, 0.278]).view(-1, 1, 1)
    normed = [(x / 255.0 - mean) / std for x in imgs]
    il = ImageList.from_tensors(normed, 32)
    np.savez_compressed(os.path.join(HERE, "g6_preprocess.npz"), img0=imgs[0].numpy(), img1=imgs[1].numpy(),
                        batch=il.tensor.numpy(), sizes=np.array(il.image_sizes))

    # ---------------- G7/G8: DLA-34 (+ heads) with name-keyed weights ----------------
    cfg = types.SimpleNamespace(MODEL=types.SimpleNamespace(CENTERNET=types.SimpleNamespace(
        DOWN_RATIO=4, NUM_CLASSES=80, LAST_LEVEL=5, LEVELS=[1, 1, 1, 2, 2, 1], CHANNELS=[16, 32, 64, 128, 256, 512],
        SIZE_DIVISIBILITY=32)))
    torch.manual_seed(7)
    model = dla.DLA34(cfg, pretrained=False)
    model.eval()
    sd = fill_state_dict(model.state_dict(), seed=7)
    model.load_state_dict(sd)
    gg = torch.Generator().manual_seed(700)
    x = torch.randn(1, 3, 64, 96, generator=gg)
    with torch.no_grad():
        base_maps = model.base(x)
        y = model(x)
    g7 = {"x": x.numpy()}
    for i, m in enumerate(base_maps):
        g7[f"base{i}"] = m.numpy()
    for i, m in enumerate(y):
        g7[f"y{i}"] = m.numpy()
    # depthwise up-conv initialisation (fill_up_weights) for f = 2, 4
    for f in (2, 4):
        up = torch.nn.ConvTranspose2d(4, 4, f * 2, stride=f, padding=f // 2, output_padding=0, groups=4, bias=False)
        dla.fill_up_weights(up)
        g7[f"up_w{f}"] = up.weight.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "g7_dla34.npz"), **g7)
    keys = sorted(model.state_dict().keys())
    with open(os.path.join(HERE, "g8_dla34_state_dict_keys.txt"), "w") as f:
        for k in keys:
            f.write(f"{k} {tuple(model.state_dict()[k].shape)}\n")
    # ---------------- G9: ResNet-50 (res4, FrozenBN, stride in 1x1) + CenterNet deconv layers ----
    bn_mod = load("detectron2.layers.batch_norm")
    for n in ("FrozenBatchNorm2d", "get_norm", "NaiveSyncBatchNorm"):
        setattr(sys.modules["detectron2.layers"], n, getattr(bn_mod, n))
    sys.modules["detectron2.layers"].ModulatedDeformConv = deform.ModulatedDeformConv
    sys.modules["detectron2.layers"].DeformConv = deform.DeformConv
    resnet = load("detectron2.modeling.backbone.resnet")
    torch.manual_seed(9)
    stem = resnet.BasicStem(in_channels=3, out_channels=64, norm="FrozenBN")
    stages, cin, cout, bott = [], 64, 256, 64
    for idx, nblk in enumerate([3, 4, 6]):            # res2..res4 (OUT_FEATURES = ["res4"]), resnet.py:609-642
        first_stride = 1 if idx == 0 else 2
        stages.append(resnet.ResNet.make_stage(block_class=resnet.BottleneckBlock, num_blocks=nblk,
                                               stride_per_block=[first_stride] + [1] * (nblk - 1), in_channels=cin,
                                               out_channels=cout, norm="FrozenBN", bottleneck_channels=bott,
                                               stride_in_1x1=True, dilation=1, num_groups=1))
        cin, cout, bott = cout, cout * 2, bott * 2
    r50 = resnet.ResNet(stem, stages, out_features=["res4"]).freeze(2)
    deconv = cn.CenterNet._make_deconv_layer(None, 1024, 2, [256, 256], [4, 4])
    r50.eval(); deconv.eval()
    sd_r = fill_state_dict(r50.state_dict(), seed=9)
    r50.load_state_dict(sd_r)
    sd_d = fill_state_dict(deconv.state_dict(), seed=10)
    deconv.load_state_dict(sd_d)
    gg = torch.Generator().manual_seed(900)
    x = torch.randn(1, 3, 64, 96, generator=gg)
    with torch.no_grad():
        res4 = r50(x)["res4"]
        up = deconv(res4)
    np.savez_compressed(os.path.join(HERE, "g9_resnet50.npz"), x=x.numpy(), res4=res4.numpy(), up=up.numpy())
    with open(os.path.join(HERE, "g9_resnet50_state_dict_keys.txt"), "w") as f:
        for k in sorted(r50.state_dict().keys()):
            f.write(f"backbone.{k} {tuple(r50.state_dict()[k].shape)}\n")
        for k in sorted(deconv.state_dict().keys()):
            f.write(f"deconv_layers.{k} {tuple(deconv.state_dict()[k].shape)}\n")
    # ---------------- G10: ResNet-18 (BasicBlock, FrozenBN) res4 + deconv layers: the ctdet_res_18/34 configs ----------
    torch.manual_seed(10)
    stem18 = resnet.BasicStem(in_channels=3, out_channels=64, norm="FrozenBN")
    stages18, cin, cout = [], 64, 64
    for idx, nblk in enumerate([2, 2, 2]):
        first_stride = 1 if idx == 0 else 2
        stages18.append(resnet.ResNet.make_stage(block_class=resnet.BasicBlock, num_blocks=nblk,
                                                 stride_per_block=[first_stride] + [1] * (nblk - 1), in_channels=cin,
                                                 out_channels=cout, norm="FrozenBN"))
        cin, cout = cout, cout * 2
    r18 = resnet.ResNet(stem18, stages18, out_features=["res4"]).freeze(2)
    deconv18 = cn.CenterNet._make_deconv_layer(None, 256, 2, [256, 256], [4, 4])
    r18.eval(); deconv18.eval()
    r18.load_state_dict(fill_state_dict(r18.state_dict(), seed=11))
    deconv18.load_state_dict(fill_state_dict(deconv18.state_dict(), seed=12))
    gg = torch.Generator().manual_seed(1000)
    x = torch.randn(1, 3, 64, 96, generator=gg)
    with torch.no_grad():
        res4 = r18(x)["res4"]
        up = deconv18(res4)
    np.savez_compressed(os.path.join(HERE, "g10_resnet18.npz"), x=x.numpy(), res4=res4.numpy(), up=up.numpy())
    with open(os.path.join(HERE, "g10_resnet18_state_dict_keys.txt"), "w") as f:
        for k in sorted(r18.state_dict().keys()):
            f.write(f"backbone.{k} {tuple(r18.state_dict()[k].shape)}\n")
        for k in sorted(deconv18.state_dict().keys()):
            f.write(f"deconv_layers.{k} {tuple(deconv18.state_dict()[k].shape)}\n")
    # ---------------- G12: VoVNet-19-slim-eSE (the ctdet_vovnet2_19_slim config's backbone), FrozenBN, stages 2..5 --------
    sys.modules["detectron2.modeling.backbone"].Backbone = load("detectron2.modeling.backbone.backbone").Backbone
    vov = load("detectron2.modeling.backbone.vovnet")
    from types import SimpleNamespace as NS
    vcfg = NS(MODEL=NS(VOVNET=NS(NORM="FrozenBN", CONV_BODY="V-19-slim-eSE"), BACKBONE=NS(FREEZE_AT=0)))
    torch.manual_seed(12)
    v19 = vov.VoVNet(vcfg, 3, out_features=["stage2", "stage3", "stage4", "stage5"]).eval()
    v19.load_state_dict(fill_state_dict(v19.state_dict(), seed=13))
    gg = torch.Generator().manual_seed(1200)
    x = torch.randn(1, 3, 70, 100, generator=gg)       # odd sizes: the ceil-mode pooling matters
    with torch.no_grad():
        outs = v19(x)
    np.savez_compressed(os.path.join(HERE, "g12_vovnet19slim.npz"), x=x.numpy(), **{k: v.numpy() for k, v in outs.items()})
    with open(os.path.join(HERE, "g12_vovnet19slim_state_dict_keys.txt"), "w") as f:
        for k in sorted(v19.state_dict().keys()):
            f.write(f"backbone.{k} {tuple(v19.state_dict()[k].shape)}\n")
    # ---------------- G13: the COCO results wire format (coco_evaluation.py:321-382 instances_to_coco_json) ----------------
    # records of three images (one without detections) from the reference's own Instances / Boxes; stored as flat arrays
    st.BoxMode = boxes.BoxMode
    ce = load("detectron2.evaluation.coco_evaluation")
    gg = torch.Generator().manual_seed(1300)
    g13 = {}
    for i, n in enumerate([7, 0, 3]):
        xy = torch.rand(n, 2, generator=gg) * 300
        wh_ = torch.rand(n, 2, generator=gg) * 200 + 0.25
        bx = torch.cat([xy, xy + wh_], 1)
        sc = torch.rand(n, generator=gg)
        cl = torch.randint(0, 80, (n,), generator=gg, dtype=torch.int64).to(torch.int32)
        inst = instances.Instances((480, 640))
        inst.pred_boxes = boxes.Boxes(bx)
        inst.scores = sc
        inst.pred_classes = cl
        recs = ce.instances_to_coco_json(inst, 4100 + i)
        assert all(set(r) == {"image_id", "category_id", "bbox", "score"} for r in recs)
        g13[f"boxes{i}"], g13[f"scores{i}"], g13[f"classes{i}"] = bx.numpy(), sc.numpy(), cl.numpy()
        g13[f"out_image_id{i}"] = np.array([r["image_id"] for r in recs], dtype=np.int64)
        g13[f"out_category_id{i}"] = np.array([r["category_id"] for r in recs], dtype=np.int64)
        g13[f"out_bbox{i}"] = np.array([r["bbox"] for r in recs], dtype=np.float64).reshape(len(recs), 4)
        g13[f"out_score{i}"] = np.array([r["score"] for r in recs], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "g13_coco_json.npz"), **g13)
    # ---------------- G14: dataset-dict annotations -> Instances (detection_utils.py:256-287, 362-384, 456-483) ----------------
    # the reference's own transform_instance_annotations / annotations_to_instances / filter_empty_instances on XYWH and XYXY
    # annotations (boxes leaving the image, degenerate after clipping, crowd flags ignored here as in the functions) under a
    # duck-typed affine transform (x' = sx x + tx, y' = sy y + ty on the 4 corners: what fvcore's ScaleTransform / crop do);
    # the transform classes themselves are fvcore's (absent), so only `apply_box` is supplied
    class _Affine:
        def __init__(self, sx, sy, tx, ty):
            self.sx, self.sy, self.tx, self.ty = sx, sy, tx, ty

        def apply_box(self, box):
            b = np.asarray(box, dtype=np.float64).reshape(-1, 4)
            idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
            c = b[:, idxs].reshape(-1, 2)
            c = c * [self.sx, self.sy] + [self.tx, self.ty]
            c = c.reshape(-1, 4, 2)
            return np.concatenate((c.min(axis=1), c.max(axis=1)), axis=1)

    rng = np.random.RandomState(1400)
    g14 = {}
    for i, (sx, sy, tx, ty, hw) in enumerate([(1.5, 1.5, 0.0, 0.0, (720, 960)), (0.6, 0.75, -40.0, -25.0, (200, 300)),
                                               (-1.0, 1.0, 639.0, 0.0, (480, 640))]):
        n = 9
        xywh = np.concatenate([rng.uniform(-30, 600, (n, 2)), rng.uniform(0.0, 220, (n, 2))], 1)
        xywh[2, 2:] = 0.0                      # empty box
        xywh[5, :2] = [5000.0, 5000.0]         # far outside: empty after clipping
        modes = [boxes.BoxMode.XYWH_ABS if k % 2 == 0 else boxes.BoxMode.XYXY_ABS for k in range(n)]
        raw = [xywh[k] if modes[k] == boxes.BoxMode.XYWH_ABS else np.concatenate([xywh[k, :2], xywh[k, :2] + xywh[k, 2:]])
               for k in range(n)]
        cats = rng.randint(0, 80, n)
        annos = [{"bbox": raw[k].tolist(), "bbox_mode": modes[k], "category_id": int(cats[k])} for k in range(n)]
        tr = _Affine(sx, sy, tx, ty)
        out = [du.transform_instance_annotations(dict(a), tr, hw) for a in annos]
        inst = du.annotations_to_instances(out, hw)
        kept = du.filter_empty_instances(inst)
        g14[f"bbox{i}"] = np.array(raw, dtype=np.float64)
        g14[f"mode{i}"] = np.array([int(m) for m in modes], dtype=np.int64)
        g14[f"cat{i}"] = cats.astype(np.int64)
        g14[f"affine{i}"] = np.array([sx, sy, tx, ty, hw[0], hw[1]], dtype=np.float64)
        g14[f"out_bbox{i}"] = np.array([o["bbox"] for o in out], dtype=np.float64)
        g14[f"inst_boxes{i}"] = inst.gt_boxes.tensor.numpy()
        g14[f"kept_boxes{i}"] = kept.gt_boxes.tensor.numpy()
        g14[f"kept_classes{i}"] = kept.gt_classes.numpy()
    np.savez_compressed(os.path.join(HERE, "g14_annotations.npz"), **g14)
    # ---------------- G15: the serving split of the export path (export/meta_modeling.py:151-201) ----------------
    # the reference's own CenterNetModel.inference ({images} -> {hm after sigmoid + clamp, wh, reg}) on a ResNet-18 CenterNet
    # assembled from the reference's modules: G10's backbone + deconv layers and heads built as centernet.py:111-134 builds
    # them (HEAD_CONV 64), name-keyed weights.  The wrapped model subclasses the reference CenterNet (the function asserts
    # the type) without running its constructor, which needs a dataset and the network.
    sys.modules["detectron2.modeling.meta_arch"].CenterNet = cn.CenterNet
    sys.modules["detectron2.modeling"].meta_arch = sys.modules["detectron2.modeling.meta_arch"]
    mm = load("detectron2.export.meta_modeling")

    class _Wrapped(cn.CenterNet):
        def __init__(self):
            torch.nn.Module.__init__(self)
            self.backbone, self.deconv_layers, self.backbone_type = r18, deconv18, "resnet"
            self.heads = {"HM": 80, "WH": 2, "REG": 2}
            for head, classes in self.heads.items():
                self.__setattr__(head.lower(), torch.nn.Sequential(
                    torch.nn.Conv2d(256, 64, kernel_size=3, padding=1, bias=True), torch.nn.ReLU(inplace=True),
                    torch.nn.Conv2d(64, classes, kernel_size=1, stride=1, padding=0, bias=True)))

    wm = _Wrapped().eval()
    heads_sd = {k: v for k, v in wm.state_dict().items() if k.split(".")[0] in ("hm", "wh", "reg")}
    wm.load_state_dict(fill_state_dict(heads_sd, seed=15), strict=False)
    em = mm.CenterNetModel.__new__(mm.CenterNetModel)
    torch.nn.Module.__init__(em)
    em._wrapped_model = wm
    gg = torch.Generator().manual_seed(1000)
    x = torch.randn(1, 3, 64, 96, generator=gg)          # G10's input
    with torch.no_grad():
        res = em.inference({"images": x})
    assert em.get_input_names() == ["images", "im_info"]
    np.savez_compressed(os.path.join(HERE, "g15_export_split.npz"), x=x.numpy(), hm=res["hm"].numpy(), wh=res["wh"].numpy(),
                        reg=res["reg"].numpy(), output_names=np.array(em.get_output_names()))
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
