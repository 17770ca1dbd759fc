"""SURVEY 8(f) rank 1, first version: COCO-json loader, the CenterNet dataset mapper (ResizeShortestEdge + colour
augmentations + box transforms) and the training sampler / loader.  Known answers follow the reference's rules
(data/transforms/augmentation_impl.py:153-173, data/detection_utils.py:256-287, data/datasets/coco.py:62-196,
data/samplers/distributed_sampler.py:43-55)."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

from detectron2_centernet_amd.config import get_cfg
from detectron2_centernet_amd.data import (DatasetCatalog, MetadataCatalog, TrafficLightDatasetMapper, TrainingSampler,
                                           build_detection_train_loader, load_coco_json, register_coco_instances)
from detectron2_centernet_amd.data import detection_utils as utils
from detectron2_centernet_amd.data import transforms as T
from detectron2_centernet_amd.structures import BoxMode


def test_resize_shortest_edge_size_rule():
    f = T.ResizeShortestEdge.output_size
    assert f(480, 640, 800, 1333) == (800, 1067)
    assert f(640, 480, 800, 1333) == (1067, 800)
    assert f(427, 640, 800, 1333) == (800, 1199)
    assert f(500, 1500, 800, 1333) == (444, 1333)      # longer edge capped, then both rounded half up
    assert f(512, 512, 640, 1333) == (640, 640)
    np.random.seed(0)
    aug = T.ResizeShortestEdge((640, 672, 704, 736, 768, 800), 1333, "choice")
    sizes = {aug.get_transform(np.zeros((480, 640, 3), np.uint8)).new_h for _ in range(200)}
    assert sizes == {640, 672, 704, 736, 768, 800}
    aug = T.ResizeShortestEdge((100, 110), 1333, "range")
    sizes = {aug.get_transform(np.zeros((480, 640, 3), np.uint8)).new_h for _ in range(300)}
    assert sizes == set(range(100, 111))


def test_transforms_boxes_and_blend():
    t = T.ResizeTransform(100, 200, 50, 300)
    assert np.allclose(t.apply_box(np.array([[10., 20., 110., 80.]])), [[15., 10., 165., 40.]])
    img = (np.arange(100 * 200 * 3) % 251).astype(np.uint8).reshape(100, 200, 3)
    out = t.apply_image(img)
    assert out.shape == (50, 300, 3) and out.dtype == np.uint8
    assert np.array_equal(out, np.asarray(Image.fromarray(img).resize((300, 50), Image.BILINEAR)))
    fl = T.HFlipTransform(200)
    assert np.allclose(fl.apply_box(np.array([[10., 20., 110., 80.]])), [[90., 20., 190., 80.]])
    both = T.TransformList([t, T.NoOpTransform(), T.HFlipTransform(300)])
    assert len(both) == 2 and np.allclose(both.apply_box(np.array([[10., 20., 110., 80.]])), [[135., 10., 285., 40.]])
    b = T.BlendTransform(src_image=0, src_weight=-0.2, dst_weight=1.2)        # brightness 1.2
    x = np.array([[[0, 100, 250]]], dtype=np.uint8)
    assert b.apply_image(x).tolist() == [[[0, 120, 255]]]                       # float32 blend, clipped, truncated
    np.random.seed(3)
    sat = T.RandomSaturation(0.5, 0.5).get_transform(x.astype(np.uint8))
    gray = 0.587 * 100 + 0.114 * 250
    assert np.allclose(sat.src_image.reshape(-1), [gray]) and sat.src_weight == 0.5
    light = T.RandomLighting(0.8).get_transform(x)
    assert light.src_image.shape == (3,) and light.src_weight == 1.0 and light.dst_weight == 1.0


def _write_dataset(root, n=3):
    os.makedirs(root, exist_ok=True)
    rng = np.random.RandomState(7)
    images, anns = [], []
    aid = 1
    for i in range(n):
        h, w = 60 + 10 * i, 90 - 8 * i
        arr = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        arr[0, 0] = (200, 100, 50)                                   # RGB marker pixel
        Image.fromarray(arr).save(os.path.join(root, f"im{i}.png"))
        images.append({"id": 10 - i, "file_name": f"im{i}.png", "height": h, "width": w})
        for bb, cat, crowd in (([5, 6, 30, 20], 7, 0), ([w - 10, h - 10, 30, 30], 3, 0), ([1, 1, 4, 4], 3, 1),
                               ([w + 5, 2, 10, 10], 9, 0)):
            anns.append({"id": aid, "image_id": 10 - i, "bbox": bb, "category_id": cat, "iscrowd": crowd, "area": 1.0})
            aid += 1
    cats = [{"id": 9, "name": "nine"}, {"id": 3, "name": "three"}, {"id": 7, "name": "seven"}]
    jf = os.path.join(root, "ann.json")
    with open(jf, "w") as f:
        json.dump({"images": images, "annotations": anns, "categories": cats}, f)
    return jf


def test_coco_json_loader(tmp_path):
    jf = _write_dataset(str(tmp_path))
    recs = load_coco_json(jf, str(tmp_path), "data_test_coco")
    assert [r["image_id"] for r in recs] == [8, 9, 10]                # sorted by image id
    meta = MetadataCatalog.get("data_test_coco")
    assert meta.thing_classes == ["three", "seven", "nine"]
    assert meta.thing_dataset_id_to_contiguous_id == {3: 0, 7: 1, 9: 2}
    r = recs[-1]                                                      # image id 10 = im0
    assert r["file_name"].endswith("im0.png") and (r["height"], r["width"]) == (60, 90)
    assert [a["category_id"] for a in r["annotations"]] == [1, 0, 0, 2]
    assert all(a["bbox_mode"] == BoxMode.XYWH_ABS for a in r["annotations"])
    assert r["annotations"][2]["iscrowd"] == 1


def _cfg(min_sizes=(48,), sampling="choice", max_size=1333, test_size=40):
    cfg = get_cfg()
    cfg.INPUT.MIN_SIZE_TRAIN = tuple(min_sizes)
    cfg.INPUT.MIN_SIZE_TRAIN_SAMPLING = sampling
    cfg.INPUT.MAX_SIZE_TRAIN = max_size
    cfg.INPUT.MIN_SIZE_TEST = test_size
    cfg.INPUT.MAX_SIZE_TEST = max_size
    return cfg


def test_mapper_contract(tmp_path):
    jf = _write_dataset(str(tmp_path))
    recs = load_coco_json(jf, str(tmp_path), "data_test_mapper")
    rec = recs[-1]                                                    # 60 x 90
    np.random.seed(11)
    out = TrafficLightDatasetMapper(_cfg((48,)), is_train=True)(rec)
    assert "annotations" not in out and "annotations" in rec          # deep copy: the dataset dict is untouched
    img = out["image"]
    assert img.dtype == torch.uint8 and tuple(img.shape) == (3, 48, 72)
    inst = out["instances"]
    assert inst.image_size == (48, 72)
    # crowd annotation dropped; the box entirely right of the image is clipped to zero width and filtered out
    assert inst.gt_classes.tolist() == [1, 0]
    s = 48 / 60
    want = torch.tensor([[5 * s, 6 * s, 35 * s, 26 * s], [80 * s, 50 * s, 72.0, 48.0]])   # second one clipped to the image
    assert torch.allclose(inst.gt_boxes.tensor, want, atol=1e-5)
    # INPUT.FORMAT BGR: channel order flipped relative to the file (no resize: identity-size mapper)
    raw = TrafficLightDatasetMapper(_cfg(test_size=60), is_train=False)(rec)["image"]
    assert raw[:, 0, 0].tolist() == [50, 100, 200]
    ev = TrafficLightDatasetMapper(_cfg(), is_train=False)(rec)
    assert "instances" not in ev and "annotations" not in ev and tuple(ev["image"].shape) == (3, 40, 60)
    assert ev["height"] == 60 and ev["width"] == 90
    bad = dict(rec, width=91)
    with pytest.raises(utils.SizeMismatchError):
        TrafficLightDatasetMapper(_cfg(), True)(bad)


def test_mapper_colour_augmentations_are_applied_with_their_probability(tmp_path):
    jf = _write_dataset(str(tmp_path), n=1)
    rec = load_coco_json(jf, str(tmp_path))[0]
    m = TrafficLightDatasetMapper(_cfg((60,)), is_train=True)           # no resize: any change comes from the colour ops
    base = TrafficLightDatasetMapper(_cfg(test_size=60), is_train=False)(rec)["image"]
    np.random.seed(5)
    changed = sum(not torch.equal(m(rec)["image"], base) for _ in range(300))
    p_any = 1 - 0.85 ** 4                                              # four independent RandomApply(prob 0.15)
    assert abs(changed / 300 - p_any) < 0.09


def test_training_sampler_partitions_a_shared_permutation():
    import itertools
    a = list(itertools.islice(iter(TrainingSampler(10, seed=3, rank=0, world_size=2)), 15))
    b = list(itertools.islice(iter(TrainingSampler(10, seed=3, rank=1, world_size=2)), 15))
    full = list(itertools.islice(iter(TrainingSampler(10, seed=3)), 30))
    assert a == full[0::2] and b == full[1::2]
    assert sorted(full[:10]) == list(range(10)) and sorted(full[10:20]) == list(range(10))
    assert full[:10] != full[10:20]                                    # reshuffled every epoch


def test_train_loader_batches(tmp_path):
    jf = _write_dataset(str(tmp_path))
    register_coco_instances("data_test_loader", {}, jf, str(tmp_path))
    cfg = _cfg((32, 40), "choice")
    cfg.DATASETS.TRAIN = ("data_test_loader",)
    cfg.SOLVER.IMS_PER_BATCH = 4
    np.random.seed(0)
    it = build_detection_train_loader(cfg, rank=1, world_size=2, seed=1, num_workers=2)
    for _ in range(3):
        batch = next(it)
        assert len(batch) == 2
        for d in batch:
            assert d["image"].dtype == torch.uint8 and min(d["image"].shape[1:]) in (32, 40)
            assert len(d["instances"]) == 2
    DatasetCatalog.remove("data_test_loader")


@pytest.mark.gpu
def test_mapped_batches_of_unequal_sizes_train_and_infer(tmp_path):
    """the loader's ragged batches go through the HIP path unchanged: finite losses in training mode, Instances in eval"""
    import bench
    dev = torch.device("cuda:0")
    model, cfg = bench.build_model("f16", dev, seed=2)
    jf = _write_dataset(str(tmp_path))
    name = "data_test_gpu"
    if name in DatasetCatalog:
        DatasetCatalog.remove(name)
    register_coco_instances(name, {}, jf, str(tmp_path))
    cfg.DATASETS.TRAIN = (name,)
    cfg.SOLVER.IMS_PER_BATCH = 3
    cfg.INPUT.MIN_SIZE_TRAIN = (96, 128, 160)
    cfg.INPUT.MIN_SIZE_TRAIN_SAMPLING = "choice"
    cfg.INPUT.MAX_SIZE_TRAIN = 256
    np.random.seed(1)
    batch = next(build_detection_train_loader(cfg, num_workers=0))
    for d in batch:                       # class ids of this toy dataset (0..2) are valid for the 80-class model
        d["image"] = d["image"].to(dev)
    assert len({tuple(d["image"].shape) for d in batch}) > 1
    model.train()
    losses = model(batch)
    assert set(losses) == {"hm_loss", "wh_loss", "off_loss"} and all(torch.isfinite(v).item() for v in losses.values())
    model.eval()
    with torch.no_grad():
        out = model([{"image": d["image"], "height": d["height"], "width": d["width"]} for d in batch])
    assert len(out) == 3 and out[0]["instances"].image_size == (batch[0]["height"], batch[0]["width"])
    DatasetCatalog.remove(name)


def test_annotations_to_instances_match_reference_g14():
    """`transform_instance_annotations` -> `annotations_to_instances` -> `filter_empty_instances` against
    what the REFERENCE's own functions (detectron2/data/detection_utils.py:256-287, 362-384, 456-483) returned for the same
    annotations (G14: XYWH / XYXY boxes, boxes leaving the image, empty boxes) under the same affine box transform (scale,
    crop shift, horizontal flip): transformed + clipped boxes, the Instances tensor, and which instances survive"""
    import os
    from detectron2_centernet_amd.data import detection_utils as du
    from detectron2_centernet_amd.data import transforms as T
    from detectron2_centernet_amd.structures import BoxMode

    class Affine(T.Transform):
        def __init__(self, sx, sy, tx, ty):
            self.sx, self.sy, self.tx, self.ty = sx, sy, tx, ty

        def apply_coords(self, coords):
            return np.asarray(coords, dtype=np.float64) * [self.sx, self.sy] + [self.tx, self.ty]

        def apply_image(self, img):
            return img

    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g14_annotations.npz"))
    for i in range(3):
        sx, sy, tx, ty, h, w = d[f"affine{i}"]
        hw = (int(h), int(w))
        annos = [{"bbox": d[f"bbox{i}"][k].tolist(), "bbox_mode": int(d[f"mode{i}"][k]), "category_id": int(d[f"cat{i}"][k])}
                 for k in range(len(d[f"cat{i}"]))]
        out = [du.transform_instance_annotations(dict(a), Affine(sx, sy, tx, ty), hw) for a in annos]
        assert np.array_equal(np.array([o["bbox"] for o in out]), d[f"out_bbox{i}"]), i
        assert all(o["bbox_mode"] == BoxMode.XYXY_ABS for o in out)
        inst = du.annotations_to_instances(out, hw)
        assert np.array_equal(inst.gt_boxes.tensor.numpy(), d[f"inst_boxes{i}"])
        kept = du.filter_empty_instances(inst)
        assert np.array_equal(kept.gt_boxes.tensor.numpy(), d[f"kept_boxes{i}"])
        assert np.array_equal(kept.gt_classes.numpy(), d[f"kept_classes{i}"])
