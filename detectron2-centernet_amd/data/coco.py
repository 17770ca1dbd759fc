"""COCO-format instance annotations -> Detectron2 dataset dicts, without pycocotools (absent here; the reference goes
through `pycocotools.coco.COCO`, data/datasets/coco.py:28-196, whose only use on this path is indexing the json by image).
Category ids are remapped to contiguous [0, #categories) when the file's ids are not already 1..N in order, exactly as
the reference does (:62-87)."""
import json
import os

from ..structures import BoxMode
from .catalog import DatasetCatalog, MetadataCatalog


def load_coco_json(json_file, image_root, dataset_name=None):
    with open(json_file) as f:
        data = json.load(f)
    cats = sorted(data.get("categories", []), key=lambda c: c["id"])
    cat_ids = [c["id"] for c in cats]
    id_map = None
    if dataset_name is not None and cats:
        meta = MetadataCatalog.get(dataset_name)
        meta.thing_classes = [c["name"] for c in cats]
        id_map = {v: i for i, v in enumerate(cat_ids)}
        meta.thing_dataset_id_to_contiguous_id = id_map
    imgs = sorted(data["images"], key=lambda im: im["id"])
    by_image = {}
    for ann in data.get("annotations", []):
        by_image.setdefault(ann["image_id"], []).append(ann)
    records = []
    for im in imgs:
        rec = {"file_name": os.path.join(image_root, im["file_name"]), "height": im["height"], "width": im["width"],
               "image_id": im["id"]}
        objs = []
        for ann in by_image.get(im["id"], []):
            assert ann["image_id"] == im["id"]
            assert ann.get("ignore", 0) == 0, '"ignore" in COCO json file is not supported.'
            obj = {k: ann[k] for k in ("iscrowd", "bbox", "category_id") if k in ann}
            obj["bbox_mode"] = BoxMode.XYWH_ABS
            if id_map:
                obj["category_id"] = id_map[obj["category_id"]]
            objs.append(obj)
        rec["annotations"] = objs
        records.append(rec)
    return records


def register_coco_instances(name, metadata, json_file, image_root):
    """reference: data/datasets/register_coco.py (lazy load on first DatasetCatalog.get)"""
    DatasetCatalog.register(name, lambda: load_coco_json(json_file, image_root, name))
    MetadataCatalog.get(name).set(json_file=json_file, image_root=image_root, evaluator_type="coco", **metadata)
