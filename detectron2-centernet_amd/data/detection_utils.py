"""Dataset-dict -> model-input helpers with the names and behaviour of detectron2/data/detection_utils.py (read_image
:167-185 with convert_PIL_to_numpy :61-91, check_image_size :188-210, transform_instance_annotations :256-287,
annotations_to_instances :362-384, filter_empty_instances :456-483).  Boxes only: masks / keypoints are off in the CenterNet
configs (MODEL.MASK_ON / KEYPOINT_ON False)."""
import numpy as np
import torch
from PIL import Image, ImageOps

from ..structures import Boxes, BoxMode, Instances
from . import transforms as T


class SizeMismatchError(ValueError):
    """the image on disk does not have the size its dataset dict declares"""


def convert_PIL_to_numpy(image, format):
    """HWC array; "BGR" is RGB with the channel axis reversed (PIL has no BGR mode), "L" keeps a channel axis"""
    if format is None:
        return np.asarray(image)
    arr = np.asarray(image.convert("RGB" if format == "BGR" else format))
    if format == "BGR":
        return arr[:, :, ::-1]
    return arr[:, :, None] if format == "L" else arr


def read_image(file_name, format=None):
    """uint8 HWC pixels of `file_name` in `format`, EXIF orientation applied"""
    with open(file_name, "rb") as fh:
        return convert_PIL_to_numpy(ImageOps.exif_transpose(Image.open(fh)), format)


def check_image_size(dataset_dict, image):
    """the dict's width / height (when present) must match the pixels; missing ones are filled in"""
    h, w = image.shape[:2]
    declared = (dataset_dict.get("width", w), dataset_dict.get("height", h))
    if declared != (w, h):
        where = " for image " + dataset_dict["file_name"] if "file_name" in dataset_dict else ""
        raise SizeMismatchError("Mismatched (W,H){}, got {}, expect {}".format(where, (w, h), declared))
    dataset_dict["width"], dataset_dict["height"] = w, h


def _xyxy(annotation):
    # the stored object (a json list, or an array from an earlier transform) goes to BoxMode.convert as it is: the
    # reference's conversion of a list runs in float32 (structures/boxes.py)
    return np.asarray(BoxMode.convert(annotation["bbox"], annotation["bbox_mode"], BoxMode.XYXY_ABS), dtype=np.float64).reshape(4)


def transform_instance_annotations(annotation, transforms, image_size):
    """the annotation's box through `transforms`, clipped to the (h, w) of the transformed image, stored as XYXY_ABS;
    modifies and returns `annotation`"""
    tl = transforms if isinstance(transforms, T.Transform) else T.TransformList(list(transforms))
    h, w = image_size
    box = tl.apply_box(_xyxy(annotation)[None])[0]
    annotation["bbox"] = np.minimum(np.maximum(box, 0.0), [w, h, w, h])
    annotation["bbox_mode"] = BoxMode.XYXY_ABS
    return annotation


def annotations_to_instances(annos, image_size):
    out = Instances(image_size)
    coords = np.stack([_xyxy(a) for a in annos]).astype(np.float32) if annos else np.zeros((0, 4), np.float32)
    out.gt_boxes = Boxes(torch.from_numpy(coords))
    out.gt_classes = torch.tensor([int(a["category_id"]) for a in annos], dtype=torch.int64)
    return out


def filter_empty_instances(instances, box_threshold=1e-5):
    """drop boxes whose width or height is not above the threshold"""
    return instances[instances.gt_boxes.nonempty(threshold=box_threshold)]
