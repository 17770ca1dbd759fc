"""Dataset-dict -> model-input helpers (reference: detectron2/data/detection_utils.py; read_image :167-185,
convert_PIL_to_numpy :61-91, check_image_size :188-210, transform_instance_annotations :256-316,
annotations_to_instances :362-409, filter_empty_instances :456-483).  Boxes only: masks / keypoints are off in the
CenterNet configs (MODEL.MASK_ON / KEYPOINT_ON False)."""
import numpy as np
import torch
from PIL import Image, ImageOps

from ..structures import Boxes, BoxMode, Instances
from . import transforms as T


class SizeMismatchError(ValueError):
    """the image on disk does not have the size its dataset dict declares"""


def convert_PIL_to_numpy(image, format):
    if format is not None:
        image = image.convert("RGB" if format == "BGR" else format)
    image = np.asarray(image)
    if format == "L":
        image = np.expand_dims(image, -1)
    elif format == "BGR":
        image = image[:, :, ::-1]
    return image


def read_image(file_name, format=None):
    """HWC uint8 array in `format` ("RGB", "BGR", "L"); EXIF orientation applied like the reference (:184)"""
    with open(file_name, "rb") as f:
        image = Image.open(f)
        image = ImageOps.exif_transpose(image)
        return convert_PIL_to_numpy(image, format)


def check_image_size(dataset_dict, image):
    if "width" in dataset_dict or "height" in dataset_dict:
        image_wh = (image.shape[1], image.shape[0])
        expected_wh = (dataset_dict["width"], dataset_dict["height"])
        if image_wh != expected_wh:
            raise SizeMismatchError("Mismatched (W,H){}, got {}, expect {}".format(
                " for image " + dataset_dict["file_name"] if "file_name" in dataset_dict else "", image_wh, expected_wh))
    dataset_dict.setdefault("width", image.shape[1])
    dataset_dict.setdefault("height", image.shape[0])


def transform_instance_annotations(annotation, transforms, image_size):
    """box -> XYXY_ABS, transformed, clipped to the (h, w) of the transformed image; modifies and returns `annotation`"""
    if isinstance(transforms, (tuple, list)):
        transforms = T.TransformList(transforms)
    bbox = BoxMode.convert(np.asarray(annotation["bbox"], dtype=np.float64), annotation["bbox_mode"], BoxMode.XYXY_ABS)
    bbox = transforms.apply_box(np.array([bbox.reshape(4)]))[0].clip(min=0)
    annotation["bbox"] = np.minimum(bbox, list(image_size + image_size)[::-1])
    annotation["bbox_mode"] = BoxMode.XYXY_ABS
    return annotation


def annotations_to_instances(annos, image_size):
    boxes = [np.asarray(BoxMode.convert(np.asarray(o["bbox"], dtype=np.float64), o["bbox_mode"], BoxMode.XYXY_ABS)).reshape(4)
             for o in annos]
    target = Instances(image_size)
    target.gt_boxes = Boxes(torch.as_tensor(np.array(boxes, dtype=np.float32).reshape(-1, 4)))
    target.gt_classes = torch.tensor([int(o["category_id"]) for o in annos], dtype=torch.int64)
    return target


def filter_empty_instances(instances, box_threshold=1e-5):
    return instances[instances.gt_boxes.nonempty(threshold=box_threshold)]
