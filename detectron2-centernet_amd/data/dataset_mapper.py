"""The CenterNet project's dataset mapper.  Contract (reference: projects/CenterNet/center_net/dataset_mapper.py:17-175):
input one Detectron2 dataset dict; output a copy with "image" (uint8 CHW tensor, channel order INPUT.FORMAT) and, when
training, "instances" (gt_boxes XYXY in the resized image, gt_classes int64; crowd and empty boxes removed); the
"annotations" list is consumed.  Pipeline: read -> ResizeShortestEdge -> four colour jitters, each applied with probability
0.15 (training only) -> boxes through the same transform list."""
import copy

import numpy as np
import torch

from . import detection_utils as du
from . import transforms as T

# colour jitters of the training pipeline (:36-43 of the reference mapper): (augmentation, constructor arguments)
_COLOUR_JITTER = ((T.RandomContrast, (0.8, 1.2)), (T.RandomBrightness, (0.8, 1.2)), (T.RandomSaturation, (0.8, 1.2)),
                  (T.RandomLighting, (0.8,)))
_JITTER_PROB = 0.15


def bulb_traffic_light_augmentation(cfg, is_train):
    """resize rule from INPUT.{MIN,MAX}_SIZE_{TRAIN,TEST}; training adds the colour jitters"""
    inp = cfg.INPUT
    if is_train:
        sizes, longest, style = inp.MIN_SIZE_TRAIN, inp.MAX_SIZE_TRAIN, inp.MIN_SIZE_TRAIN_SAMPLING
        if style == "range" and len(sizes) != 2:
            raise AssertionError("more than 2 ({}) min_size(s) are provided for ranges".format(len(sizes)))
    else:
        sizes, longest, style = inp.MIN_SIZE_TEST, inp.MAX_SIZE_TEST, "choice"
    pipeline = [T.ResizeShortestEdge(sizes, longest, style)]
    if is_train:
        pipeline += [T.RandomApply(aug(*args), prob=_JITTER_PROB) for aug, args in _COLOUR_JITTER]
    return pipeline


class TrafficLightDatasetMapper:
    def __init__(self, cfg, is_train=True):
        unsupported = [k for k in ("MASK_ON", "KEYPOINT_ON", "LOAD_PROPOSALS") if getattr(cfg.MODEL, k, False)]
        if unsupported:
            raise NotImplementedError(f"MODEL.{unsupported[0]}: masks / keypoints / proposals are not part of the CenterNet path")
        crop = getattr(cfg.INPUT, "CROP", None)
        if is_train and crop is not None and getattr(crop, "ENABLED", False):
            raise NotImplementedError("INPUT.CROP is off in the CenterNet configs and not built")
        self.is_train = is_train
        self.img_format = cfg.INPUT.FORMAT
        self.augmentation = bulb_traffic_light_augmentation(cfg, is_train)

    def _image(self, record):
        pixels = du.read_image(record["file_name"], format=self.img_format)
        du.check_image_size(record, pixels)
        return T.apply_augmentations(self.augmentation, pixels)

    @staticmethod
    def _targets(annotations, transforms, hw):
        kept = [du.transform_instance_annotations(a, transforms, hw) for a in annotations if not a.get("iscrowd", 0)]
        return du.filter_empty_instances(du.annotations_to_instances(kept, hw))

    def __call__(self, dataset_dict):
        record = copy.deepcopy(dataset_dict)       # the caller's dict (shared by every epoch) is never modified
        pixels, transforms = self._image(record)
        record["image"] = torch.as_tensor(np.ascontiguousarray(pixels.transpose(2, 0, 1)))
        annotations = record.pop("annotations", None)
        if self.is_train and annotations is not None:
            record["instances"] = self._targets(annotations, transforms, pixels.shape[:2])
        return record
