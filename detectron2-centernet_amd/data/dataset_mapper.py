"""The CenterNet project's dataset mapper (reference: projects/CenterNet/center_net/dataset_mapper.py:17-175):
read the image, ResizeShortestEdge, four colour augmentations applied with probability 0.15 each (training), boxes through
the same transforms, `Instances(gt_boxes, gt_classes)` with empty boxes removed.  Output contract: the dataset dict plus
"image" (uint8 CHW tensor in INPUT.FORMAT order) and, in training, "instances"; "annotations" removed."""
import copy

import numpy as np
import torch

from . import detection_utils as utils
from . import transforms as T


def bulb_traffic_light_augmentation(cfg, is_train):
    if is_train:
        min_size, max_size, sample_style = cfg.INPUT.MIN_SIZE_TRAIN, cfg.INPUT.MAX_SIZE_TRAIN, cfg.INPUT.MIN_SIZE_TRAIN_SAMPLING
    else:
        min_size, max_size, sample_style = cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST, "choice"
    if sample_style == "range":
        assert len(min_size) == 2, "more than 2 ({}) min_size(s) are provided for ranges".format(len(min_size))
    augmentation = [T.ResizeShortestEdge(min_size, max_size, sample_style)]
    if is_train:
        augmentation.extend([
            T.RandomApply(T.RandomContrast(intensity_min=0.8, intensity_max=1.2), prob=0.15),
            T.RandomApply(T.RandomBrightness(intensity_min=0.8, intensity_max=1.2), prob=0.15),
            T.RandomApply(T.RandomSaturation(intensity_min=0.8, intensity_max=1.2), prob=0.15),
            T.RandomApply(T.RandomLighting(0.8), prob=0.15),
        ])
    return augmentation


class TrafficLightDatasetMapper:
    def __init__(self, cfg, is_train=True):
        self.augmentation = bulb_traffic_light_augmentation(cfg, is_train)
        crop = getattr(cfg.INPUT, "CROP", None)
        if crop is not None and getattr(crop, "ENABLED", False) and is_train:
            raise NotImplementedError("INPUT.CROP is off in the CenterNet configs and not built")
        self.img_format = cfg.INPUT.FORMAT
        if cfg.MODEL.MASK_ON or cfg.MODEL.KEYPOINT_ON or cfg.MODEL.LOAD_PROPOSALS:
            raise NotImplementedError("masks / keypoints / proposals are not part of the CenterNet path")
        self.is_train = is_train

    def __call__(self, dataset_dict):
        dataset_dict = copy.deepcopy(dataset_dict)
        image = utils.read_image(dataset_dict["file_name"], format=self.img_format)
        utils.check_image_size(dataset_dict, image)
        image, transforms = T.apply_augmentations(self.augmentation, image)
        image_shape = image.shape[:2]
        dataset_dict["image"] = torch.as_tensor(np.ascontiguousarray(image.transpose(2, 0, 1)))
        if not self.is_train:
            dataset_dict.pop("annotations", None)
            return dataset_dict
        if "annotations" in dataset_dict:
            annos = [utils.transform_instance_annotations(obj, transforms, image_shape)
                     for obj in dataset_dict.pop("annotations") if obj.get("iscrowd", 0) == 0]
            instances = utils.annotations_to_instances(annos, image_shape)
            dataset_dict["instances"] = utils.filter_empty_instances(instances)
        return dataset_dict
