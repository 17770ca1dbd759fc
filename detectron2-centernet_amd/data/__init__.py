from .catalog import DatasetCatalog, MetadataCatalog
from .build import TrainingSampler, build_detection_train_loader
from .coco import load_coco_json, register_coco_instances
from .dataset_mapper import TrafficLightDatasetMapper

__all__ = ["DatasetCatalog", "MetadataCatalog", "TrainingSampler", "build_detection_train_loader", "load_coco_json",
           "register_coco_instances", "TrafficLightDatasetMapper"]
