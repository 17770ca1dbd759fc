from .coco_results import instances_to_coco_json, results_to_coco_json

__all__ = ["instances_to_coco_json", "results_to_coco_json"]
