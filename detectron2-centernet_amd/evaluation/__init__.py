from .coco_results import instances_to_coco_json, results_to_coco_json
from .evaluator import COCOResultsWriter, DatasetEvaluator, DatasetEvaluators, inference_context, inference_on_dataset

__all__ = ["instances_to_coco_json", "results_to_coco_json", "DatasetEvaluator", "DatasetEvaluators", "COCOResultsWriter",
           "inference_context", "inference_on_dataset"]
