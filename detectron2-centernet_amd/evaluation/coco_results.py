"""Hand-off of the decoded detections to COCO evaluation: the wire format of
detectron2/evaluation/coco_evaluation.py:321-382 (`instances_to_coco_json`: XYWH boxes, one dict per detection) and the
category-id remapping of `COCOEvaluator._eval_predictions` (:147-163).  Scoring itself (pycocotools) is out of scope."""
import torch


def instances_to_coco_json(instances, img_id):
    """list of {"image_id", "category_id", "bbox" [x, y, w, h], "score"} for one image (boxes only)."""
    n = len(instances) if instances.has("scores") else 0
    if n == 0:
        return []
    boxes = instances.pred_boxes.tensor.detach().float().cpu().clone()
    boxes[:, 2] -= boxes[:, 0]          # BoxMode.XYXY_ABS -> XYWH_ABS (structures/boxes.py:100-103)
    boxes[:, 3] -= boxes[:, 1]
    boxes = boxes.tolist()
    scores = instances.scores.detach().cpu().tolist()
    classes = instances.pred_classes.detach().cpu().tolist()
    return [{"image_id": img_id, "category_id": classes[k], "bbox": boxes[k], "score": scores[k]} for k in range(n)]


def results_to_coco_json(outputs, image_ids, dataset_id_to_contiguous_id=None):
    """model outputs (list of {"instances": Instances}) -> flat COCO results list; with the dataset's
    `thing_dataset_id_to_contiguous_id` the contiguous class indices are mapped back to dataset category ids."""
    results = []
    for out, img_id in zip(outputs, image_ids):
        results.extend(instances_to_coco_json(out["instances"], img_id))
    if dataset_id_to_contiguous_id is not None:
        rev = {v: k for k, v in dataset_id_to_contiguous_id.items()}
        for r in results:
            assert r["category_id"] in rev, f"A prediction has category_id={r['category_id']}, which is not available in the dataset."
            r["category_id"] = rev[r["category_id"]]
    return results
