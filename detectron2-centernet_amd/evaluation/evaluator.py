"""`inference_on_dataset` and the evaluator interface (detectron2/evaluation/evaluator.py:12-180): the loop that follows the
decode -- feeds the loader's batches to the model in eval mode, times pure compute per image after a warm-up, hands
(inputs, outputs) to the evaluator and returns its result.  `COCOResultsWriter` collects the detections in the COCO
results wire format (coco_evaluation.py:109-126, 321-382); scoring with pycocotools is out of scope (absent here)."""
import datetime
import json
import logging
import os
import time
from collections import OrderedDict
from contextlib import contextmanager

import torch

from ..utils.comm import get_world_size, is_main_process
from .coco_results import instances_to_coco_json


class DatasetEvaluator:
    """process(inputs, outputs) per batch, evaluate() at the end (evaluator.py:12-53)"""

    def reset(self):
        pass

    def process(self, inputs, outputs):
        pass

    def evaluate(self):
        pass


class DatasetEvaluators(DatasetEvaluator):
    def __init__(self, evaluators):
        self._evaluators = list(evaluators)

    def reset(self):
        for e in self._evaluators:
            e.reset()

    def process(self, inputs, outputs):
        for e in self._evaluators:
            e.process(inputs, outputs)

    def evaluate(self):
        results = OrderedDict()
        for e in self._evaluators:
            r = e.evaluate()
            if is_main_process() and r is not None:
                for k, v in r.items():
                    assert k not in results, "Different evaluators produce results with the same key {}".format(k)
                    results[k] = v
        return results


class COCOResultsWriter(DatasetEvaluator):
    """collects `instances_to_coco_json` records per image (`image_id` from the loader's dicts) and writes
    `coco_instances_results.json` like COCOEvaluator does before scoring (coco_evaluation.py:140-170)"""

    def __init__(self, output_dir=None, dataset_id_to_contiguous_id=None):
        self._dir, self._rev = output_dir, None
        if dataset_id_to_contiguous_id is not None:
            self._rev = {v: k for k, v in dataset_id_to_contiguous_id.items()}
        self._results = []

    def reset(self):
        self._results = []

    def process(self, inputs, outputs):
        for inp, out in zip(inputs, outputs):
            recs = instances_to_coco_json(out["instances"], inp["image_id"])
            if self._rev is not None:
                for r in recs:
                    r["category_id"] = self._rev[r["category_id"]]
            self._results.extend(recs)

    def evaluate(self):
        if self._dir:
            os.makedirs(self._dir, exist_ok=True)
            with open(os.path.join(self._dir, "coco_instances_results.json"), "w") as f:
                json.dump(self._results, f)
        return OrderedDict(bbox={"num_detections": len(self._results)})


@contextmanager
def inference_context(model):
    """eval mode for the duration, the previous mode afterwards (evaluator.py:183-196)"""
    training_mode = model.training
    model.eval()
    try:
        yield
    finally:
        model.train(training_mode)


def inference_on_dataset(model, data_loader, evaluator):
    """Run model on the data_loader and evaluate with `evaluator` (None: benchmark only).  Returns evaluator.evaluate()
    (an empty dict if it returns None).  The timing figures are logged and kept on the function's `last_timing`."""
    num_devices = get_world_size()
    logger = logging.getLogger(__name__)
    total = len(data_loader)  # inference data loader must have a fixed length
    logger.info("Start inference on {} images".format(total))
    if evaluator is None:
        evaluator = DatasetEvaluators([])
    evaluator.reset()
    num_warmup = min(5, total - 1)
    start_time = time.perf_counter()
    total_compute_time = 0.0
    with inference_context(model), torch.no_grad():
        for idx, inputs in enumerate(data_loader):
            if idx == num_warmup:
                start_time = time.perf_counter()
                total_compute_time = 0.0
            start_compute_time = time.perf_counter()
            outputs = model(inputs)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            total_compute_time += time.perf_counter() - start_compute_time
            evaluator.process(inputs, outputs)
    total_time = time.perf_counter() - start_time
    n = max(1, total - num_warmup)
    logger.info("Total inference time: {} ({:.6f} s / img per device, on {} devices)".format(
        str(datetime.timedelta(seconds=total_time)), total_time / n, num_devices))
    logger.info("Total inference pure compute time: {} ({:.6f} s / img per device, on {} devices)".format(
        str(datetime.timedelta(seconds=int(total_compute_time))), total_compute_time / n, num_devices))
    inference_on_dataset.last_timing = {"total_s_per_iter": total_time / n, "compute_s_per_iter": total_compute_time / n,
                                        "iters": n, "devices": num_devices}
    results = evaluator.evaluate()
    return {} if results is None else results
