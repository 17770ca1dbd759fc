"""The evaluation loop behind the decode (the role of detectron2/evaluation/evaluator.py:101-180) for the HIP engine.

The reference runs `model(inputs)`, calls `torch.cuda.synchronize()` and only then hands the batch to the evaluator, so the
GPU idles while the host builds COCO records and the host idles while the GPU runs.  The native engine's eval step is
asynchronous (one HIP-graph replay, the detection counts read back through a pinned buffer): this loop keeps ONE batch in
flight -- batch i+1 is enqueued before batch i's result is awaited and processed -- and measures a batch's compute time on
the device, between two HIP events on the launch stream, instead of by stalling the host.  What tools depend on is kept:
the `DatasetEvaluator` protocol (reset / process / evaluate), the warm-up rule (the first min(5, len - 1) batches are not
timed) and the two log lines ("Total inference time", "Total inference pure compute time", evaluator.py:163-174).
`COCOResultsWriter` collects the detections in the COCO results wire format (coco_evaluation.py:109-126, 321-382), gathers
them on the main process and writes the json there; scoring with pycocotools is out of scope (absent here)."""
import datetime
import json
import logging
import os
import time
from collections import OrderedDict
from contextlib import contextmanager

import torch

from ..utils import comm
from .coco_results import instances_to_coco_json


class DatasetEvaluator:
    """reset() once, process(inputs, outputs) per batch, evaluate() at the end (evaluator.py:12-53)"""

    def reset(self):
        pass

    def process(self, inputs, outputs):
        pass

    def evaluate(self):
        pass


class DatasetEvaluators(DatasetEvaluator):
    """several evaluators over one pass of the data; their result dicts are merged on the main process"""

    def __init__(self, evaluators):
        self._evaluators = list(evaluators)

    def reset(self):
        for ev in self._evaluators:
            ev.reset()

    def process(self, inputs, outputs):
        for ev in self._evaluators:
            ev.process(inputs, outputs)

    def evaluate(self):
        merged = OrderedDict()
        for res in [ev.evaluate() for ev in self._evaluators]:       # every evaluator finishes (they may hold collectives)
            if res and comm.is_main_process():
                clash = set(res) & set(merged)
                if clash:
                    raise ValueError(f"evaluators report the same result keys: {sorted(clash)}")
                merged.update(res)
        return merged


class COCOResultsWriter(DatasetEvaluator):
    """per-image `instances_to_coco_json` records (`image_id` from the loader's dicts); `evaluate()` gathers every rank's
    records on the main process, which writes `coco_instances_results.json` (what COCOEvaluator does before scoring,
    coco_evaluation.py:130-170) and returns the counts; the other ranks return {}"""

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
        shards = comm.gather(self._results, dst=0)
        if not comm.is_main_process():
            return {}
        results = [r for shard in shards for r in shard]
        if self._dir:
            os.makedirs(self._dir, exist_ok=True)
            with open(os.path.join(self._dir, "coco_instances_results.json"), "w") as f:
                json.dump(results, f)
        return OrderedDict(bbox={"num_detections": len(results)})


@contextmanager
def inference_context(model):
    """eval mode inside the block, the caller's mode afterwards"""
    was_training = model.training
    model.eval()
    try:
        yield
    finally:
        model.train(was_training)


class _Step:
    """one enqueued batch: its inputs, the engine's handle and the HIP events that bracket its launches"""

    def __init__(self, model, inputs, timed):
        self.inputs, self.timed = inputs, timed
        on_gpu = torch.cuda.is_available() and next(model.parameters()).is_cuda
        self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if on_gpu else None
        if self.ev:
            self.ev[0].record()
        start = getattr(model, "forward_async", None)
        self.handle = start(inputs) if start is not None else None
        self.outputs = None if self.handle is not None else model(inputs)
        if self.ev:
            self.ev[1].record()

    def finish(self):
        """(outputs, device seconds of this batch)"""
        if self.handle is not None:
            self.outputs = self.handle.result()
        if not self.ev:
            return self.outputs, 0.0
        self.ev[1].synchronize()
        return self.outputs, self.ev[0].elapsed_time(self.ev[1]) * 1e-3


def inference_on_dataset(model, data_loader, evaluator):
    """Run `model` over `data_loader` (fixed length) in eval mode and feed `evaluator` (None: timing only); returns
    evaluator.evaluate() or {}.  One batch stays in flight while the previous one is processed on the host.  The timing
    figures are logged in the reference's two lines and kept on `inference_on_dataset.last_timing`."""
    devices = comm.get_world_size()
    log = logging.getLogger(__name__)
    total = len(data_loader)
    log.info("Start inference on {} images".format(total))
    evaluator = DatasetEvaluators([]) if evaluator is None else evaluator
    evaluator.reset()
    warmup = min(5, total - 1)
    wall0, compute_s, timed_iters = time.perf_counter(), 0.0, 0

    def retire(step):
        nonlocal compute_s, timed_iters
        outputs, dev_s = step.finish()
        if step.timed:
            compute_s += dev_s
            timed_iters += 1
        evaluator.process(step.inputs, outputs)

    with inference_context(model), torch.no_grad():
        pending = None
        for idx, inputs in enumerate(data_loader):
            if idx == warmup:
                if pending is not None:      # the warm-up batches end before the clock starts
                    retire(pending)
                    pending = None
                wall0 = time.perf_counter()
            step = _Step(model, inputs, timed=idx >= warmup)
            if pending is not None:
                retire(pending)
            pending = step
        if pending is not None:
            retire(pending)
    wall = time.perf_counter() - wall0
    n = max(1, timed_iters)
    log.info("Total inference time: {} ({:.6f} s / img per device, on {} devices)".format(
        str(datetime.timedelta(seconds=wall)), wall / n, devices))
    log.info("Total inference pure compute time: {} ({:.6f} s / img per device, on {} devices)".format(
        str(datetime.timedelta(seconds=int(compute_s))), compute_s / n, devices))
    inference_on_dataset.last_timing = {"total_s_per_iter": wall / n, "compute_s_per_iter": compute_s / n, "iters": n,
                                        "devices": devices}
    results = evaluator.evaluate()
    return {} if results is None else results
