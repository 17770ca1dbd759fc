"""detectron2/modeling/postprocessing.py:11-72 for box-only detectors (the CenterNet path has no masks/keypoints).

The batched device version used by `CenterNet.forward` is `ops.postprocess` (kernel `dec_postprocess_kernel`);
this function is the per-image API of the reference, kept for drop-in use on an `Instances`."""
import torch

from ..structures import Instances


def detector_postprocess(results, output_height, output_width, mask_threshold=0.5):
    if isinstance(output_width, torch.Tensor):
        output_width = output_width.float().item()
    if isinstance(output_height, torch.Tensor):
        output_height = output_height.float().item()
    scale_x, scale_y = output_width / results.image_size[1], output_height / results.image_size[0]
    results = Instances((output_height, output_width), **results.get_fields())
    if results.has("pred_boxes"):
        output_boxes = results.pred_boxes
    elif results.has("proposal_boxes"):
        output_boxes = results.proposal_boxes
    else:
        return results
    output_boxes.scale(scale_x, scale_y)
    output_boxes.clip(results.image_size)
    return results[output_boxes.nonempty()]
