"""The serving-side split of CenterNet used by the reference's export path (detectron2/export/meta_modeling.py:151-201):
`convert_inputs` (host: normalise + pad + stack) -> `inference` (the network: {images, im_info} -> {hm after sigmoid+clamp,
wh, reg}) -> `convert_outputs` (decode + per-image filtering + detector_postprocess).  The reference traces `inference` to
ONNX/TensorRT; here the same three stages run on the HIP kernels, so a serving stack built around that contract can call
them unchanged.  Tensors are logical NCHW like the reference's (views of the NHWC device buffers)."""
import torch

from ..layers import hipnn
from ..modeling.postprocessing import detector_postprocess


class CenterNetModel:
    def __init__(self, cfg, torch_model):
        from ..modeling.meta_arch.centernet import CenterNet
        assert isinstance(torch_model, CenterNet)
        self._cfg, self._wrapped_model = cfg, torch_model
        torch_model.eval()

    def convert_inputs(self, batched_inputs):
        images, _ = self._wrapped_model.preprocess_image(batched_inputs)
        return {"images": images.tensor, "im_info": torch.tensor(images.image_sizes), "_nhwc": getattr(images, "nhwc", None)}

    @torch.no_grad()
    def inference(self, inputs):
        m = self._wrapped_model
        x = inputs.get("_nhwc")
        if x is None:    # a caller-made normalised NCHW batch: bring it into the kernels' layout (8 channels, 3 used)
            x = hipnn.to_nhwc(inputs["images"], m._ctx, pad_to=8)
        hm, wh, reg = m._network_outputs(x, apply_sigmoid=True)     # hm: sigmoid + clamp(1e-4, 1 - 1e-4) in the head epilogue
        return {"hm": hipnn.to_nchw_view(hm), "wh": hipnn.to_nchw_view(wh), "reg": hipnn.to_nchw_view(reg)}

    def convert_outputs(self, batched_inputs, inputs, results):
        sizes = [tuple(int(v) for v in s) for s in inputs["im_info"]]
        per_image = self._wrapped_model.inference(results, sizes)
        out = []
        for r, inp, size in zip(per_image, batched_inputs, sizes):
            out.append({"instances": detector_postprocess(r, inp.get("height", size[0]), inp.get("width", size[1]))})
        return out

    def __call__(self, batched_inputs):
        inputs = self.convert_inputs(batched_inputs)
        return self.convert_outputs(batched_inputs, inputs, self.inference(inputs))

    @staticmethod
    def get_input_names():
        return ["images", "im_info"]

    @staticmethod
    def get_output_names():
        return ["hm", "wh", "reg"]
