"""Reference-format checkpoint I/O: detectron2/checkpoint/detection_checkpoint.py:11-73 on top of the
file layout fvcore's `Checkpointer` writes -- `torch.save({"model": state_dict, **checkpointables, "iteration": n})`,
a `last_checkpoint` text file in the save directory, `module.` prefixes (DistributedDataParallel) stripped on load.
The state-dict keys of this build equal the reference's (tests/golden/g8_*, g9_*), so a `.pth` trained with the
reference loads unchanged; BatchNorm folding for inference happens lazily at the first forward after a load (the packed
weights are keyed by tensor versions, which `load_state_dict` bumps).  Not handled: Caffe2 / model-zoo `.pkl` files and
their name-matching heuristics (`c2_model_loading.py`)."""
import logging
import os

import torch


class _Incompatible:
    def __init__(self, missing, unexpected, shapes):
        self.missing_keys, self.unexpected_keys, self.incorrect_shapes = missing, unexpected, shapes


class DetectionCheckpointer:
    def __init__(self, model, save_dir="", *, save_to_disk=None, **checkpointables):
        """save_to_disk: None = only the main process writes (detection_checkpoint.py:18-25: `comm.is_main_process()`),
        so data-parallel ranks do not race on the same `.pth` / `last_checkpoint`"""
        from ..utils.comm import is_main_process

        self.model = model.module if hasattr(model, "module") and isinstance(model.module, torch.nn.Module) else model
        self.save_dir = save_dir
        self.save_to_disk = is_main_process() if save_to_disk is None else save_to_disk
        self.checkpointables = dict(checkpointables)
        self.logger = logging.getLogger(__name__)

    # ---- save (fvcore Checkpointer.save) ----
    def save(self, name, **kwargs):
        if not self.save_dir or not self.save_to_disk:
            return None
        data = {"model": {k: v.detach().cpu() for k, v in self.model.state_dict().items()}}
        for k, obj in self.checkpointables.items():
            data[k] = obj.state_dict()
        data.update(kwargs)
        os.makedirs(self.save_dir, exist_ok=True)
        path = os.path.join(self.save_dir, f"{name}.pth")
        torch.save(data, path)
        with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as f:
            f.write(os.path.basename(path))
        return path

    def has_checkpoint(self):
        return bool(self.save_dir) and os.path.exists(os.path.join(self.save_dir, "last_checkpoint"))

    def get_checkpoint_file(self):
        try:
            with open(os.path.join(self.save_dir, "last_checkpoint")) as f:
                return os.path.join(self.save_dir, f.read().strip())
        except OSError:
            return ""

    def resume_or_load(self, path, *, resume=True):
        if resume and self.has_checkpoint():
            return self.load(self.get_checkpoint_file())
        return self.load(path, checkpointables=[])

    # ---- load ----
    def load(self, path, checkpointables=None):
        if not path:
            self.logger.info("No checkpoint found. Initializing model from scratch")
            return {}
        if path.endswith(".pkl"):
            raise NotImplementedError("Caffe2 / model-zoo .pkl checkpoints are not handled (torch .pth only)")
        if not os.path.isfile(path):
            raise FileNotFoundError(f"Checkpoint {path} not found!")
        checkpoint = torch.load(path, map_location="cpu", weights_only=False)
        if "model" not in checkpoint:                      # detection_checkpoint.py:44-46: a bare state dict
            checkpoint = {"model": checkpoint}
        self.incompatible = self._load_model(checkpoint)
        for key in self.checkpointables if checkpointables is None else checkpointables:
            if key in checkpoint:
                self.checkpointables[key].load_state_dict(checkpoint.pop(key))
        return checkpoint                                  # remaining entries, e.g. "iteration"

    def _load_model(self, checkpoint):
        sd = dict(checkpoint.pop("model"))
        if sd and all(k.startswith("module.") for k in sd):  # fvcore _strip_prefix_if_present
            sd = {k[len("module."):]: v for k, v in sd.items()}
        model_sd = self.model.state_dict()
        shapes = []
        for k in list(sd.keys()):
            v = sd[k]
            if not torch.is_tensor(v):
                v = sd[k] = torch.as_tensor(v)
            if k in model_sd and tuple(model_sd[k].shape) != tuple(v.shape):
                shapes.append((k, tuple(v.shape), tuple(model_sd[k].shape)))
                sd.pop(k)                                  # fvcore: skip, report
        res = self.model.load_state_dict(sd, strict=False)
        missing = [k for k in res.missing_keys if k not in ("pixel_mean", "pixel_std")]  # detection_checkpoint.py:64-72
        for k, s1, s2 in shapes:
            self.logger.warning("Skip loading parameter '%s': checkpoint shape %s vs model shape %s", k, s1, s2)
        if missing:
            self.logger.warning("Some model parameters are not found in the checkpoint: %s", missing)
        if res.unexpected_keys:
            self.logger.warning("The checkpoint contains keys that are not used by the model: %s", res.unexpected_keys)
        return _Incompatible(missing, list(res.unexpected_keys), shapes)
