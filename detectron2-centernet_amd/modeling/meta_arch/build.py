"""detectron2/modeling/meta_arch/build.py:6-23."""
import torch

from ...utils.registry import Registry

META_ARCH_REGISTRY = Registry("META_ARCH")


def build_model(cfg):
    """Builds `cfg.MODEL.META_ARCHITECTURE` and moves it to `cfg.MODEL.DEVICE` (does not load weights)."""
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model
