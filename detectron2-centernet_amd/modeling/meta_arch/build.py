"""Meta-architecture registry (`MODEL.META_ARCHITECTURE` -> class called with cfg); contract of
detectron2/modeling/meta_arch/build.py:6-23."""
import torch

from ...utils.registry import Registry

META_ARCH_REGISTRY = Registry("META_ARCH")


def build_model(cfg):
    """instantiate the configured meta-architecture on `cfg.MODEL.DEVICE`; weights are whatever the constructor made"""
    arch = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)
    return arch(cfg).to(torch.device(cfg.MODEL.DEVICE))
