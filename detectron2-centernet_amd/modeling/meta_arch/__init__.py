from .build import META_ARCH_REGISTRY, build_model
from .centernet import CenterNet, ctdet_decode

__all__ = ["META_ARCH_REGISTRY", "build_model", "CenterNet", "ctdet_decode"]
