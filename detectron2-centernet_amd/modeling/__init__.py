from .backbone import BACKBONE_REGISTRY, Backbone, build_backbone, DLA, DLA34, DLAUp, IDAUp, build_dla34_backbone
from .meta_arch import META_ARCH_REGISTRY, build_model, CenterNet, ctdet_decode
from .postprocessing import detector_postprocess

__all__ = [k for k in globals().keys() if not k.startswith("_")]
