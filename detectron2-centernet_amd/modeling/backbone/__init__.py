from .backbone import Backbone
from .build import BACKBONE_REGISTRY, build_backbone
from .resnet import BasicBlock, BasicStem, BottleneckBlock, ResNet, build_resnet_backbone
from .vovnet import VoVNet, build_vovnet_backbone
from .dla import DLA, DLA34, DLAUp, IDAUp, DLABasicBlock, Root, Tree, build_dla34_backbone, fill_up_weights

__all__ = [k for k in globals().keys() if not k.startswith("_")]
