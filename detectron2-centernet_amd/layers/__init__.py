from .shape_spec import ShapeSpec
from .deform_conv import DCN, DeformConvV2, ModulatedDeformConv, modulated_deform_conv
from .batch_norm import Conv2d, FrozenBatchNorm2d, get_norm
from . import hipnn

__all__ = ["ShapeSpec", "DCN", "DeformConvV2", "ModulatedDeformConv", "modulated_deform_conv", "hipnn", "Conv2d",
           "FrozenBatchNorm2d", "get_norm"]
