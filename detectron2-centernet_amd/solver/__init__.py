from .build import FlatSGD, build_optimizer, param_groups
from .lr_scheduler import WarmupMultiStepLR, warmup_multistep_factor

__all__ = ["FlatSGD", "build_optimizer", "param_groups", "WarmupMultiStepLR", "warmup_multistep_factor"]
