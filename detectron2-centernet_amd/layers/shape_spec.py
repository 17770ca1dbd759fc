"""Shape descriptor handed to backbone factories (`f(cfg, input_shape)`); same four optional fields, keyword or positional,
immutable and tuple-like, as the reference's ShapeSpec (detectron2/layers/shape_spec.py)."""
from typing import NamedTuple, Optional


class ShapeSpec(NamedTuple):
    channels: Optional[int] = None
    height: Optional[int] = None
    width: Optional[int] = None
    stride: Optional[int] = None
