"""Backbone registry: factories registered under the names the yaml files use (`MODEL.BACKBONE.NAME`) and called as
`factory(cfg, input_shape)`; contract of detectron2/modeling/backbone/build.py:9-33."""
from ...layers import ShapeSpec
from ...utils.registry import Registry
from .backbone import Backbone

BACKBONE_REGISTRY = Registry("BACKBONE")


def build_backbone(cfg, input_shape=None):
    """the backbone named by the config; the default input shape has one channel per PIXEL_MEAN entry"""
    shape = input_shape if input_shape is not None else ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
    factory = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)
    net = factory(cfg, shape)
    if not isinstance(net, Backbone):
        raise AssertionError(f"{cfg.MODEL.BACKBONE.NAME} returned {type(net).__name__}, not a Backbone")
    return net
