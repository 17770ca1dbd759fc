from .meta_modeling import CenterNetModel

__all__ = ["CenterNetModel"]
