from .detection_checkpoint import DetectionCheckpointer

__all__ = ["DetectionCheckpointer"]
