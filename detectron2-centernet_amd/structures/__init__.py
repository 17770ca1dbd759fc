from .boxes import Boxes, BoxMode
from .image_list import ImageList
from .instances import Instances

__all__ = ["Boxes", "BoxMode", "ImageList", "Instances"]
