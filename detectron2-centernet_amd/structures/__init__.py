from .boxes import Boxes
from .image_list import ImageList
from .instances import Instances

__all__ = ["Boxes", "ImageList", "Instances"]
