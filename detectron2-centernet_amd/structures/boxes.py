"""XYXY box container with the subset of the reference API the CenterNet path touches
(detectron2/structures/boxes.py: to/clone/area/clip :184-213, nonempty :215-228, scale :271-278, cat)."""
from typing import List, Tuple

import torch


class Boxes:
    def __init__(self, tensor):
        device = tensor.device if isinstance(tensor, torch.Tensor) else torch.device("cpu")
        tensor = torch.as_tensor(tensor, dtype=torch.float32, device=device)
        if tensor.numel() == 0:
            tensor = tensor.reshape((0, 4)).to(dtype=torch.float32, device=device)
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clone(self):
        return Boxes(self.tensor.clone())

    def to(self, *args, **kwargs):
        return Boxes(self.tensor.to(*args, **kwargs))

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def clip(self, box_size: Tuple[int, int]) -> None:
        assert torch.isfinite(self.tensor).all(), "Box tensor contains infinite or NaN!"
        h, w = box_size
        self.tensor[:, 0].clamp_(min=0, max=w)
        self.tensor[:, 1].clamp_(min=0, max=h)
        self.tensor[:, 2].clamp_(min=0, max=w)
        self.tensor[:, 3].clamp_(min=0, max=h)

    def nonempty(self, threshold: float = 0.0):
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def scale(self, scale_x: float, scale_y: float) -> None:
        self.tensor[:, 0::2] *= scale_x
        self.tensor[:, 1::2] *= scale_y

    def get_centers(self):
        return (self.tensor[:, :2] + self.tensor[:, 2:]) / 2

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        b = self.tensor[item]
        assert b.dim() == 2, f"Indexing on Boxes with {item} failed to return a matrix!"
        return Boxes(b)

    def __len__(self):
        return self.tensor.shape[0]

    def __repr__(self):
        return "Boxes(" + str(self.tensor) + ")"

    @staticmethod
    def cat(boxes_list: List["Boxes"]) -> "Boxes":
        assert isinstance(boxes_list, (list, tuple))
        if len(boxes_list) == 0:
            return Boxes(torch.empty(0))
        return Boxes(torch.cat([b.tensor for b in boxes_list], dim=0))

    @property
    def device(self):
        return self.tensor.device

    def __iter__(self):
        yield from self.tensor
