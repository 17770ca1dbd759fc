"""XYXY box container with the subset of the reference API the CenterNet path touches
(detectron2/structures/boxes.py: to/clone/area/clip :184-213, nonempty :215-228, scale :271-278, cat)."""
from typing import List, Tuple

import torch


class BoxMode:
    """the two absolute box encodings the CenterNet data path meets (structures/boxes.py:18-129): COCO json stores XYWH,
    everything downstream is XYXY"""
    XYXY_ABS = 0
    XYWH_ABS = 1

    @staticmethod
    def convert(box, from_mode, to_mode):
        """the reference's arithmetic (structures/boxes.py:41-129), including its dtypes: a list / tuple (what a COCO json
        annotation holds) goes through `torch.tensor`, i.e. float32 -- all four numbers are rounded, not only the two that
        change --, an ndarray keeps its dtype, a tensor is cloned; same mode: the input object itself (pinned by G14)"""
        import numpy as np
        if from_mode == to_mode:
            return box
        single = isinstance(box, (list, tuple))
        is_numpy = isinstance(box, np.ndarray)
        if single:
            arr = torch.tensor(box)[None, :]
        elif is_numpy:
            arr = torch.from_numpy(np.asarray(box)).clone().reshape(-1, 4)
        else:
            arr = box.clone().reshape(-1, 4)
        if from_mode == BoxMode.XYWH_ABS and to_mode == BoxMode.XYXY_ABS:
            arr[:, 2] += arr[:, 0]
            arr[:, 3] += arr[:, 1]
        elif from_mode == BoxMode.XYXY_ABS and to_mode == BoxMode.XYWH_ABS:
            arr[:, 2] -= arr[:, 0]
            arr[:, 3] -= arr[:, 1]
        else:
            raise NotImplementedError(f"BoxMode conversion {from_mode} -> {to_mode}")
        if single:
            return type(box)(arr.flatten().tolist())
        if is_numpy:
            return arr.numpy() if box.ndim == 2 else arr.numpy().reshape(box.shape)
        return arr


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
