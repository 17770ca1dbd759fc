"""Padded image batch (detectron2/structures/image_list.py:10-130, incl. the fork's max_height/max_width)."""
from typing import List, Sequence, Tuple

import torch
from torch.nn import functional as F


class ImageList:
    def __init__(self, tensor: torch.Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)

    def __getitem__(self, idx):
        size = self.image_sizes[idx]
        return self.tensor[idx, ..., : size[0], : size[1]]

    def to(self, *args, **kwargs):
        return ImageList(self.tensor.to(*args, **kwargs), self.image_sizes)

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def padded_size(sizes, size_divisibility=0, max_height=0, max_width=0):
        """(Hp, Wp) the batch is padded to: per-dimension max, rounded up to the divisibility."""
        h, w = max(s[0] for s in sizes), max(s[1] for s in sizes)
        if size_divisibility > 1:
            d = size_divisibility
            h, w = (h + d - 1) // d * d, (w + d - 1) // d * d
            if max_height > 0 and max_width > 0:
                mh, mw = (max_height + d - 1) // d * d, (max_width + d - 1) // d * d
                assert h <= mh and w <= mw
                h, w = mh, mw
        return h, w

    @staticmethod
    def from_tensors(tensors: Sequence[torch.Tensor], size_divisibility: int = 0, pad_value: float = 0.0,
                     max_height: int = 0, max_width: int = 0) -> "ImageList":
        assert len(tensors) > 0 and isinstance(tensors, (tuple, list))
        for t in tensors:
            assert isinstance(t, torch.Tensor), type(t)
            assert t.shape[1:-2] == tensors[0].shape[1:-2], t.shape
        image_sizes = [tuple(im.shape[-2:]) for im in tensors]
        hp, wp = ImageList.padded_size(image_sizes, size_divisibility, max_height, max_width)
        batch_shape = (len(tensors),) + tuple(tensors[0].shape[:-2]) + (hp, wp)
        batched = tensors[0].new_full(batch_shape, pad_value)
        for img, pad_img in zip(tensors, batched):
            pad_img[..., : img.shape[-2], : img.shape[-1]].copy_(img)
        return ImageList(batched.contiguous(), image_sizes)
