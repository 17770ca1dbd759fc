"""Dataset / metadata catalogs (detectron2/data/catalog.py:13-236) plus a synthetic COCO-shaped source.

`CenterNet.__init__` requires `DatasetCatalog.get(cfg.DATASETS.TRAIN[0])` and takes the class count from
`MetadataCatalog.get(...).thing_classes` (centernet.py:59-63).  The reference yaml names `bulb_train`, a private
dataset (detectron2/data/datasets/builtin.py:251-279) that is not available; `register_synthetic` registers
any such name with generated 512x512 samples so the yaml loads unchanged."""
import types

import torch


class _DatasetCatalog:
    def __init__(self):
        self._registered = {}

    def register(self, name, func):
        assert callable(func), "You must register a function with `DatasetCatalog.register`!"
        assert name not in self._registered, f"Dataset '{name}' is already registered!"
        self._registered[name] = func

    def get(self, name):
        try:
            f = self._registered[name]
        except KeyError:
            raise KeyError(f"Dataset '{name}' is not registered! Available datasets are: {', '.join(self._registered)}")
        return f()

    def list(self):
        return list(self._registered.keys())

    def remove(self, name):
        self._registered.pop(name)

    def clear(self):
        self._registered.clear()

    def __contains__(self, name):
        return name in self._registered


class Metadata(types.SimpleNamespace):
    name: str = "N/A"

    def as_dict(self):
        return dict(self.__dict__)

    def set(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)
        return self

    def get(self, key, default=None):
        return getattr(self, key, default)


class _MetadataCatalog:
    def __init__(self):
        self._name_to_meta = {}

    def get(self, name):
        assert len(name)
        if name not in self._name_to_meta:
            self._name_to_meta[name] = Metadata(name=name)
        return self._name_to_meta[name]

    def list(self):
        return list(self._name_to_meta.keys())

    def remove(self, name):
        self._name_to_meta.pop(name)


DatasetCatalog = _DatasetCatalog()
MetadataCatalog = _MetadataCatalog()


def synthetic_sample(index, size=512, num_classes=80, max_boxes=32, seed=1234):
    """One COCO-shaped synthetic sample: uint8 image, 1..max_boxes boxes with
    w,h ~ U[8,256] inside the image, class ~ U{0..C-1}.  Deterministic in (seed, index)."""
    g = torch.Generator().manual_seed(seed * 100003 + index)
    image = torch.randint(0, 256, (3, size, size), generator=g, dtype=torch.uint8)
    n = int(torch.randint(1, max_boxes + 1, (1,), generator=g))
    wh = torch.rand(n, 2, generator=g) * min(248, size - 8) + 8
    ctr = torch.rand(n, 2, generator=g) * (size - wh) + wh / 2
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], dim=1)
    classes = torch.randint(0, num_classes, (n,), generator=g)
    return {"image": image, "boxes": boxes, "classes": classes, "height": size, "width": size}


def register_synthetic(name, num_classes=80, length=1000, size=512):
    """Registers `name` (e.g. the yaml's "bulb_train") as a synthetic COCO-shaped dataset."""
    if name in DatasetCatalog:
        return

    def _load():
        return [{"file_name": f"synthetic://{name}/{i}", "image_id": i, "height": size, "width": size}
                for i in range(length)]

    DatasetCatalog.register(name, _load)
    MetadataCatalog.get(name).set(thing_classes=[f"class_{i}" for i in range(num_classes)], synthetic=True)
