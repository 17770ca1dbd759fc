"""Per-image container of equally long fields with attribute access -- the public behaviour of
detectron2/structures/instances.py (`image_size`, `set/get/has/remove/get_fields`, `.to()`, indexing by int / slice /
mask, `len`, `cat`, repr) on a different internal layout: an ordered name -> value table plus one cached length."""
import itertools

import torch

_RESERVED = ("_hw", "_table", "_n")


def _concat(values):
    """fields of several Instances joined along the instance axis"""
    first = values[0]
    if isinstance(first, torch.Tensor):
        return torch.cat(values, dim=0)
    if isinstance(first, list):
        return list(itertools.chain.from_iterable(values))
    joiner = getattr(type(first), "cat", None)
    if joiner is None:
        raise ValueError(f"Unsupported type {type(first)} for concatenation")
    return joiner(values)


class Instances:
    def __init__(self, image_size, **fields):
        object.__setattr__(self, "_hw", image_size)
        object.__setattr__(self, "_table", {})
        object.__setattr__(self, "_n", None)
        for name, value in fields.items():
            self.set(name, value)

    # ---- metadata
    @property
    def image_size(self):
        return self._hw

    def __len__(self):
        if self._n is None:
            raise NotImplementedError("Empty Instances does not support __len__!")
        return self._n

    def __iter__(self):
        raise NotImplementedError("`Instances` object is not iterable!")

    # ---- field access
    def set(self, name, value):
        n = len(value)
        if self._n is not None and self._table and n != self._n:
            raise AssertionError(f"Adding a field of length {n} to a Instances of length {self._n}")
        self._table[name] = value
        object.__setattr__(self, "_n", n)

    def get(self, name):
        return self._table[name]

    def has(self, name):
        return name in self._table

    def remove(self, name):
        self._table.pop(name)
        if not self._table:
            object.__setattr__(self, "_n", None)

    def get_fields(self):
        return self._table

    def __setattr__(self, name, value):
        if name in _RESERVED or name.startswith("_"):
            object.__setattr__(self, name, value)
        else:
            self.set(name, value)

    def __getattr__(self, name):
        table = self.__dict__.get("_table")
        if table is None or name not in table:
            raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
        return table[name]

    # ---- derived containers
    def _like(self, convert):
        out = Instances(self._hw)
        for name, value in self._table.items():
            out.set(name, convert(value))
        return out

    def to(self, *args, **kwargs):
        return self._like(lambda v: v.to(*args, **kwargs) if hasattr(v, "to") else v)

    def __getitem__(self, item):
        if type(item) == int:
            n = len(self)
            if not -n <= item < n:
                raise IndexError("Instances index out of range!")
            item = slice(item, None, n)          # one element, kept as a length-1 container
        return self._like(lambda v: v[item])

    @staticmethod
    def cat(instance_lists):
        if not instance_lists or not all(isinstance(i, Instances) for i in instance_lists):
            raise AssertionError("cat() takes a non-empty list of Instances")
        head = instance_lists[0]
        if len(instance_lists) == 1:
            return head
        if any(i.image_size != head.image_size for i in instance_lists[1:]):
            raise AssertionError("cat(): image sizes differ")
        out = Instances(head.image_size)
        for name in head.get_fields():
            out.set(name, _concat([i.get(name) for i in instance_lists]))
        return out

    def __repr__(self):
        fields = ", ".join(f"{k}: {v}" for k, v in self._table.items())
        return (f"{type(self).__name__}(num_instances={self._n or 0}, image_height={self._hw[0]}, "
                f"image_width={self._hw[1]}, fields=[{fields}])")

    __str__ = __repr__
