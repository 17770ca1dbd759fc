"""yacs-style config tree with `_BASE_` inheritance, enough to ingest the reference's CenterNet yamls
byte-for-byte (detectron2/config/config.py:11-89; projects/CenterNet/configs/COCO-Detection/*.yaml)."""
import ast
import copy
import os

import yaml

BASE_KEY = "_BASE_"


class CfgNode(dict):
    IMMUTABLE = "__immutable__"

    def __init__(self, init_dict=None):
        super().__init__()
        self.__dict__[CfgNode.IMMUTABLE] = False
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    # attribute access
    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.is_frozen():
            raise AttributeError(f"Attempted to set {name} to {value}, but CfgNode is immutable")
        self[name] = value

    def is_frozen(self):
        return self.__dict__[CfgNode.IMMUTABLE]

    def _set_immutable(self, flag):
        self.__dict__[CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_immutable(flag)

    def freeze(self):
        self._set_immutable(True)

    def defrost(self):
        self._set_immutable(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        out.__dict__[CfgNode.IMMUTABLE] = self.__dict__[CfgNode.IMMUTABLE]
        return out

    # ---- merging ----
    @staticmethod
    def _decode(v):
        """yacs semantics: strings that parse as python literals become those literals
        (this is how '(640, 672)' and '("bulb_train",)' in the yaml become tuples)."""
        if not isinstance(v, str):
            return v
        try:
            return ast.literal_eval(v)
        except (ValueError, SyntaxError):
            return v

    @staticmethod
    def _coerce(new, old, key):
        if old is None or type(new) is type(old):
            return new
        if isinstance(old, (list, tuple)) and isinstance(new, (list, tuple)):
            return type(old)(new)
        if isinstance(old, float) and isinstance(new, int):
            return float(new)
        if isinstance(old, int) and not isinstance(old, bool) and isinstance(new, float) and new == int(new):
            return new
        if isinstance(old, str) and isinstance(new, str):
            return new
        raise ValueError(f"Type mismatch ({type(old)} vs. {type(new)}) for config key: {key}")

    def _merge(self, other, path):
        for k, v in other.items():
            full = ".".join(path + [k])
            if isinstance(v, dict):
                if k in self and isinstance(self[k], CfgNode):
                    self[k]._merge(v, path + [k])
                elif k in self:
                    raise ValueError(f"config key {full} is not a section")
                else:
                    raise KeyError(f"Non-existent config key: {full}")
            else:
                if k not in self:
                    raise KeyError(f"Non-existent config key: {full}")
                self[k] = self._coerce(self._decode(v), self[k], full)

    @staticmethod
    def load_yaml_with_base(filename):
        with open(filename, "r") as f:
            cfg = yaml.safe_load(f) or {}

        def merge_a_into_b(a, b):
            for k, v in a.items():
                if isinstance(v, dict) and k in b and isinstance(b[k], dict):
                    merge_a_into_b(v, b[k])
                else:
                    b[k] = v

        if BASE_KEY in cfg:
            base = cfg.pop(BASE_KEY)
            if base.startswith("~"):
                base = os.path.expanduser(base)
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(filename), base)  # relative to the including file
            base_cfg = CfgNode.load_yaml_with_base(base)
            merge_a_into_b(cfg, base_cfg)
            return base_cfg
        return cfg

    def merge_from_file(self, cfg_filename):
        loaded = CfgNode.load_yaml_with_base(cfg_filename)
        loaded.pop("VERSION", None) if "VERSION" not in self else None
        self._merge(loaded, [])

    def merge_from_other_cfg(self, other):
        self._merge(other, [])

    def merge_from_list(self, cfg_list):
        assert len(cfg_list) % 2 == 0, f"Override list has odd length: {cfg_list}"
        for full, v in zip(cfg_list[0::2], cfg_list[1::2]):
            node = self
            keys = full.split(".")
            for k in keys[:-1]:
                if k not in node:
                    raise KeyError(f"Non-existent config key: {full}")
                node = node[k]
            if keys[-1] not in node:
                raise KeyError(f"Non-existent config key: {full}")
            node[keys[-1]] = self._coerce(self._decode(v), node[keys[-1]], full)

    def dump(self):
        def plain(n):
            return {k: plain(v) if isinstance(v, CfgNode) else (list(v) if isinstance(v, tuple) else v) for k, v in n.items()}
        return yaml.safe_dump(plain(self))


def get_cfg():
    """A fresh copy of the defaults (detectron2/config/config.py:92-103)."""
    from .defaults import _C

    return _C.clone()
