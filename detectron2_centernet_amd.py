"""Import shim: the package directory is named ``detectron2-centernet_amd`` (not a valid Python
identifier), so ``import detectron2_centernet_amd`` loads it from that directory."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "detectron2-centernet_amd")
_spec = importlib.util.spec_from_file_location(
    "detectron2_centernet_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["detectron2_centernet_amd"] = _mod
_spec.loader.exec_module(_mod)
