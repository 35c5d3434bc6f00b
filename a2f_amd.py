"""Import shim: the package directory is named ``audio2face-pytorch_amd`` (not a Python identifier);
``import a2f_amd`` loads it under that name, sub-packages included (``a2f_amd.model.voca`` ...)."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "audio2face-pytorch_amd")
_spec = importlib.util.spec_from_file_location("a2f_amd", os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["a2f_amd"] = _mod
_spec.loader.exec_module(_mod)
