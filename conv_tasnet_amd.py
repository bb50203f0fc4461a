"""Import alias: the package directory is ``conv-tasnet_amd/`` (not a Python identifier), so
``import conv_tasnet_amd`` loads it from there under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conv-tasnet_amd")
_spec = importlib.util.spec_from_file_location("conv_tasnet_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["conv_tasnet_amd"] = _mod
_spec.loader.exec_module(_mod)
