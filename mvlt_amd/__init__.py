"""Import alias: the product package lives in ``medical-vision-langauge-transformer_amd/``
(a directory name that is not a Python identifier); ``import mvlt_amd`` loads it."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                    "medical-vision-langauge-transformer_amd")
_spec = importlib.util.spec_from_file_location(
    "mvlt_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mvlt_amd"] = _mod
_spec.loader.exec_module(_mod)
