"""Import alias: the package directory is `zk-saas_amd/` (not a valid Python identifier).

`import zksaas_amd` loads that directory as the package `zksaas_amd`.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zk-saas_amd")
_spec = importlib.util.spec_from_file_location(
    "zksaas_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["zksaas_amd"] = _mod
_spec.loader.exec_module(_mod)
