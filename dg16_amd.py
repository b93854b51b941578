"""Import shim: the package directory `distributed-groth16_amd/` is not a valid Python identifier, so
`import dg16_amd` loads it under this name."""

import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "distributed-groth16_amd")
_spec = importlib.util.spec_from_file_location("dg16_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dg16_amd"] = _mod
_spec.loader.exec_module(_mod)
