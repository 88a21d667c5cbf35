"""Import shim: `import sfron` loads the package that lives in the directory
``unified-unlearning-w-remain-geometry_amd/`` (its name is not a valid Python identifier)."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "unified-unlearning-w-remain-geometry_amd")
_spec = importlib.util.spec_from_file_location("sfron", os.path.join(_DIR, "__init__.py"),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sfron"] = _mod
_spec.loader.exec_module(_mod)
