"""Import alias: ``import dimx`` loads the package that lives in
``dyadic-interaction-modeling_amd/`` (a directory name Python cannot import
directly because of the hyphens).  After this module runs, ``sys.modules['dimx']``
is the real package and ``import dimx.<sub>`` resolves inside that directory.
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "dyadic-interaction-modeling_amd")
_spec = importlib.util.spec_from_file_location(
    "dimx", os.path.join(_PKG_DIR, "__init__.py"),
    submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dimx"] = _mod
_spec.loader.exec_module(_mod)
