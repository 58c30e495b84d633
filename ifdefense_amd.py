"""Import shim: ``import ifdefense_amd`` loads the package that lives in ``if-defense_amd/``.

The directory name is fixed by the project layout and is not a valid Python identifier, so this
module replaces itself in ``sys.modules`` with the real package.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "if-defense_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
