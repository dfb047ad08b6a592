"""Import shim: makes the package directory ``ternary-vllm_amd/`` importable as ``ternary_vllm_amd``.

A hyphen cannot appear in an import statement, so this module loads the directory's
``__init__.py`` under the importable name and replaces itself in ``sys.modules`` with it.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ternary-vllm_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
