"""Import shim: the package directory is named ``d3feat.pytorch_amd`` (not a legal dotted import), so this module
loads it under the importable name ``d3feat_pytorch_amd`` and replaces itself in ``sys.modules``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "d3feat.pytorch_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
