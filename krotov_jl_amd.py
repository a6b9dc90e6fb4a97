"""Import shim: the package directory is named `krotov.jl_amd` (not a valid Python identifier),
so `import krotov_jl_amd` loads it from that directory under this module name."""
import importlib.util
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
_pkg = os.path.join(_root, "krotov.jl_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_pkg, "__init__.py"),
                                               submodule_search_locations=[_pkg])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
