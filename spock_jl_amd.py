"""Import shim: the package directory is named ``spock.jl_amd`` (not a valid Python identifier), so this
module loads it under the importable name ``spock_jl_amd`` and replaces itself in ``sys.modules``."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "spock.jl_amd")
_spec = _ilu.spec_from_file_location("spock_jl_amd", _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["spock_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
