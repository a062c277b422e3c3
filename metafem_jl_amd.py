"""Import shim: the package directory is named ``metafem.jl_amd`` (not a valid dotted module
name), so this module loads it under the importable name ``metafem_jl_amd``:

    import metafem_jl_amd as mf
"""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "metafem.jl_amd")
_spec = _u.spec_from_file_location("metafem_jl_amd", _os.path.join(_dir, "__init__.py"),
                                   submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["metafem_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
