"""Loader for the host package, whose directory name (`compressedsensing.jl_amd`, fixed by the
repo layout contract) contains a dot and therefore cannot be named in an `import` statement.

    from csmp_pkg import load; cs = load()        # -> module `compressedsensing_jl_amd`
"""
import importlib.util
import os
import sys

_NAME = "compressedsensing_jl_amd"
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "compressedsensing.jl_amd")


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_ROOT, "__init__.py"), submodule_search_locations=[_ROOT])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
