"""Loader of the `_rarc_hostmap` CPython extension (csrc/hostmap.c): search answers -> the python objects the
reference's interface returns.  In-tree, built by csrc/Makefile (`__graft_entry__.build()`); no pure-python stand-in
is shipped — a missing extension is a build error and says so."""
import importlib.machinery
import importlib.util
import os
import sysconfig

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_mod = None


def module_path() -> str:
    return os.path.join(_PKG, "lib", "_rarc_hostmap" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def load():
    global _mod
    if _mod is None:
        path = module_path()
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: run `make -C rag-arc_amd/csrc` (or __graft_entry__.build())")
        loader = importlib.machinery.ExtensionFileLoader("_rarc_hostmap", path)
        spec = importlib.util.spec_from_loader("_rarc_hostmap", loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
        _mod = mod
    return _mod
