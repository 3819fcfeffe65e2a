"""Loads the package directory `fluent-bit_amd/` (the hyphen makes it un-importable by name)
and registers it as module `fluent_bit_amd`."""
import importlib.util, os, sys


def load():
    if "fluent_bit_amd" in sys.modules:
        return sys.modules["fluent_bit_amd"]
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location(
        "fluent_bit_amd", os.path.join(here, "fluent-bit_amd", "__init__.py"),
        submodule_search_locations=[os.path.join(here, "fluent-bit_amd")])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["fluent_bit_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
