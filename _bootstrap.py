"""Register the hyphenated package directory ``continual-skeletons_amd/`` as the importable module
``continual_skeletons_amd`` (a hyphen cannot appear in an import statement)."""
import importlib.util
import os
import sys

NAME = "continual_skeletons_amd"
ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "continual-skeletons_amd")


def load():
    if NAME in sys.modules:
        return sys.modules[NAME]
    spec = importlib.util.spec_from_file_location(
        NAME, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR]
    )
    mod = importlib.util.module_from_spec(spec)
    sys.modules[NAME] = mod
    spec.loader.exec_module(mod)
    return mod
