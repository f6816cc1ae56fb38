"""Load the product package under the importable name ``comfyui_gguf_amd``.

The package directory is ``comfyui-gguf_amd`` -- a ComfyUI custom-node style directory name whose
hyphen rules out a plain ``import``.  ComfyUI itself imports custom nodes by path; this helper does
the same for tests, bench.py and __graft_entry__.py.
"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "comfyui-gguf_amd")
PKG_NAME = "comfyui_gguf_amd"


def load_package():
    if PKG_NAME in sys.modules:
        return sys.modules[PKG_NAME]
    spec = importlib.util.spec_from_file_location(
        PKG_NAME, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[PKG_NAME] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        sys.modules.pop(PKG_NAME, None)
        raise
    return mod
