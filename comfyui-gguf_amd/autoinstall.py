"""The literal drop-in: this directory copied (or symlinked) into ``ComfyUI/custom_nodes/`` next to ``ComfyUI-GGUF``.

ComfyUI imports every ``custom_nodes/*/__init__.py`` (the way the reference itself is activated, reference __init__.py:1-9).  When
this package is imported that way, ``arm()`` finds the ComfyUI-GGUF package -- already imported, or imported later: both orders
happen, ComfyUI walks ``custom_nodes`` in directory order -- and calls ``install(ref_dequant, ref_ops)`` on it (install.py; options
from the ``GGQ_*`` environment variables described there).  ComfyUI-GGUF is recognised by what this path needs from it, not by its
directory name: a package whose ``.ops`` defines ``GGMLOps`` and ``GGMLTensor`` and whose ``.dequant`` defines ``dequantize_tensor``
and ``dequantize_functions`` (reference ops.py:44,213; dequant.py:15,287).  Nothing else of ComfyUI is touched; no node is added.
"""
import importlib.abc
import logging
import sys

import os

log = logging.getLogger("comfyui-gguf_amd")
_state = {"armed": False, "installed": None, "gave_up": False}


def _is_ref_ops(mod):
    return all(hasattr(mod, a) for a in ("GGMLOps", "GGMLTensor", "GGMLLayer")) and "." in getattr(mod, "__name__", "")


def _install_over(ops_mod):
    """``ops_mod`` = the reference's ops module, fully executed (so its ``.dequant`` sibling exists).  Never raises into the
    reference's import: a failure is logged ONCE with its traceback, the reference keeps its own torch path, and this package stops
    trying (``gave_up``: a second ``*.ops`` import must not retry and log again)."""
    if _state["installed"] is not None or _state["gave_up"]:
        return
    pkg = ops_mod.__name__.rsplit(".", 1)[0]
    deq = sys.modules.get(pkg + ".dequant")
    if deq is None or not hasattr(deq, "dequantize_tensor") or not hasattr(deq, "dequantize_functions"):
        return
    try:
        from . import install as inst
        loader = sys.modules.get(pkg + ".loader")
        if loader is None and inst._env_flag("GGQ_NATIVE_READER"):
            # the hook fires right after <package>.ops has executed -- before nodes.py gets to `from .loader import ...`.  The option rebinds a name inside
            # the loader module, so import it now (it only needs .ops and .dequant, both executed); a checkout without loader.py keeps gguf-py's reader.
            try:
                loader = importlib.import_module(pkg + ".loader")
            except Exception:                              # noqa: BLE001
                log.warning("comfyui-gguf_amd: GGQ_NATIVE_READER=1 but %s.loader cannot be imported; gguf-py's GGUFReader stays", pkg)
        inst.install(deq, ops_mod, loader, native_reader=None if loader is not None else False)
        _state["installed"] = pkg
        log.info("comfyui-gguf_amd: MI355X HIP dequant path installed over %s (dequantize, dequantize_tensor%s)", pkg, inst.describe(deq))
    except Exception:                                      # noqa: BLE001 -- see docstring
        _state["gave_up"] = True
        log.exception("comfyui-gguf_amd: could not install over %s; the reference's torch path stays in place", pkg)


def _may_be_reference_ops(fullname):
    """Cheap test BEFORE any other finder is consulted: ``<package>.ops`` where <package> is already imported and has a ``dequant``
    sibling (imported, or a dequant.py in its directory).  torchvision.ops, comfy.ops & co. fail it and pass through untouched."""
    parent = fullname[:-len(".ops")]
    if parent + ".dequant" in sys.modules:
        return True
    pkg = sys.modules.get(parent)
    for d in getattr(pkg, "__path__", None) or ():
        if os.path.isfile(os.path.join(d, "dequant.py")):
            return True
    return False


class _HookedLoader(importlib.abc.Loader):
    """Proxy of the loader the regular finders chose for ``<package>.ops``: same module creation and execution, then ``after(module)``.
    The real loader object is left untouched (round 4 rebound ``exec_module`` on the loader instance itself)."""

    def __init__(self, loader, after):
        self._loader, self._after = loader, after

    def create_module(self, spec):
        return self._loader.create_module(spec)

    def exec_module(self, module):
        self._loader.exec_module(module)
        self._after(module)

    def __getattr__(self, name):                           # get_code / get_source / get_filename / is_package / ... of the real loader
        return getattr(self._loader, name)


class _AfterOpsImport(importlib.abc.MetaPathFinder):
    """One-shot post-import hook: lets the regular finders locate ``<package>.ops``, then runs ``_install_over`` right after the
    module body has executed (the spec's loader is wrapped in a proxy).  Removes itself once it has installed -- or given up."""

    def _retire(self):
        if self in sys.meta_path:
            sys.meta_path.remove(self)

    def find_spec(self, fullname, path, target=None):
        if _state["installed"] is not None or _state["gave_up"]:
            self._retire()
            return None
        if not fullname.endswith(".ops") or not _may_be_reference_ops(fullname):
            return None
        for finder in sys.meta_path:
            spec = finder.find_spec(fullname, path, target) if finder is not self and hasattr(finder, "find_spec") else None
            if spec is not None:
                break
        else:
            return None
        loader = spec.loader
        if loader is not None and hasattr(loader, "exec_module"):
            def after(module):
                if _is_ref_ops(module):
                    _install_over(module)
                    if _state["installed"] is not None or _state["gave_up"]:
                        self._retire()
            spec.loader = _HookedLoader(loader, after)
        return spec


def arm():
    """Install now if ComfyUI-GGUF is already imported, else as soon as it is.  Raises (in THIS package's import, where ComfyUI
    reports it against this directory) if the HIP library is missing: there is no fallback to fall back to."""
    if _state["armed"]:
        return _state["installed"]
    from . import _native
    _native.lib()
    _state["armed"] = True
    for mod in list(sys.modules.values()):
        if mod is not None and getattr(mod, "__name__", "").endswith(".ops") and _is_ref_ops(mod):
            _install_over(mod)
            if _state["installed"] is not None or _state["gave_up"]:
                return _state["installed"]
    sys.meta_path.insert(0, _AfterOpsImport())
    return None
