"""Import the reference's dequant.py VERBATIM from /root/reference (this container only).

TEST INFRASTRUCTURE.  ``gguf`` is not installed here and cannot be (no network).  dequant.py
uses three symbols of it: ``GGMLQuantizationType`` (dequant.py:7,288-300), ``GGML_QUANT_SIZES``
(dequant.py:34) and ``quants.dequantize`` (numpy fallback, dequant.py:27, unreachable for the
formats in scope).  A stub module providing the first two is injected as ``sys.modules['gguf']``
and the reference file is then executed unmodified.  /root/reference does not exist on the GPU
box; callers must check ``available()``.
"""
import importlib.util
import os
import sys
import types

REFERENCE_DIR = os.environ.get("GGQ_REFERENCE_DIR", "/root/reference")
_cached = None


def available():
    return os.path.isfile(os.path.join(REFERENCE_DIR, "dequant.py"))


def _gguf_stub():
    # the package's own restatement of ggml's public enum / block sizes
    from ._pkg import load_package
    qt = load_package().qtypes
    mod = types.ModuleType("gguf")
    mod.GGMLQuantizationType = qt.GGMLQuantizationType
    mod.GGML_QUANT_SIZES = dict(qt.GGML_QUANT_SIZES)
    quants = types.ModuleType("gguf.quants")

    def _no_numpy_fallback(*a, **k):
        raise RuntimeError("gguf stub: gguf.quants.dequantize is third-party code and is not available")

    quants.dequantize = _no_numpy_fallback
    mod.quants = quants
    mod.__ggq_stub__ = True
    return mod


def ensure_gguf():
    """Make ``import gguf`` work: the real package if installed, else the stub."""
    try:
        import gguf  # noqa: F401
    except ImportError:
        stub = _gguf_stub()
        sys.modules["gguf"] = stub
        sys.modules["gguf.quants"] = stub.quants
    return sys.modules["gguf"]


def load_reference_dequant():
    """The reference ``dequant`` module, executed from its own source file, unmodified."""
    global _cached
    if _cached is None:
        if not available():
            raise FileNotFoundError(f"{REFERENCE_DIR}/dequant.py not present (expected on the GPU box)")
        ensure_gguf()
        spec = importlib.util.spec_from_file_location("ggq_reference_dequant", os.path.join(REFERENCE_DIR, "dequant.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _cached = mod
    return _cached
